"""Writes tests/golden/dispatch_plan.json: which kernel / tile / loaders / slab count every convolution layer of the
BASELINE.json configurations dispatches to (gz_conv2d_plan, pure host logic of libgz_hip.so -- runs without a GPU).

    python tests/golden/make_dispatch_golden.py            # regenerate after a DELIBERATE heuristic change

tests/test_dispatch_plan.py compares the library against this table, so an edit of a threshold in csrc/gz_conv.hip
that moves a layer to another kernel shows up as a diff of this file."""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

# (name, C_image_side, K_feature_side, H (image side), KH, S, P): the Conv2d view -- x [N, C, H, H] -> y [N, K, OH, OH].
# A ConvTranspose2d layer is listed by its Conv2d adjoint (its forward is op Dg on the same shape).
DCGAN_LAYERS = [
    ("D.conv_in 3->64 @64", 3, 64, 64, 4, 2, 1),
    ("D.block1 64->128 @32", 64, 128, 32, 4, 2, 1),
    ("D.block2 128->256 @16", 128, 256, 16, 4, 2, 1),
    ("D.block3 256->512 @8", 256, 512, 8, 4, 2, 1),
    ("G.block2 1024->512 @4->8 (adjoint)", 512, 1024, 8, 4, 2, 1),
    ("G.block3 512->256 @8->16 (adjoint)", 256, 512, 16, 4, 2, 1),
    ("G.block4 256->128 @16->32 (adjoint)", 128, 256, 32, 4, 2, 1),
    ("G.out 128->3 @32->64 (adjoint)", 3, 128, 64, 4, 2, 1),
]
HOLOGAN_LAYERS_64 = [
    ("D.conv2d 3->64 @64 k5", 3, 64, 64, 5, 2, 2),
    ("D.blocks0 64->128 @32 k5", 64, 128, 32, 5, 2, 2),
    ("D.blocks1 128->256 @16 k5", 128, 256, 16, 5, 2, 2),
    ("D.blocks2 256->512 @8 k5", 256, 512, 8, 5, 2, 2),
    ("G.projection 1024->1024 @16 k1 (adjoint)", 1024, 1024, 16, 1, 1, 0),
    ("G.block3 1024->256 @16->32 (adjoint)", 256, 1024, 32, 4, 2, 1),
    ("G.block4 256->64 @32->64 (adjoint)", 64, 256, 64, 4, 2, 1),
    ("G.final 64->3 @64 k3", 64, 3, 64, 3, 1, 1),
]
HOLOGAN_LAYERS_128 = [
    ("D.conv2d 3->64 @128 k5", 3, 64, 128, 5, 2, 2),
    ("D.blocks0 64->128 @64 k5", 64, 128, 64, 5, 2, 2),
    ("D.blocks1 128->256 @32 k5", 128, 256, 32, 5, 2, 2),
    ("D.blocks2 256->512 @16 k5", 256, 512, 16, 5, 2, 2),
    ("G.final(EXT-128) 64->3 @64->128 (adjoint)", 3, 64, 128, 4, 2, 1),
]
CONFIGS = [
    ("dc_gan bs=64 (config 1)", 64, DCGAN_LAYERS),
    ("dc_gan bs=128 (metric)", 128, DCGAN_LAYERS),
    ("dc_gan bs=512 (configs 2, 4)", 512, DCGAN_LAYERS),
    ("wgan_gp bs=256 (config 3)", 256, DCGAN_LAYERS),
    ("hologan bs=64 @64x64 (config 5, parity-pinned size)", 64, HOLOGAN_LAYERS_64),
    ("hologan bs=64 @128x128 (config 5, EXT-128)", 64, HOLOGAN_LAYERS_128),
]
OPS = ("F", "Dg", "Wg")


def plan_table():
    from lightning_gan_zoo_amd._lib import lib
    table = {}
    buf = ctypes.create_string_buffer(512)
    for cfg, bs, layers in CONFIGS:
        rows = {}
        for name, C, K, H, KH, S, P in layers:
            OH = (H + 2 * P - KH) // S + 1
            for op in range(3):
                rc = lib.gz_conv2d_plan(op, bs, C, H, H, K, OH, OH, KH, KH, S, P, buf, 512)
                if rc < 0:
                    raise RuntimeError("gz_conv2d_plan failed for %s / %s: %d" % (cfg, name, rc))
                rows["%s | %s" % (name, OPS[op])] = buf.value.decode()
        table[cfg] = rows
    return table


if __name__ == "__main__":
    out = os.path.join(HERE, "dispatch_plan.json")
    json.dump(plan_table(), open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)

"""The dispatch is pinned: gz_conv2d_plan -- the library's own description of the kernel, tile,
operand loaders and slab count a launch takes, pure host logic -- must reproduce tests/golden/dispatch_plan.json for
every convolution layer of the five BASELINE configurations.  A threshold edit in csrc/gz_conv.hip that silently moves
a layer onto a slower kernel (parity stays green) fails here; a deliberate change regenerates the table
(python tests/golden/make_dispatch_golden.py) and shows up as its diff."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_every_baseline_layer_takes_the_pinned_kernel():
    import make_dispatch_golden as G
    golden = json.load(open(os.path.join(HERE, "golden", "dispatch_plan.json")))
    now = G.plan_table()
    assert set(now) == set(golden)
    diffs = []
    for cfg in sorted(golden):
        assert set(now[cfg]) == set(golden[cfg]), cfg
        for key in sorted(golden[cfg]):
            if now[cfg][key] != golden[cfg][key]:
                diffs.append("%s :: %s\n    pinned: %s\n    now:    %s" % (cfg, key, golden[cfg][key], now[cfg][key]))
    assert not diffs, "dispatch changed for %d launches:\n%s" % (len(diffs), "\n".join(diffs))


def test_the_heavy_layers_are_on_the_hand_ordered_skeleton():
    """What the table must say for the launches that carry the step: the 137 / 34 GFLOP k4 s2 p1
    layers of dc_gan at bs 512 run on igemm2 / igemm2w with LDS-DMA loaders, unsplit F / Dg launches carry the
    BatchNorm statistics, the 3-channel transposed convolution is the direct kernel."""
    golden = json.load(open(os.path.join(HERE, "golden", "dispatch_plan.json")))
    t = golden["dc_gan bs=512 (configs 2, 4)"]
    for layer in ("G.block3 512->256 @8->16 (adjoint)", "G.block4 256->128 @16->32 (adjoint)"):
        assert t[layer + " | Dg"].startswith("Dg igemm2<256x128> ConvDgA2") and "bn_stats_rows=0" not in t[layer + " | Dg"]
        assert t[layer + " | F"].startswith("F igemm2<256x128> ConvFwdA2")
        assert t[layer + " | Wg"].startswith("Wg igemm2w<")
    assert t["D.block1 64->128 @32 | Dg"].startswith("Dg igemm2<512x64>")
    assert t["G.out 128->3 @32->64 (adjoint) | Dg"].startswith("Dg direct dgrad_smallc4_k4s2p1<C=3,KS=4>")
    m = golden["dc_gan bs=128 (metric)"]
    assert m["G.out 128->3 @32->64 (adjoint) | Dg"].startswith("Dg direct dgrad_smallc4_k4s2p1<C=3,KS=8>")
    assert m["D.conv_in 3->64 @64 | Wg"].startswith("Wg direct wgrad_k4s2p1_fewc<mfma16x16x4,C=3,KT=4>")


def test_environment_switches_are_ignored_without_gz_experiments():
    """csrc/gz_knobs.h: the GZ_* experiment switches are read only when the process runs with GZ_EXPERIMENTS=1 -- a
    stray variable in a user's shell cannot move a layer onto another kernel."""
    import subprocess
    code = ("import ctypes, sys; sys.path.insert(0, %r); from lightning_gan_zoo_amd._lib import lib; "
            "b = ctypes.create_string_buffer(256); lib.gz_conv2d_plan(1, 512, 256, 16, 16, 512, 8, 8, 4, 4, 2, 1, b, 256); "
            "print(b.value.decode())" % os.path.dirname(HERE))
    base = {k: v for k, v in os.environ.items() if not k.startswith("GZ_")}
    plain = subprocess.run([sys.executable, "-c", code], env=dict(base, GZ_NO_IGEMM2="1"), capture_output=True, text=True)
    forced = subprocess.run([sys.executable, "-c", code], env=dict(base, GZ_NO_IGEMM2="1", GZ_EXPERIMENTS="1"),
                            capture_output=True, text=True)
    assert plain.returncode == 0 and forced.returncode == 0, plain.stderr + forced.stderr
    assert "igemm2<256x128>" in plain.stdout, plain.stdout
    assert "igemm2" not in forced.stdout and "igemm<" in forced.stdout, forced.stdout


def test_the_critics_first_layer_takes_the_fused_backward_at_every_baseline_batch():
    """Round 5: `LeakyReLU(conv(x) + b)` of the critics' first layer (reference standard_networks.py:62-66) has a one-launch
    first-order backward in the D step (gz_conv2d_wgrad_act_partial) and forms the activation mask on load in the G step
    (gz_conv2d_dgrad_act).  Pure host predicates: they must say yes for every BASELINE batch size (and the stacked
    2 x batch pass), for ReLU / LeakyReLU only, and no for the shapes the direct kernels do not take."""
    sys.path.insert(0, os.path.dirname(HERE))
    from lightning_gan_zoo_amd import functional as F
    from lightning_gan_zoo_amd._lib import lib
    for n in (64, 128, 256, 512, 1024):
        shape = (n, 3, 64, 64, 64, 32, 32, 4, 4, 2, 1)
        assert lib.gz_conv2d_wgrad_act_fuses(*shape, F.ACT_LRELU) == 1, n
        assert lib.gz_conv2d_wgrad_act_fuses(*shape, F.ACT_RELU) == 1, n
        assert lib.gz_conv2d_dgrad_act_fuses(*shape, F.ACT_LRELU) == 1, n
        assert lib.gz_conv2d_wgrad_act_fuses(*shape, F.ACT_TANH) == 0 and lib.gz_conv2d_dgrad_act_fuses(*shape, F.ACT_NONE) == 0
    assert lib.gz_conv2d_wgrad_act_fuses(4, 3, 64, 64, 64, 32, 32, 4, 4, 2, 1, F.ACT_LRELU) == 0      # too few row segments
    assert lib.gz_conv2d_wgrad_act_fuses(128, 3, 64, 64, 128, 32, 32, 4, 4, 2, 1, F.ACT_LRELU) == 0   # 128 channels: tile path
    assert lib.gz_conv2d_wgrad_act_fuses(128, 64, 32, 32, 128, 16, 16, 4, 4, 2, 1, F.ACT_LRELU) == 0  # not an image layer
    assert lib.gz_conv2d_dgrad_act_fuses(128, 64, 32, 32, 128, 16, 16, 4, 4, 2, 1, F.ACT_LRELU) == 0
    assert lib.gz_conv2d_dgrad_act_fuses(128, 3, 64, 64, 64, 32, 32, 5, 5, 2, 2, F.ACT_LRELU) == 0    # HoloGAN's k5: unfused

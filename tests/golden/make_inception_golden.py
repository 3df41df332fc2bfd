#!/usr/bin/env python3
"""Generate tests/golden/inception_ref.npz + inception_state_keys.json by running the reference's OWN FID network
code -- ``InceptionV3`` / ``fid_inception_v3`` / ``FIDInceptionA, C, E_1, E_2`` of
/root/reference/core/submodules/gan_stability/metrics/inception.py, unmodified (the vendored copy of the
``pytorch_fid.inception`` that core/callback_inception_metrics.py:8,210-211 imports) -- on seeded weights.

Build container only.  torchvision is not installed, so the module's ``from torchvision import models`` resolves to
oracle/torchvision_inception.py (the layer definitions of torchvision's blocks, restated from its published source);
the weight download (inception.py:13,181-182; no network) is replaced by a provider of seeded weights.  Everything the
FID path adds on top of torchvision -- which blocks are patched, the pooling variants (average without padding in A / C
/ E_1, MAX in E_2), the bilinear resize to 299 x 299, the 2x - 1 rescale, the block partition and the 2048-d pool
output -- is the reference's code executing.

    python tests/golden/make_inception_golden.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

from helpers import seeded_inception_state           # noqa: E402
from oracle import torchvision_inception as tvi      # noqa: E402

REF_FILE = "/root/reference/core/submodules/gan_stability/metrics/inception.py"
SEED = 3
CASES = ((64, 2), (299, 2), (32, 3), (128, 1))       # (image size, batch): inputs = torch.rand with seed = size


def load_reference_module():
    utils = tvi.install()
    spec = importlib.util.spec_from_file_location("ref_fid_inception", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # `from torchvision import models`, `load_state_dict_from_url`
    return mod, utils


def main():
    torch.set_num_threads(8)
    mod, _ = load_reference_module()
    planted = {}

    def provider(url, progress=True):
        # fid_inception_v3() calls this right before inception.load_state_dict(...): hand it seeded weights for the
        # network it has just built (torchvision base + the reference's patched blocks)
        assert url == mod.FID_WEIGHTS_URL
        return planted["sd"]

    # build once to learn the key / shape listing, then plant weights drawn for exactly that listing
    mod.load_state_dict_from_url = lambda url, progress=True: {}
    orig_load = torch.nn.Module.load_state_dict
    torch.nn.Module.load_state_dict = lambda self, sd, strict=True: None        # (first build: no weights yet)
    try:
        skeleton = mod.fid_inception_v3()
    finally:
        torch.nn.Module.load_state_dict = orig_load
    planted["sd"] = seeded_inception_state(skeleton, SEED)
    mod.load_state_dict_from_url = provider
    net = mod.InceptionV3([mod.InceptionV3.BLOCK_INDEX_BY_DIM[2048]]).eval()   # callback_inception_metrics.py:210-211
    assert not any(p.requires_grad for p in net.parameters())

    inner = mod.fid_inception_v3()
    listing = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in inner.state_dict().items()]
    blob = {"seed": np.int64(SEED)}
    with torch.no_grad():
        for size, n in CASES:
            x = torch.rand(n, 3, size, size, generator=torch.Generator().manual_seed(size))
            logits, feats = net(x)
            assert len(feats) == 1 and feats[0].shape == (n, 2048, 1, 1)
            blob["pool3/%d" % size] = feats[0].reshape(n, 2048).numpy()
            blob["logits/%d" % size] = logits.numpy()
    np.savez_compressed(os.path.join(HERE, "inception_ref.npz"), **blob)
    with open(os.path.join(HERE, "inception_state_keys.json"), "w") as f:
        json.dump({"source": "state_dict of fid_inception_v3() built by the reference's code over the torchvision "
                             "0.10.0 layer definitions (oracle/torchvision_inception.py); == the keys of "
                             "pt_inception-2015-12-05-6726825d.pth up to BatchNorm's num_batches_tracked entries",
                   "weights_url": mod.FID_WEIGHTS_URL, "keys": listing}, f, indent=0)
    n_params = sum(int(np.prod(s)) for k, s, _ in listing if not k.endswith("num_batches_tracked"))
    print("inception_ref.npz: %d cases, %d state entries, %d numbers; pool3 |max| %.3f"
          % (len(CASES), len(listing), n_params, max(float(np.abs(v).max()) for k, v in blob.items() if k.startswith("pool3"))))
    tvi.uninstall()


if __name__ == "__main__":
    main()

"""north_star: index tensors bit-exact vs the reference.  tests/golden/hologan_indices.npz holds the int64 voxel
indices ``idx_a ... idx_h`` the UNMODIFIED reference's trilinear resampler produced (hologan_generator.py:245-288,
captured by tests/golden/make_index_golden.py) together with its inverse view matrices; the oracle (CPU) and the
HIP kernel (GPU) must reproduce every one of the 163,840 integers -- ``array_equal``, no float link."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR

NAMES = "abcdefgh"


def load():
    b = np.load(os.path.join(GOLDEN_DIR, "hologan_indices.npz"))
    n, c, s = (int(v) for v in b["shape"])
    vox = torch.randn(n, c, s, s, s, generator=torch.Generator().manual_seed(int(b["vox_seed"])))
    return b, vox


def test_oracle_indices_equal_the_reference_indices():
    from oracle import hologan_cpu as H
    b, vox = load()
    minv = torch.from_numpy(b["minv"])
    # the oracle's own host matrices agree with the reference's up to the host's LAPACK rounding ...
    own = H.view_matrices(b["view"])
    assert torch.allclose(own, minv, rtol=1e-5, atol=1e-5)
    # ... and from the reference's matrices the indices are the reference's, all of them
    x, y, z = H.resample_coords(minv)
    idx, _ = H.trilinear_indices(vox.shape, x, y, z)
    for name, got in zip(NAMES, idx):
        ref = b["idx_" + name]
        assert got.dtype == torch.int64 and ref.dtype == np.int64
        assert np.array_equal(got.numpy(), ref), "idx_%s: %d entries differ" % (name, int((got.numpy() != ref).sum()))
    # clamped corners are present (the fixture would be a weak one without them)
    s = vox.shape[-1]
    assert sum(int(((b["idx_" + k] % s) == s - 1).sum()) for k in NAMES) > 1000
    # values: the resampled volume itself, oracle vs reference
    n, c = vox.shape[:2]
    flat = vox.permute(0, 2, 3, 4, 1).reshape(-1, c)
    _, wts = H.trilinear_indices(vox.shape, x, y, z)
    out = sum(w.unsqueeze(1) * flat[i] for i, w in zip(idx, wts))
    out = out.reshape(n, s, s, s, c).permute(0, 4, 1, 2, 3)
    assert float((out - torch.from_numpy(b["out"])).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_hip_indices_equal_the_reference_indices():
    from lightning_gan_zoo_amd import functional as F
    b, vox = load()
    n = vox.shape[0]
    minv = torch.from_numpy(b["minv"]).reshape(n, 16).contiguous().cuda()
    out, idx = F.rigid_resample_indices(vox.cuda(), minv)
    assert idx.dtype == torch.int64
    idx = idx.cpu().numpy()
    for k, name in enumerate(NAMES):
        ref = b["idx_" + name]
        assert np.array_equal(idx[k], ref), "idx_%s: %d entries differ" % (name, int((idx[k] != ref).sum()))
    # the projected feature map is the reference's volume with its two middle axes swapped, one mirrored, folded
    ref = torch.from_numpy(b["out"]).permute(0, 1, 3, 2, 4).flip(2).reshape(n, -1, 16, 16)
    assert float((out.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())

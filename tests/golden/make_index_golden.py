#!/usr/bin/env python3
"""tests/golden/hologan_indices.npz: the int64 voxel indices ``idx_a ... idx_h`` of the UNMODIFIED reference's
trilinear resampler (core/models/hologan_generator.py:245-288) for a fixed set of views, captured while its own
``transformation3d`` runs (a TorchDispatchMode that observes ``aten.index.Tensor`` and the 4x4 inverse; the
reference is not modified).  north_star: index tensors are bit-exact vs the reference -- the oracle
(oracle/hologan_cpu.py) and the HIP kernel (csrc/gz_resample.hip) are compared with these arrays with
``array_equal``, not through resampled float values.

Build container only:   python tests/golden/make_index_golden.py

The fixture also stores the reference's inverse view matrices: floor() of a source coordinate is what makes an
index, the coordinates come from those matrices, and their last bit depends on the host's LAPACK (the build
container's Xeon and the GPU box's EPYC differ).  The views are chosen (seed search) so that every non-integer source
coordinate is at least MARGIN away from an integer: then a last-bit difference in `matmul(inverse, grid)` cannot
move an index, and the comparison is about the algorithm (clamping, corner order, flat layout), which is the point.
View 0 is the identity (all coordinates exactly integral, exactly representable: floor is exact on every host).
"""
import os
import sys

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_import                        # noqa: E402
from lightning_gan_zoo_amd.config import make_cfg    # noqa: E402

MARGIN = 2e-5
N, C, S = 5, 3, 16


class Capture(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.index_calls, self.inverses = [], []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith("aten.index.Tensor"):
            (idx,) = args[1]
            self.index_calls.append(idx.clone())
        if "inv" in name and torch.is_tensor(out if not isinstance(out, tuple) else out[0]):
            t = out if not isinstance(out, tuple) else out[0]
            if t.shape[-2:] == (4, 4) and t.dtype == torch.float32:
                self.inverses.append((name, t.clone()))
        return out


def views(seed):
    rng = np.random.RandomState(seed)
    v = np.zeros((N, 6))
    v[:, 0] = (rng.randint(220, 320, N) + rng.rand(N)) * np.pi / 180.0     # conf/expt/hologan.yaml's ranges,
    v[:, 1] = (rng.randint(70, 110, N) + rng.rand(N)) * np.pi / 180.0      # off the integer-degree lattice
    v[:, 2] = 1.0
    # every rotated view is also shifted by a fraction of a voxel: a rotation about the volume centre maps the
    # lattice point (8, 8, 8) onto itself up to rounding, i.e. onto 8 +- 1 ulp, and which side of 8 it lands on is
    # the host's last bit (the reference's own index for that voxel differs between hosts; its value does not)
    v[:, 3:] = rng.rand(N, 3) * 0.8 + 0.1
    v[0] = (0.0, 0.0, 1.0, 0.0, 0.0, 0.0)           # identity
    v[1, 3:] = (0.7, -1.2, 2.4)                     # shifted: part of the lattice leaves the volume (clamped corners)
    v[3, 2] = 2.0                                   # zoom in
    v[4, 2] = 0.6                                   # zoom out: most of the lattice is outside
    return v


def run(gen, vox, view):
    with Capture() as cap:
        out = gen.transformation3d(vox, view, S, S)
    assert len(cap.index_calls) == 8, len(cap.index_calls)
    minv = cap.inverses[-1][1]
    return out, cap.index_calls, minv


def main():
    ns = ref_import.load_reference()
    va = make_cfg("hologan", features=8, batch_size=N, noise_dim=16).generator.view_args
    torch.manual_seed(0)
    gen = ns.hologan_generator.Generator(8, 3, 16, va, 64, gpu=False)
    g = torch.Generator().manual_seed(17)
    vox = torch.randn(N, C, S, S, S, generator=g)
    best = None
    for seed in range(200):
        view = views(seed)
        out, idx, minv = run(gen, vox, view)
        grid = gen.meshgrid(S, S, S).reshape(1, 4, -1).repeat(N, 1, 1)
        pts = torch.matmul(minv, grid)[:, :3, :]
        frac = (pts - pts.round()).abs()
        generic = frac[frac > 0]                     # exact integers (identity view) are safe on every host
        margin = float(generic.min())
        if best is None or margin > best[0]:
            best = (margin, seed)
        if margin >= MARGIN:
            break
    margin, seed = best
    view = views(seed)
    out, idx, minv = run(gen, vox, view)
    assert margin >= MARGIN, (margin, seed)
    blob = {"view": view, "minv": minv.numpy(), "seed": np.int64(seed), "margin": np.float64(margin),
            "vox_seed": np.int64(17), "shape": np.asarray([N, C, S], dtype=np.int64),
            "out": out.detach().numpy()}
    for name, t in zip("abcdefgh", idx):
        assert t.dtype == torch.int64
        blob["idx_" + name] = t.numpy()
    path = os.path.join(HERE, "hologan_indices.npz")
    np.savez_compressed(path, **blob)
    clamped = sum(int(((t % S == 0) | (t % S == S - 1)).sum()) for t in idx)
    print(f"{path}: seed {seed}, margin {margin:.2e}, {8 * idx[0].numel()} indices ({clamped} on an x face), "
          f"{os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

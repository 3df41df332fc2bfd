"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/gz_ops.h declares; the product path refuses CPU tensors instead of falling back."""
import ctypes
import os

import pytest
import torch

from lightning_gan_zoo_amd import _lib


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 20
    assert os.path.exists(_lib.LIB_PATH), "build with python -m lightning_gan_zoo_amd.build"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, missing
    _lib.lib.load()
    assert b"gfx950" in _lib.lib.gz_build_info()


def test_size_queries_need_no_gpu():
    lib = _lib.lib
    assert lib.gz_conv2d_pack_fwd_elems(6, 3, 4, 4) == 3 * 16 * 8          # K padded to a multiple of 4
    assert lib.gz_conv2d_pack_dgrad_elems(6, 3, 4, 4, 2) == 4 * 6 * 4 * 4  # 4 phases x K*2*2 x round4(C)
    assert lib.gz_norm_workspace_bytes(4, 8) >= 4 * 8 * 8
    assert lib.gz_norm_coef_elems(4, 8, 1) == 32 and lib.gz_norm_coef_elems(4, 8, 0) == 128


def test_product_path_fails_loudly_on_cpu_tensors():
    from lightning_gan_zoo_amd.core.models.standard_networks import Discriminator, Generator
    g, d = Generator(16, 3, 8), Discriminator(3, 8, final_sigmoid=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.randn(2, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d(torch.randn(2, 3, 64, 64))


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; import lightning_gan_zoo_amd.core.lightning_module, lightning_gan_zoo_amd.functional, "
            "lightning_gan_zoo_amd.harness; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """build.py stamps the library with a content digest of the kernel sources / headers / flags (no mtimes); the
    loader compares it with the tree it runs from, so a snapshot carrying an old .so next to newer sources cannot
    run stale kernels silently."""
    from lightning_gan_zoo_amd import build
    dll = ctypes.CDLL(_lib.LIB_PATH)
    dll.gz_source_digest.restype = ctypes.c_char_p
    assert dll.gz_source_digest().decode() == build.source_digest(), "rebuild: python -m lightning_gan_zoo_amd.build"
    real = build._read
    target = os.path.join(build.CSRC, "gz_common.h")
    monkeypatch.setattr(build, "_read", lambda p: real(p) + (b"// edited\n" if p == target else b""))
    assert build.source_digest() != dll.gz_source_digest().decode()
    monkeypatch.delenv("GZ_LIB", raising=False)
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib._check_digest(dll)
    monkeypatch.setenv("GZ_ALLOW_STALE", "1")
    with pytest.warns(UserWarning, match="built from other sources"):
        _lib._check_digest(dll)

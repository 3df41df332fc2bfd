"""Build recipe for libgz_hip.so (the C-ABI kernel library), in-tree, for gfx950 only.

    python -m lightning_gan_zoo_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to
the GPU box with the repo snapshot.

Staleness is decided by CONTENT, not by mtime (a snapshot or checkout can reset mtimes): every object carries a
sidecar ``<obj>.sha`` = sha256 of its source, every header and the flags it was compiled with, and the library
embeds ``source_digest()`` -- sha256 over all kernel sources, headers and flags -- as ``gz_source_digest()``.
``_lib.load`` compares that with the digest of the tree it runs from and refuses a library built from other sources.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(CSRC, "libgz_hip.so")
ARCH = "gfx950"
DIGEST_SOURCE = "gz_build_id.hip"          # holds gz_source_digest(); compiled with -DGZ_SOURCE_DIGEST=...
SOURCES = ["gz_conv.hip", "gz_conv3d.hip", "gz_norm.hip", "gz_misc.hip", "gz_resample.hip", "gz_optim.hip", "gz_resnet.hip", "gz_loss.hip", "gz_infer.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _headers():
    out = []
    for d in (CSRC, INCLUDE):
        out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".h")]
    return out


def _sha(parts):
    h = hashlib.sha256()
    for p in parts:
        b = p if isinstance(p, bytes) else str(p).encode()
        h.update(len(b).to_bytes(8, "little"))
        h.update(b)
    return h.hexdigest()


def source_digest(extra_flags=()):
    """sha256 over every kernel source, every header (csrc/*.h, include/*.h) and the compile flags, by content."""
    files = [os.path.join(CSRC, s) for s in SOURCES + [DIGEST_SOURCE]] + _headers()
    parts = []
    for f in files:
        parts += [os.path.basename(f), _read(f)]
    flags = [f for f in FLAGS if f != INCLUDE] + list(extra_flags)
    return "g" + _sha(parts + flags)[:31]      # an identifier: travels through hipcc's -D quoting unharmed


def build(force=False, verbose=True, extra_flags=None, suffix=""):
    """``extra_flags`` / ``suffix`` build an experimental variant (libgz_hip<suffix>.so, own objects);
    select it at run time with GZ_LIB=<path>."""
    hipcc = _hipcc()
    extra_flags = list(extra_flags or [])
    lib_path = LIB_PATH.replace(".so", suffix + ".so")
    digest = source_digest(extra_flags)
    hdr = [_read(h) for h in _headers()]
    flags = [f for f in FLAGS if f != INCLUDE] + extra_flags
    jobs, objs, stamps = [], [], []
    for s in SOURCES + [DIGEST_SOURCE]:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", suffix + ".o"))
        objs.append(obj)
        define = ['-DGZ_SOURCE_DIGEST=%s' % digest] if s == DIGEST_SOURCE else []
        want = _sha([_read(src)] + hdr + flags + define)
        have = _read(obj + ".sha").decode() if os.path.exists(obj + ".sha") and os.path.exists(obj) else None
        if force or have != want:
            jobs.append([hipcc] + FLAGS + extra_flags + define + ["-c", src, "-o", obj])
            stamps.append((obj + ".sha", want))

    def run(cmd):
        if verbose:
            print("[gz build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    if jobs:
        for path, _ in stamps:          # a half-finished build must not look fresh
            if os.path.exists(path):
                os.remove(path)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
        for path, want in stamps:
            with open(path, "w") as f:
                f.write(want)
    link_stamp = lib_path + ".sha"
    linked = _read(link_stamp).decode() if os.path.exists(link_stamp) and os.path.exists(lib_path) else None
    if jobs or linked != digest:
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs)
        with open(link_stamp, "w") as f:
            f.write(digest)
    return lib_path


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    sfx = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    print(build(force="--force" in sys.argv, extra_flags=flags, suffix=sfx))

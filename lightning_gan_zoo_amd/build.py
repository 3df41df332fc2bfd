"""Build recipe for libgz_hip.so (the C-ABI kernel library), in-tree, for gfx950 only.

    python -m lightning_gan_zoo_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to
the GPU box with the repo snapshot.  Objects are rebuilt only when a source or header
is newer (or with --force).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(CSRC, "libgz_hip.so")
ARCH = "gfx950"
SOURCES = ["gz_conv.hip", "gz_conv3d.hip", "gz_norm.hip", "gz_misc.hip", "gz_resample.hip", "gz_optim.hip", "gz_resnet.hip", "gz_loss.hip", "gz_infer.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_header():
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force=False, verbose=True, extra_flags=None, suffix=""):
    """``extra_flags`` / ``suffix`` build an experimental variant (libgz_hip<suffix>.so, own objects);
    select it at run time with GZ_LIB=<path>."""
    hipcc = _hipcc()
    extra_flags = list(extra_flags or [])
    lib_path = LIB_PATH.replace(".so", suffix + ".so")
    hdr_t = _newest_header()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", suffix + ".o"))
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
        if stale:
            jobs.append([hipcc] + FLAGS + extra_flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[gz build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    need_link = bool(jobs) or not os.path.exists(lib_path) or any(
        os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs)
    if need_link:
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    sfx = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    print(build(force="--force" in sys.argv, extra_flags=flags, suffix=sfx))

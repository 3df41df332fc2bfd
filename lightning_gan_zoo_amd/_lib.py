"""ctypes binding of libgz_hip.so (C ABI declared in include/gz_ops.h).

The product path has no CPU or eager fallback: if the library cannot be loaded, or a
kernel reports an error, a RuntimeError is raised.
"""
import ctypes
import os
import re
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GZ_LIB") or os.path.join(HERE, "csrc", "libgz_hip.so")   # GZ_LIB: experimental builds
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "gz_ops.h")

ERRORS = {-1: "bad shape / alignment", -2: "unsupported configuration", -3: "workspace too small",
          -4: "HIP error", -5: "tensor too large for 32-bit buffer addressing"}

_C = {
    "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "long long": ctypes.c_longlong, "hipStream_t": ctypes.c_void_p, "void": None,
    "const char*": ctypes.c_char_p,
}


def _ctype(decl):
    decl = decl.strip()
    if decl.endswith("*") and decl != "const char*":
        return ctypes.c_void_p
    return _C[decl]


def parse_header(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every function declared in gz_ops.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    text = text.replace('extern "C" {', "")
    protos = {}
    for m in re.finditer(r"((?:const\s+)?(?:long long|size_t|int|float|void|char)\s*\*?)\s*\b(gz_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                tm = re.match(r"(.*?[\*\s])\s*\w+$", a)
                t = tm.group(1).strip().replace(" *", "*")
                t = re.sub(r"\s+", " ", t)
                if t.endswith("*"):
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_C[t])
        ret = re.sub(r"\s+", " ", ret).replace(" *", "*")
        protos[name] = (_ctype(ret), argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._lock = threading.Lock()

    def load(self):
        with self._lock:
            if self._dll is not None:
                return self._dll
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "lightning_gan_zoo_amd: HIP kernel library %s is missing; build it with "
                    "`python -m lightning_gan_zoo_amd.build` (there is no CPU fallback)" % LIB_PATH)
            # torch ships its own libamdhip64 / libhsa-runtime64; load it first so that this library
            # binds to the SAME HIP runtime instance (otherwise the process ends up with two runtimes
            # and ours reports "no ROCm-capable device")
            import torch  # noqa: F401
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in parse_header().items():
                fn = getattr(dll, name)   # AttributeError if the library does not export it
                fn.restype = restype
                fn.argtypes = argtypes
            _check_digest(dll)
            self._dll = dll
            return dll

    def __getattr__(self, name):
        return getattr(self.load(), name)


def _check_digest(dll):
    """The library must have been built from the kernel sources lying next to it: a snapshot that carries an old
    .so beside newer sources would otherwise run kernels the tests are not about.  GZ_LIB builds (experimental
    variants with extra -D flags, tools/) are exempt; GZ_ALLOW_STALE=1 turns the refusal into a warning."""
    if os.environ.get("GZ_LIB"):
        return
    from .build import source_digest
    built, tree = dll.gz_source_digest().decode(), source_digest()
    if built != tree:
        msg = ("lightning_gan_zoo_amd: %s was built from other sources (library digest %s, tree digest %s); "
               "rebuild with `python -m lightning_gan_zoo_amd.build`" % (LIB_PATH, built, tree))
        if os.environ.get("GZ_ALLOW_STALE") == "1":
            import warnings
            warnings.warn(msg)
        else:
            raise RuntimeError(msg)


lib = _Lib()


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -4:
            try:
                detail = ": " + lib.gz_last_error().decode()
            except Exception:  # noqa: BLE001
                pass
        raise RuntimeError("libgz_hip: %s failed with code %d (%s%s)" % (what, rc, ERRORS.get(rc, "?"), detail))

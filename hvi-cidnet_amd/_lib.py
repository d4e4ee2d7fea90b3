"""ctypes binding of libcidnet_hip.so.  Signatures are parsed from include/cidnet_hip.h, so the
header is the single source of truth for the C ABI.  There is no fallback: if the library is
missing or a call fails, this raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "cidnet_hip.h")
LIB_PATH = os.environ.get("CIDNET_LIB_PATH") or os.path.join(_HERE, "libcidnet_hip.so")   # override: A/B builds (dev)

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "void": None}
_PROTO = re.compile(r"^\s*(int|long|void)\s+(cidnet_\w+)\s*\(([^;{]*?)\)\s*;", re.M | re.S)

ERR_NAMES = {-1: "CIDNET_ERR_ARG (null pointer / non-positive size)", -2: "CIDNET_ERR_SHAPE (unsupported shape)",
             -3: "CIDNET_ERR_WS (workspace too small)"}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [(argtype, argname), ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"#ifdef CIDNET_DEBUG.*?#endif", "", text, flags=re.S)    # timing-study switches: debug builds only
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for ret, name, args in _PROTO.findall(text):
        sig = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    sig.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    sig.append((_CTYPES[ty.replace("const", "").strip()], nm))
        out[name] = (_CTYPES[ret], sig)
    return out


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP library is the only compute path of this package. "
                "Build it with `python hvi-cidnet_amd/build.py` (hipcc cross-compiles gfx950 without a GPU).")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, sig) in self.protos.items():
            fn = getattr(self._dll, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for t, _ in sig]
        v = self._dll.cidnet_abi_version()
        hv = int(re.search(r"#define\s+CIDNET_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
        if v != hv:
            raise ImportError(f"libcidnet_hip.so ABI version {v} != header version {hv}: rebuild the library")

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        """Invoke an `int cidnet_*` entry point; raise RuntimeError on a non-zero status."""
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            what = ERR_NAMES.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
            raise RuntimeError(f"{name} failed: {what}")


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

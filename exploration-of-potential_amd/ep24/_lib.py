"""ctypes binding of libep24.so (the C ABI declared in include/ep24.h).

The product path has no CPU fallback: if the library cannot be loaded, or a kernel entry point is called
without a GPU, this module raises.  Signatures are parsed from the header so the binding, the header and
the exported symbols cannot drift apart (tests/test_abi.py checks all three).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EP24_LIB") or os.path.join(_HERE, "libep24.so")     # EP24_LIB: a diagnostic build (make stamps)
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "ep24.h"))

_CTYPES = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "double": ctypes.c_double,
}


class Ep24Error(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [(argtype_str, argname), ...])} for every `ep24_*` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(ep24_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        args = " ".join(args.split())
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, params)
    return protos


def _to_ctype(tstr):
    if "*" in tstr:
        return ctypes.c_void_p
    base = tstr.replace("const", "").strip()
    return _CTYPES[base]


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise Ep24Error(
                "ep24: %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C exploration-of-potential_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (ret, params) in self.protos.items():
            f = getattr(self.cdll, name)          # AttributeError here = header/library drift
            f.restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
            f.argtypes = [_to_ctype(t) for t, _ in params]
            self.fn[name] = f
        # a library built from another header must not be called into (arguments would be mis-passed silently)
        m = re.search(r"#define\s+EP24_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read())
        want, got = (int(m.group(1)) if m else None), self.fn["ep24_abi_version"]()
        if want is None or got != want:
            raise Ep24Error("ep24: %s reports ABI version %s, include/ep24.h declares %s - rebuild the library "
                            "(make -C exploration-of-potential_amd/csrc)" % (LIB_PATH, got, want))

    def last_error(self):
        return self.fn["ep24_last_error"]().decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t, offset_elems=0):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke ep24_<name>; raises Ep24Error with the library's message on a non-zero return code."""
    L = lib()
    rc = L.fn["ep24_" + name](*args)
    if rc != 0:
        raise Ep24Error("ep24_%s failed (%d): %s" % (name, rc, L.last_error()))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise Ep24Error("ep24: no GPU visible - the hot path runs only as HIP kernels on gfx950 (no CPU fallback)")
    q = os.environ.get("GPU_MAX_HW_QUEUES", "")
    if q.isdigit() and int(q) > 4:
        # measured in round 5: with 5 or more hardware queues per process the two lanes of the captured step no
        # longer run side by side - 587 instead of 978 images/s on one box; 2, 3 and 4 (the runtime's default) are equal
        import warnings
        warnings.warn("ep24: GPU_MAX_HW_QUEUES=%s - the captured training step loses ~40 %% with more than 4 hardware queues" % q)

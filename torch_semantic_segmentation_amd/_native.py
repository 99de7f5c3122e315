"""ctypes binding of libtss_hip.so (include/tss_hip.h).

The product path has NO fallback: if the library is missing or cannot be loaded, importing the ops raises.
Build it with ``python -m torch_semantic_segmentation_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TSS_HIP_LIB') or os.path.join(_HERE, 'libtss_hip.so')   # TSS_HIP_LIB: A/B builds (tools/ab_variants.sh)

TSS_F32, TSS_BF16 = 0, 1
ERRORS = {-1: 'TSS_ERR_DTYPE (unsupported dtype)', -2: 'TSS_ERR_SHAPE (unsupported shape or pitch)',
          -3: 'TSS_ERR_ALIGN (pointer not 16-byte aligned)', -4: 'TSS_ERR_HIP (HIP launch error)'}

_P, _L, _I, _F, _D = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_double

HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'tss_hip.h')


def parse_header(path=HEADER_PATH):
    """{function name: [ctypes argument types]} for every `int tss_*(...)` declared in include/tss_hip.h.

    The binding is generated from the header so the two cannot drift apart; pointer parameters of any
    pointee type map to c_void_p (callers pass tensor.data_ptr() or None).
    """
    import re
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    decls = {}
    for m in re.finditer(r'\bint\s+(tss_\w+)\s*\(([^)]*)\)\s*;', text):
        name, params = m.group(1), m.group(2).strip()
        types = []
        if params and params != 'void':
            for prm in params.split(','):
                prm = prm.strip()
                if '*' in prm:
                    types.append(_P)
                else:
                    words = prm.split()[:-1]  # drop the parameter name
                    base = ' '.join(w for w in words if w != 'const')
                    types.append({'int': _I, 'long': _L, 'float': _F, 'double': _D}[base])
        decls[name] = types
    return decls


_lib = None


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libtss_hip.so is not built (%s). Run `python -m torch_semantic_segmentation_amd.build`; '
                'this package has no non-HIP fallback.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in parse_header().items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        handle.tss_last_error.restype = ctypes.c_char_p
        handle.tss_arch.restype = ctypes.c_char_p
        handle.tss_prof_name.restype = ctypes.c_char_p
        handle.tss_prof_name.argtypes = [ctypes.c_int]
        handle.tss_prof_symbol.restype = ctypes.c_char_p
        handle.tss_prof_symbol.argtypes = [ctypes.c_int]
        handle.tss_prof_get.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        handle.tss_prof_records.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_long]
        handle.tss_prof_records.restype = ctypes.c_long
        handle.tss_pwconv_bwd_weight_ws.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        handle.tss_pwconv_bwd_weight_ws.restype = ctypes.c_long
        handle.tss_upsample_ce_ws.argtypes = [ctypes.c_int] * 6
        handle.tss_upsample_ce_ws.restype = ctypes.c_long
        handle.tss_ohem_workspace_bytes.argtypes = []
        handle.tss_ohem_workspace_bytes.restype = ctypes.c_long
        handle.tss_bn_xchg_bytes.argtypes = []
        handle.tss_bn_xchg_bytes.restype = ctypes.c_long
        _lib = handle
    return _lib


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        detail = lib().tss_last_error().decode() if rc == -4 else ''
        raise RuntimeError('%s failed: %s %s' % (name, ERRORS.get(rc, rc), detail))


def fast_paths_disabled():
    """tss_set_option(TSS_OPT_DISABLE_FAST_PATHS, 1) is in force (parity tests of the generic kernels)."""
    return lib().tss_get_option(1) == 1


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_slabs = None


def stat_slabs():
    global _slabs
    if _slabs is None:
        _slabs = lib().tss_stat_slabs()
    return _slabs


def dtype_code(dtype):
    if dtype == torch.float32:
        return TSS_F32
    if dtype == torch.bfloat16:
        return TSS_BF16
    raise TypeError('the HIP path computes in float32 or bfloat16 activations, got %s' % dtype)


# ----------------------------------------------------------------------------- profiler helpers

K_COUNT = 33


def prof_enable(on=True):
    call('tss_prof_enable', 1 if on else 0)


def prof_reset():
    call('tss_prof_reset')


def prof_records(limit=100000):
    """Per-launch (op_name, ms, algorithmic bytes) in launch order (after prof_table() collected them)."""
    h = lib()
    ids = (ctypes.c_int * limit)()
    ms = (ctypes.c_double * limit)()
    by = (ctypes.c_double * limit)()
    n = min(h.tss_prof_records(ids, ms, by, limit), limit)
    return [(h.tss_prof_name(ids[i]).decode(), ms[i], by[i]) for i in range(n)]


def prof_table():
    """Collect recorded launches; returns {op_name: dict(symbol, launches, ms, bytes, flops)}."""
    call('tss_prof_collect')
    out = {}
    h = lib()
    for k in range(K_COUNT):
        n, ms, by, fl = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        h.tss_prof_get(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by), ctypes.byref(fl))
        if n.value:
            out[h.tss_prof_name(k).decode()] = dict(symbol=h.tss_prof_symbol(k).decode(), launches=n.value,
                                                    ms=ms.value, bytes=by.value, flops=fl.value)
    return out

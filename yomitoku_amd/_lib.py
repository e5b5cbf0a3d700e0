"""ctypes binding of libymk_hip.so (the C ABI declared in include/ymk.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C yomitoku_amd/csrc``.
There is no CPU fallback: if the shared object is missing or a call fails, this module raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libymk_hip.so")

# name -> (restype, argtypes); kept in one table so tests can check every symbol of ymk.h exists
SIGNATURES = {
    "ymk_version": (c_int, []),
    "ymk_last_error": (c_char_p, []),
    "ymk_device_count": (c_int, []),
    "ymk_model_create": (c_void_p, [c_char_p, c_int]),
    "ymk_model_destroy": (None, [c_void_p]),
    "ymk_model_set_param": (c_int, [c_void_p, c_char_p, c_double]),
    "ymk_model_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_int, POINTER(c_int64)]),
    "ymk_model_finalize": (c_int, [c_void_p]),
    "ymk_model_reserve": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "ymk_model_weight_bytes": (c_int64, [c_void_p]),
    "ymk_model_workspace_bytes": (c_int64, [c_void_p]),
    "ymk_dbnet_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ymk_parseq_dims": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "ymk_parseq_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, POINTER(c_int), POINTER(c_int), c_void_p]),
    "ymk_parseq_forward_groups": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), c_int, c_void_p, POINTER(c_int),
                                          POINTER(c_int), c_void_p]),
    "ymk_parseq_token_stats": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ymk_rtdetr_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ymk_det_preprocess": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ymk_pil_resize_to_chw": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
         c_void_p],
    ),
    "ymk_pil_resize_batch_to_chw": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ymk_pil_batch_record_words": (c_int, []),
    "ymk_crop_batch": (
        c_int,
        [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "ymk_crop_batch_levels": (
        c_int,
        [c_void_p, POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
         c_void_p],
    ),
    "ymk_halve_u8c3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ymk_crop_desc_size": (c_int, []),
    "ymk_db_postprocess": (
        c_int,
        [c_void_p, c_int, c_int, c_float, c_float, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_int,
         POINTER(c_int)],
    ),
    "ymk_table_hole_rects": (c_int, [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, POINTER(c_int)]),
    "ymk_debug_option": (c_int, [c_char_p, c_int]),
    "ymk_stat": (c_int, [c_char_p, POINTER(c_int64)]),
    "ymk_op_vit_mlp": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_float), c_void_p],
    ),
    "ymk_amax_check_counters": (c_int, [POINTER(c_int64)]),
    "ymk_prof_begin": (c_int, []),
    "ymk_prof_end": (c_int, [POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "ymk_prof_bytes": (c_int, [POINTER(c_double)]),
    "ymk_prof_launch_table": (c_int, [POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), c_int64, POINTER(c_int64)]),
    "ymk_op_conv2d": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
         c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "ymk_op_conv1x1_astat": (
        c_int,
        [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int,
         POINTER(c_float), c_void_p],
    ),
    "ymk_op_layernorm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "ymk_op_attention": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
         c_void_p],
    ),
    "ymk_op_maxpool3x3s2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ymk_op_upsample_bilinear": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
}

_lib = None


class YmkError(RuntimeError):
    pass


def load():
    """Load libymk_hip.so once; raises YmkError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YmkError(
            f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C yomitoku_amd/csrc` (there is no CPU fallback)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # evaluation switches for a whole process (include/ymk.h: ymk_debug_option): YMK_CONV_SPLIT=2 is short for
    # YMK_DEBUG_OPTIONS="conv_split=2"; several options are comma separated
    spec = os.environ.get("YMK_DEBUG_OPTIONS", "")
    if os.environ.get("YMK_CONV_SPLIT"):
        spec += ",conv_split=" + os.environ["YMK_CONV_SPLIT"]
    for item in filter(None, (x.strip() for x in spec.split(","))):
        key, _, value = item.partition("=")
        if lib.ymk_debug_option(key.strip().encode(), int(value)) != 0:
            raise YmkError(f"YMK_DEBUG_OPTIONS ({item}): " + lib.ymk_last_error().decode("utf-8", "replace"))
    return lib


def check(status: int, what: str = "ymk call"):
    if status != 0:
        msg = load().ymk_last_error()
        raise YmkError(f"{what} failed: {msg.decode('utf-8', 'replace') if msg else status}")


CONV_FAST_DEFAULT = 27  # ymk_debug_option("conv_fast"): the library's default bits (include/ymk.h)


def stat(key: str) -> int:
    """A launch counter of the library (include/ymk.h: ymk_stat)."""
    v = c_int64()
    check(load().ymk_stat(key.encode(), ctypes.byref(v)), f"ymk_stat({key})")
    return int(v.value)


def debug_option(key: str, value: int):
    """Test / measurement knob of the library (include/ymk.h: ymk_debug_option)."""
    check(load().ymk_debug_option(key.encode(), int(value)), f"ymk_debug_option({key})")


def amax_check_counters():
    """(launches checked, records below the true max|x|, records > 2^8 above it, largest exponent distance) of the
    ymk_debug_option("amax_check", 1) self-check (include/ymk.h)."""
    import ctypes

    out = (ctypes.c_int64 * 4)()
    check(load().ymk_amax_check_counters(out), "ymk_amax_check_counters")
    return tuple(int(v) for v in out)


def prof_launch_table():
    """[(ms, flop, bytes, mfma_products)] of the launches between the last ymk_prof_begin / ymk_prof_end (include/ymk.h)."""
    import ctypes

    lib, n = load(), ctypes.c_int64()
    check(lib.ymk_prof_launch_table(None, None, None, None, 0, ctypes.byref(n)), "ymk_prof_launch_table")
    cols = [(ctypes.c_double * max(1, n.value))() for _ in range(4)]
    check(lib.ymk_prof_launch_table(*cols, n.value, ctypes.byref(n)), "ymk_prof_launch_table")
    return [tuple(float(c[i]) for c in cols) for i in range(n.value)]


def ptr(t):
    """Device/host address of a torch tensor (must be contiguous) or None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def current_stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream


def dims_array(shape):
    arr = (c_int64 * max(1, len(shape)))(*[int(s) for s in shape])
    return arr

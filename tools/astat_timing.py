"""The A-stationary kernel against the register-staged fp16-split kernel ("conv_split_tile" 3: what these layers ran on in round 4) on the short-K layer shapes of the analyzer
(rows as in profiles/r04_conv_two_roof_by_layer.md).  The library's time is the profiled launch span of ymk_op_conv2d under
conv_split = 16 (its max|x| pass is inside the span, as in tools/conv_sweep.py); the candidate's is the launch alone - the
line says both and the size of the max|x| pass, so that the comparison can be read either way."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tests import hipops
from yomitoku_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [("parseq fc1 192->768 gelu", 342624, 192, 768, "gelu", False), ("parseq qkv 192->576", 342624, 192, 576, "none", False),
          ("parseq proj 192->192 +res", 342624, 192, 192, "none", True), ("dbnet l1 64->256 +res", 947200, 64, 256, "relu", True),
          ("dbnet l2 128->512 +res", 236800, 128, 512, "relu", True), ("dbnet dec 256->256", 947200, 256, 256, "none", False)]
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if any(tok in s[0] for tok in os.environ["ONLY"].split("|"))]
out = []
for name, m, c, cout, act, use_res in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(m, c, generator=g, device=dev)
    w = (torch.randn(cout, c, generator=g, device=dev) / c ** 0.5).cpu()
    b = torch.randn(cout, generator=g, device=dev).cpu()
    res = torch.randn(m, cout, generator=g, device=dev) if use_res else None
    y, ms = hipops.conv1x1_astat(x, w, None, b, res, act, reps=4)
    ms_ln = None
    if c in (128, 192) and not use_res:  # the same layer with the LayerNorm in front of it folded into the operand load
        gam, bet = torch.rand(c, generator=torch.Generator().manual_seed(2)) + 0.5, torch.randn(c, generator=torch.Generator().manual_seed(3)) * 0.1
        _, ms_ln = hipops.conv1x1_astat(x, w, None, b, None, act, reps=4, ln=(gam, bet, 1e-6))
    xn, rn = x.t().reshape(1, c, 1, m), (res.t().reshape(1, cout, 1, m) if use_res else None)
    times = []
    _lib.debug_option("conv_split", 16)
    _lib.debug_option("conv_split_tile", 3)
    for i in range(3):
        _lib.check(lib.ymk_prof_begin())
        y_lib = hipops.conv2d(xn, w.reshape(cout, c, 1, 1), None, b, rn, act=act)
        a, f, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(a), ctypes.byref(f), ctypes.byref(n)))
        times.append(a.value)
    _lib.debug_option("conv_split", -1)
    _lib.debug_option("conv_split_tile", 0)
    same = bool(torch.equal(y, y_lib.reshape(cout, m).t()))
    nbytes = 4.0 * (m * c + cout * c + m * cout * (2 if use_res else 1))
    row = {"shape": name, "rows": m, "astat_us": round(ms * 1e3, 1), "library_span_us": round(min(times[1:]) * 1e3, 1), "bit_identical": same,
           "astat_tbs": round(nbytes / (ms * 1e-3) / 1e12, 2), "astat_tflops": round(2.0 * m * c * cout / (ms * 1e-3) / 1e12, 1),
           "absmax_pass_mb": round(4.0 * m * c / 1e6, 1), "astat_with_layernorm_us": round(ms_ln * 1e3, 1) if ms_ln else None}
    print(json.dumps(row), flush=True)
    out.append(row)
    del x, res, y, y_lib
    torch.cuda.empty_cache()

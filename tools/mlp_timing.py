"""The fused ViT MLP (ymk_op_vit_mlp) at the analyzer's row count against the launches it replaces (k_layernorm-folded fc1 on the
A-stationary kernel + fc2 on the register-staged kernel, timed through ymk_op_conv1x1_astat / ymk_op_conv2d spans)."""
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
D, F = 192, 768
for m in (342624, 131072, 40960):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(m, D, generator=g) * 1.5).to(dev)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.2
    w1, b1 = torch.randn(F, D, generator=g) / D ** 0.5, torch.randn(F, generator=g) * 0.3
    w2, b2 = torch.randn(D, F, generator=g) / F ** 0.5, torch.randn(D, generator=g) * 0.3
    y, ms = hipops.vit_mlp(x, gamma, beta, 1e-6, w1, b1, w2, b2, reps=4)
    h, ms1 = hipops.conv1x1_astat(x, w1, None, b1, None, "gelu", reps=3, ln=(gamma, beta, 1e-6))
    _lib.debug_option("conv_split", 16)
    times = []
    for i in range(3):
        _lib.check(lib.ymk_prof_begin())
        hipops.conv2d(h.t().reshape(1, F, 1, m), w2.reshape(D, F, 1, 1), None, b2, x.t().reshape(1, D, 1, m))
        a, f, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(a), ctypes.byref(f), ctypes.byref(n)))
        times.append(a.value)
    _lib.debug_option("conv_split", -1)
    flops = 4.0 * m * D * F
    print(json.dumps({"rows": m, "fused_us": round(ms * 1e3, 1), "fc1_with_layernorm_us": round(ms1 * 1e3, 1), "fc2_span_us": round(min(times[1:]) * 1e3, 1),
                      "fused_tflops_equivalent": round(flops / (ms * 1e-3) / 1e12, 1)}), flush=True)
    del x, y, h
    torch.cuda.empty_cache()

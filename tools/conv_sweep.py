"""conv_igemm schedule experiments: the analyzer's heavy shapes x ymk_debug_option("conv_variant", v), timed by the
library's own per-launch HIP events (ymk_prof_*).  Prints TFLOP/s per (shape, variant); median of 5 launches."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from yomitoku_amd import _lib

dev = torch.device("cuda:0")
if os.environ.get("YMK_LIB"):  # another build of the library (tools/jobs: the ablation builds of one kernel)
    _lib.LIB_PATH = os.path.abspath(os.environ["YMK_LIB"])
lib = _lib.load()
# (name, n, h, w, cin, cout, k, stride, pad, dil, act, residual)
SHAPES = [
    ("dbnet l4 3x3 512->512 d2", 8, 100, 74, 512, 512, 3, 1, 2, 2, 1, 0),
    ("dbnet dec 3x3 256->64 400x296", 8, 400, 296, 256, 64, 3, 1, 1, 1, 1, 0),
    ("dbnet l1 3x3 64->64 400x296", 8, 400, 296, 64, 64, 3, 1, 1, 1, 1, 0),
    ("dbnet l1 1x1 256->64", 1, 1, 947200, 256, 64, 1, 1, 0, 1, 1, 0),
    ("dbnet l4 1x1 1024->2048", 1, 1, 59200, 1024, 2048, 1, 1, 0, 1, 1, 1),
    ("dbnet l4 1x1 512->2048", 1, 1, 59200, 512, 2048, 1, 1, 0, 1, 1, 1),
    ("dbnet l4 1x1 2048->512", 1, 1, 59200, 2048, 512, 1, 1, 0, 1, 1, 0),
    ("dbnet l3 3x3 256->256", 8, 100, 74, 256, 256, 3, 1, 1, 1, 1, 0),
    ("dbnet l2 3x3 128->128", 8, 200, 148, 128, 128, 3, 1, 1, 1, 1, 0),
    ("dbnet l1 1x1 64->256 +res", 1, 1, 947200, 64, 256, 1, 1, 0, 1, 1, 1),
    ("dbnet dec 1x1 256->256", 1, 1, 947200, 256, 256, 1, 1, 0, 1, 0, 0),
    ("dbnet l1 1x1 64->64", 1, 1, 947200, 64, 64, 1, 1, 0, 1, 1, 0),
    ("dbnet l2 1x1 128->512 +res", 1, 1, 236800, 128, 512, 1, 1, 0, 1, 1, 1),
    ("dbnet l3 1x1 256->1024 +res", 1, 1, 59200, 256, 1024, 1, 1, 0, 1, 1, 1),
    ("dbnet l3 1x1 1024->256", 1, 1, 59200, 1024, 256, 1, 1, 0, 1, 1, 0),
    ("rtdetr enc 3x3 256->256 80x80", 8, 80, 80, 256, 256, 3, 1, 1, 1, 2, 0),
    ("rtdetr bb 1x1 256->1024 40x40", 1, 1, 12800, 256, 1024, 1, 1, 0, 1, 1, 1),
    ("parseq qkv 192->576", 1, 1, 176496, 192, 576, 1, 1, 0, 1, 0, 0),
    ("parseq fc1 192->768 gelu", 1, 1, 176496, 192, 768, 1, 1, 0, 1, 4, 0),
    ("parseq fc2 768->192 +res", 1, 1, 176496, 768, 192, 1, 1, 0, 1, 0, 1),
    ("parseq proj 192->192 +res", 1, 1, 176496, 192, 192, 1, 1, 0, 1, 0, 1),
    ("parseq head 192->7119", 1, 1, 66155, 192, 7119, 1, 1, 0, 1, 0, 0),
    ("parseq AR head 655 rows", 1, 1, 655, 192, 7119, 1, 1, 0, 1, 0, 0),
    ("parseq AR head 200 rows", 1, 1, 200, 192, 7119, 1, 1, 0, 1, 0, 0),
    # the reference's default recogniser (parseq-large-v4_1: D = 768, 400 tokens per line), 128 lines per forward
    ("parseq-large qkv 768->2304", 1, 1, 51200, 768, 2304, 1, 1, 0, 1, 0, 0),
    ("parseq-large proj 768->768 +res", 1, 1, 51200, 768, 768, 1, 1, 0, 1, 0, 1),
    ("parseq-large fc1 768->3072 gelu", 1, 1, 51200, 768, 3072, 1, 1, 0, 1, 4, 0),
    ("parseq-large fc2 3072->768 +res", 1, 1, 51200, 3072, 768, 1, 1, 0, 1, 0, 1),
]
# "v", "v/f" or "v/f/key=val;key=val": conv_variant v with conv_fast f (bit 0: index shortcut, bit 1: residual prefetch,
# bit 2: direct epilogue everywhere, bit 3: swizzled K tiles, bit 4: direct epilogue for ragged Cout; 3 = round 2, default 27)
# and further ymk_debug_option settings (reset to EXTRA_DEFAULTS after)
EXTRA_DEFAULTS = {}
VARIANTS = [v.strip() for v in os.environ.get("VARIANTS", "0,1,2,3,4,5,6").split(",")]
if os.environ.get("ONLY"):  # substring filter on the shape names (PMC runs profile one or two shapes)
    SHAPES = [sh for sh in SHAPES if any(tok in sh[0] for tok in os.environ["ONLY"].split("|"))]
REPS = int(os.environ.get("REPS", 5))


def make_inputs(shape):
    """Operands of one shape, drawn once on the device (every variant of the shape sees the same tensors)."""
    name, n, h, w, cin, cout, k, stride, pad, dil, act, res = shape
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g, device=dev)
    wt = (torch.randn(cout, cin, k, k, generator=g, device=dev) / (cin * k * k) ** 0.5).cpu().contiguous()
    data = os.environ.get("DATA", "random")  # "zeros" / "relu": how much of the rate is the chip's power management
    if data == "zeros":
        x.zero_()
        wt.zero_()
    elif data == "relu":
        x.clamp_(min=0)
    sc = (torch.rand(cout, generator=g, device=dev) + 0.5).cpu().contiguous()
    bi = torch.randn(cout, generator=g, device=dev).cpu().contiguous()
    oh = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn(n, oh, ow, cout, generator=g, device=dev) if res else None
    return x, wt, sc, bi, r, oh, ow


def run(shape, variant, inputs, reps=REPS):
    name, n, h, w, cin, cout, k, stride, pad, dil, act, res = shape
    x, wt, sc, bi, r, oh, ow = inputs
    y = torch.empty(n, oh, ow, cout, device=dev)
    vnum, _, rest = str(variant).partition("/")
    fast, _, extra = rest.partition("/")
    extras = dict(kv.split("=") for kv in extra.split(";") if kv)
    force = -1
    split, split_tile = 0, 0
    if vnum.startswith("b"):  # "b<ns>[t<tile>]": bf16-split operands (ymk_conv_bf16.hip), ns planes, tile shape selector
        body, _, tl = vnum[1:].partition("t")
        split, split_tile, vnum = int(body), int(tl or 0), "0"
    _lib.debug_option("conv_split", split)
    _lib.debug_option("conv_split_tile", split_tile)
    if vnum.startswith("s"):  # "s<i>": split-K candidate i (ymk_debug_option splitk_force)
        force, vnum = int(vnum[1:]), "0"
    _lib.debug_option("splitk_force", force)
    _lib.debug_option("conv_variant", int(vnum))
    _lib.debug_option("conv_fast", int(fast) if fast else _lib.CONV_FAST_DEFAULT)
    for key, val in extras.items():
        _lib.debug_option(key, int(val))
    times = []
    for i in range(reps + 1):
        _lib.check(lib.ymk_prof_begin())
        _lib.check(lib.ymk_op_conv2d(x.data_ptr(), n, h, w, cin, wt.data_ptr(), cout, cin, k, k, sc.data_ptr(), bi.data_ptr(),
                                     _lib.ptr(r), stride, pad, dil, act, 0, y.data_ptr(), None))
        ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        if i:
            times.append(ms.value)
    _lib.debug_option("conv_variant", 0)
    _lib.debug_option("conv_split", 0)
    _lib.debug_option("conv_split_tile", 0)
    _lib.debug_option("splitk_force", -1)
    _lib.debug_option("conv_fast", _lib.CONV_FAST_DEFAULT)
    for key, val in EXTRA_DEFAULTS.items():
        _lib.debug_option(key, val)
    t = float(np.median(times))
    return fl.value / (t * 1e-3) / 1e12, t * 1e3, float(y.flatten()[:4096].double().sum().item()), y


print(f"{'shape':34s} " + " ".join(f"{'v' + str(v):>14s}" + (" " * 10 if str(v).startswith("b") else "") for v in VARIANTS))
for shape in SHAPES:
    cells = []
    ref = y_ref = None
    inputs = make_inputs(shape)
    run(shape, VARIANTS[0], inputs, reps=2)  # untimed: the first launches after the allocations run slow whatever the variant
    for v in VARIANTS:
        tf, us, chk, y = run(shape, v, inputs)
        if ref is None:
            ref, y_ref = chk, y
        # "=": every output bit equals the first variant's; "!": the checksums differ beyond rounding
        cell = f"{tf:6.1f}TF{us:6.0f}us" + ("=" if y is not y_ref and torch.equal(y, y_ref) else "" if abs(chk - ref) <= 1e-3 * max(1.0, abs(ref)) else "!")
        if str(v).startswith("b"):  # bf16-split: error against the first variant's (fp32) output, relative to its largest value
            cell += f" e={float((y - y_ref).abs().max() / y_ref.abs().max()):.1e}"
        cells.append(cell)
    print(f"{shape[0]:34s} " + " ".join(cells), flush=True)

"""The serial conv-roofline pass of bench.py (analyzer workload, one wave at a time, HIP events around every implicit-GEMM
launch) under several settings of the conv kernels' A/B knobs, in ONE process on the same pages:

    python tools/roofline_ab.py "conv_fast=3" "conv_fast=11" "conv_fast=27" [--dump DIR]

Per setting: conv ms per page, TFLOP/s, fraction of the fp32 MFMA peak (median of 3 passes), whether every page's schema
equals the first setting's, and - with --dump - one line per launch (shape, tile, time) in DIR/<setting>.txt."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from yomitoku_amd import _lib  # noqa: E402

DEFAULTS = {"conv_fast": _lib.CONV_FAST_DEFAULT, "conv_variant": 0, "no_splitk": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+", help='"key=val,key=val" per setting (ymk_debug_option)')
    ap.add_argument("--pages", type=int, default=16)
    ap.add_argument("--wave", type=int, default=8)
    ap.add_argument("--dump", default=None)
    args = ap.parse_args()
    device = bench.rank_device(0)
    lib = _lib.load()
    sds = bench.make_checkpoints("lite")
    sds = bench.calibrate_heads(sds, device, bench.Page(0, device))
    pages = bench.make_pages(list(range(args.pages)), device)
    an = bench.build_analyzer(device, sds, "lite")
    an.truth = pages
    an.concurrent_chains = False
    resident = [p.dev for p in pages]
    an.analyze_pages(resident, wave=args.wave)
    torch.cuda.synchronize()
    first = None
    for setting in args.settings:
        opts = dict(kv.split("=") for kv in setting.split(",") if kv)
        for key, val in opts.items():
            _lib.debug_option(key, int(val))
        out = an.analyze_pages(resident, wave=args.wave)  # once untimed under this setting
        torch.cuda.synchronize()
        same = None
        dumped = [json.dumps(r[0].model_dump(), sort_keys=True, default=str) for r in out]
        if first is None:
            first = dumped
        else:
            same = sum(a == b for a, b in zip(first, dumped))
        roof = bench.conv_roofline(lib, lambda: an.analyze_pages(resident, wave=args.wave), len(resident), "page", "conv")
        det = bench.conv_roofline(lib, lambda: an.text_detector.forward_pages(resident[: args.wave]), args.wave, "page", "conv")
        print(json.dumps({"setting": setting, "conv_ms_per_page": roof["kernel_ms_per_page"], "tflops": roof["achieved_tflops"], "frac": roof["mfma"]["frac"],
                          "passes": roof["serial_passes_tflops"], "dbnet_tflops": det["achieved_tflops"], "dbnet_frac": det["mfma"]["frac"],
                          "pages_equal_to_first_setting": same, "of": len(dumped)}), flush=True)
        if args.dump:
            os.makedirs(args.dump, exist_ok=True)
            _lib.debug_option("prof_dump", 1)
            sys.stderr.flush()
            saved = os.dup(2)
            fd = os.open(os.path.join(args.dump, setting.replace("=", "").replace(",", "_") + ".txt"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
            os.dup2(fd, 2)
            try:
                _lib.check(lib.ymk_prof_begin())
                an.analyze_pages(resident[: args.wave], wave=args.wave)
                torch.cuda.synchronize()
                ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
                _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
            finally:
                os.dup2(saved, 2)
                os.close(fd)
                os.close(saved)
                _lib.debug_option("prof_dump", 0)
        for key, val in DEFAULTS.items():
            _lib.debug_option(key, val)
    an.close()


if __name__ == "__main__":
    main()

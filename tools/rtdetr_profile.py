"""One RT-DETRv2 forward (the table-structure net: 3 classes, 640 x 640) at batch B, alone on the device: wall time per
forward and per image, and - under `rocprofv3 --kernel-trace --stats --output-format csv` - its kernels.  With --dump the
library's per-launch table of the convolution / linear launches (description, ms, TFLOP/s-equivalent, GB/s).

    python tools/rtdetr_profile.py --batch 16 [--reps 5] [--dump]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yomitoku_amd import _lib  # noqa: E402
from yomitoku_amd.nets import RTDETRv2  # noqa: E402
from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dump", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    net = RTDETRv2({"RTDETRTransformerv2": {"num_classes": 3}}).load_state_dict(rtdetr_state_dict(1243, num_classes=3)).to(dev)
    x = torch.rand(args.batch, 3, 640, 640, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    net(x)
    net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.reps
    out = {"batch": args.batch, "ms_per_forward": round(ms, 3), "ms_per_image": round(ms / args.batch, 3),
           "tflops_equivalent": round(137.5e9 * args.batch / (ms * 1e-3) / 1e12, 1), "workspace_gb": round(net.workspace_bytes / 2 ** 30, 2)}
    _lib.check(lib.ymk_prof_begin())
    net(x)
    torch.cuda.synchronize()
    cms, cfl, cln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.ymk_prof_end(ctypes.byref(cms), ctypes.byref(cfl), ctypes.byref(cln)))
    out.update(conv_launches=int(cln.value), conv_ms=round(cms.value, 3), conv_gflop=round(cfl.value / 1e9, 1),
               conv_share_of_forward=round(cms.value / ms, 3))
    print(out)
    if args.dump:
        _lib.debug_option("prof_dump", 1)
        _lib.check(lib.ymk_prof_begin())
        net(x)
        torch.cuda.synchronize()
        _lib.check(lib.ymk_prof_end(ctypes.byref(cms), ctypes.byref(cfl), ctypes.byref(cln)))
        _lib.debug_option("prof_dump", 0)
    net.close()


if __name__ == "__main__":
    main()

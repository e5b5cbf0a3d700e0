"""Per-launch conv/GEMM timings (ymk_debug_option prof_dump) for each net at the shapes the analyzer bench uses."""
import ctypes, os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from yomitoku_amd import _lib, imaging
from yomitoku_amd.nets import DBNet, PARSeq, RTDETRv2

dev = torch.device("cuda:0")
lib = _lib.load()
_lib.debug_option("prof_dump", 1)
sds = bench.make_checkpoints()


def prof(name, fn):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
    sys.stderr.write(f"==== {name}: wall {wall:.2f} ms\n"); sys.stderr.flush()
    _lib.check(lib.ymk_prof_begin())
    fn(); torch.cuda.synchronize()
    ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
    sys.stderr.write(f"==== {name}: conv/gemm launches {ln.value}, {ms.value:.3f} ms in them, {fl.value/1e9:.1f} GFLOP, "
                     f"{fl.value/ms.value/1e9:.1f} TFLOP/s\n"); sys.stderr.flush()


which = sys.argv[1:] or ["lay", "tab", "rec", "det"]
if "lay" in which:
    net = RTDETRv2({"RTDETRTransformerv2": {"num_classes": 6}}).load_state_dict(sds["lay"]).to(dev)
    x = torch.rand(1, 3, 640, 640, device=dev)
    prof("rtdetr layout B=1", lambda: net(x))
if "tab" in which:
    net = RTDETRv2({"RTDETRTransformerv2": {"num_classes": 3}}).load_state_dict(sds["tab"]).to(dev)
    x = torch.rand(2, 3, 640, 640, device=dev)
    prof("rtdetr table B=2", lambda: net(x))
if "rec" in which:
    from yomitoku_amd.text_recognizer import TextRecognizer
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True, batch_bucketing=True)
    rec.model.load_state_dict(sds["rec"])
    x = torch.rand(40, 3, 32, 320, device=dev)
    prof("parseq B=40 W=320", lambda: rec.model(x))
    sys.stderr.write("AR steps %d\n" % rec.model.last_ar_steps)
if "det" in which:
    net = DBNet().load_state_dict(sds["det"]).to(dev)
    x = torch.rand(1, 3, 1280, 1600, device=dev)
    prof("dbnet 1280x1600", lambda: net(x))

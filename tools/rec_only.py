"""Recogniser only, one page, for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/rec_only.py [stage]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
sds = bench.make_checkpoints()
stage = sys.argv[1] if len(sys.argv) > 1 else "rec"
P = bench.Page(1, dev)
if stage == "rec":
    from yomitoku_amd.text_recognizer import TextRecognizer
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                         batch_bucketing=True)
    rec.model.load_state_dict(sds["rec"])
    fn = lambda: rec(P.dev, P.quads)
elif stage == "tsr":
    from yomitoku_amd.table_structure_recognizer import TableStructureRecognizer
    m = TableStructureRecognizer(from_pretrained=False, device="cuda:0")
    m.model.load_state_dict(sds["tab"])
    fn = lambda: m(P.dev, P.tables)
elif stage == "lay":
    from yomitoku_amd.layout_parser import LayoutParser
    m = LayoutParser(from_pretrained=False, device="cuda:0")
    m.model.load_state_dict(sds["lay"])
    fn = lambda: m(P.dev)
else:
    from yomitoku_amd.text_detector import TextDetector
    m = TextDetector(from_pretrained=False, device="cuda:0")
    m.model.load_state_dict(sds["det"])
    fn = lambda: m(P.dev)
fn(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    fn()
torch.cuda.synchronize()
print("%s: %.2f ms per page" % (stage, (time.perf_counter() - t) / 5 * 1e3))

"""Host-side stage timeline of the analyzer bench: which thread is in which stage when."""
import gc, json, os, sys, threading, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from yomitoku_amd.parallel import PageParallel

dev = torch.device("cuda:0")
W = int(os.environ.get("W", 8)); NP = int(os.environ.get("NP", 32))
sds = bench.make_checkpoints()
sds = bench.calibrate_heads(sds, dev, bench.Page(0, dev))
pages = [bench.Page(i, dev) for i in range(NP)]
EV = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            EV.append((threading.get_ident(), label, t, time.perf_counter()))
    setattr(obj, name, inner)


class Timed:
    def __init__(self, inner, label):
        self.__dict__["_inner"], self.__dict__["_label"] = inner, label

    def __call__(self, *a, **k):
        t = time.perf_counter()
        try:
            return self._inner(*a, **k)
        finally:
            EV.append((threading.get_ident(), self._label, t, time.perf_counter()))

    def __getattr__(self, n):
        return getattr(self._inner, n)

    def __setattr__(self, n, v):
        setattr(self._inner, n, v)


def make(i):
    run = bench.build_analyzer(dev, sds)
    an = run.analyzer
    for o, n, l in ((an.text_detector, "preprocess", "det.pre"), 
                    (an.text_detector, "postprocess", "det.post"), (an.text_recognizer, "preprocess", "rec.plan"),
                    (an.text_recognizer, "_collate", "rec.collate"),
                    (an.text_recognizer, "postprocess", "rec.decode"),
                    (an.layout.inner.layout_parser, "postprocess", "lay.post"),
                    
                    (an.layout.inner.table_structure_recognizer, "postprocess", "tsr.post"), (an, "aggregate", "aggregate")):
        if hasattr(o, n):
            wrap(o, n, l)
        else:
            print("no", l, n)

    an.text_detector.model = Timed(an.text_detector.model, "det.net")
    an.text_recognizer.model = Timed(an.text_recognizer.model, "rec.net")
    an.layout.inner.layout_parser.model = Timed(an.layout.inner.layout_parser.model, "lay.net")
    an.layout.inner.table_structure_recognizer.model = Timed(an.layout.inner.table_structure_recognizer.model, "tsr.net")

    def timed(p):
        t = time.perf_counter()
        r = run(p)
        EV.append((threading.get_ident(), "page", t, time.perf_counter()))
        return r
    return timed


pool = PageParallel(make, n_workers=W)
pool.map(pages[:W * 2]); torch.cuda.synchronize()
if os.environ.get("GCFREEZE"):
    gc.collect(); gc.freeze()
if os.environ.get("GCOFF"):
    gc.disable()
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info["generation"], time.perf_counter())))
EV.clear()
t0 = time.perf_counter()
pool.map(pages); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("pages/s %.2f" % (NP / dt))
import collections
d = collections.defaultdict(list)
for th, l, a, b in EV:
    d[l].append((b - a) * 1e3)
for l, v in sorted(d.items()):
    v.sort()
    print("%-12s n=%4d mean %.2f p50 %.2f p90 %.2f max %.2f ms" % (l, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)], v[-1]))
g2 = [(p, g, t - t0) for p, g, t in gcs]
st = {}
tot = collections.Counter(); cnt = collections.Counter()
for p, g, t in g2:
    if p == "start": st[g] = t
    else:
        tot[g] += t - st[g]; cnt[g] += 1
print("gc:", {g: (cnt[g], round(tot[g] * 1e3, 1)) for g in cnt})
json.dump({"t0": t0, "ev": EV, "gc": g2}, open("gpurun_out/host_trace_%s.json" % os.environ.get("TAG", "x"), "w"))

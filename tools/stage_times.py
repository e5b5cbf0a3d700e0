import sys, time, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
sds = bench.make_checkpoints()
pages = [bench.Page(i, dev) for i in range(4)]
sds = bench.calibrate_heads(sds, dev, pages[0])
run = bench.build_analyzer(dev, sds); an = run.analyzer
def T(f, *a, n=3):
    f(*a); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r = f(*a)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3, r
P = pages[1]; p = P.dev; an.truth = P
det = an.text_detector
t, x = T(det.preprocess, p); print("det.pre %.2f ms" % t)
t, pr = T(det.model, x); print("det.net %.2f ms" % t)
t, q = T(det.postprocess, pr, P.img.shape[:2]); print("det.post(incl d2h) %.2f ms (%d boxes, frac %.4f)" % (t, len(q[0]), (pr["binary"]>0.3).float().mean().item()))
rec = an.text_recognizer
t, b = T(rec.preprocess, p, P.quads); print("rec.plan %.2f ms, batches %s" % (t, [len(x) for x in b[0]]))
batches, points, dataset, order = b
for plans in batches:
    t, data = T(rec._collate, dataset, plans)
    t2, lg = T(rec.model, data)
    t3, st = T(rec.model.token_stats, lg)
    t4, _ = T(lambda: rec.postprocess(st, points[:len(plans)]))
    print("  batch %d x %d: collate %.2f ms, forward %.2f ms (AR steps %d), stats %.2f ms, decode %.2f ms" % (data.shape[0], data.shape[-1], t, t2, rec.model.last_ar_steps, t3, t4))
t, _ = T(rec, p, P.quads); print("rec total %.2f ms" % t)
lp = an.layout.inner.layout_parser
t, x = T(lp.preprocess, p); print("lay.pre %.2f ms" % t)
t, pr = T(lp.model, x); print("lay.net %.2f ms" % t)
t, r = T(lp.postprocess, pr, P.img.shape[:2]); print("lay.post %.2f ms (%d elems)" % (t, len(r.paragraphs)+len(r.tables)+len(r.figures)))
ts = an.layout.inner.table_structure_recognizer
t, o = T(ts, p, P.tables); print("tsr total %.2f ms (%d tables, cells %s)" % (t, len(P.tables), [len(x.cells) for x in o[0]]))
t, _ = T(run, P); print("analyzer total %.2f ms" % t)
import cProfile, pstats
pr_ = cProfile.Profile(); pr_.enable(); run(P); pr_.disable()
pstats.Stats(pr_).sort_stats("cumulative").print_stats(14)

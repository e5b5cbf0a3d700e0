"""Stand-ins that bench.py swaps in under YMK_BENCH_DRY=1 (tests/test_bench_dry.py): pages without pixels, page workers
that only check that the rank's checkpoints reached their process, checkpoints of a few bytes.  Test scaffolding -
nothing here computes anything the benchmark reports."""
import time
from collections import OrderedDict

import torch


class Page:
    def __init__(self, seed, device):
        self.seed = int(seed)
        self.img = self.dev = None
        self.quads = [[[0, 0], [10, 0], [10, 10], [0, 10]]] * (3 + self.seed % 4)
        self.tables = [[0, 0, 50, 50]]
        self.paragraphs = [[0, 0, 100, 20], [0, 30, 100, 50]]


def make_checkpoints(model_set="lite"):
    return {k: OrderedDict(w=torch.arange(8, dtype=torch.float32) + i, steps=torch.tensor(7 + i, dtype=torch.int64))
            for i, k in enumerate(("det", "rec", "lay", "tab"))}


def calibrate_heads(sds, device, page):
    return sds


def build_analyzer(device, sds, model_set="lite"):
    want = make_checkpoints()

    def work(wave):
        for k, sd in want.items():  # the broadcast / hand-over to helper processes kept every tensor intact
            for name, t in sd.items():
                got = sds[k][name]
                assert torch.equal(torch.as_tensor(got).to(t.dtype), t), (k, name)
        time.sleep(0.002 * len(wave))
        return [page.seed for page in wave]

    work.analyzer = None
    return work


def install(namespace):
    namespace.update(Page=Page, make_checkpoints=make_checkpoints, calibrate_heads=calibrate_heads, build_analyzer=build_analyzer)

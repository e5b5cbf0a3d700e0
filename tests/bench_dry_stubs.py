"""Stand-ins that bench.py swaps in under YMK_BENCH_DRY=1 (tests/test_bench_dry.py): pages without pixels, page workers
that only check that the rank's checkpoints reached their process, checkpoints of a few bytes.  Test scaffolding -
nothing here computes anything the benchmark reports."""
import os
import time
from collections import OrderedDict

import numpy as np
import torch


class Page:
    def __init__(self, seed, device):
        self.seed = int(seed)
        self.img = np.full((2, 2, 3), self.seed % 251, dtype=np.uint8)
        self.img[0, 0, 0] = self.seed // 251
        self.dev = None
        self.quads = [[[0, 0], [10, 0], [10, 10], [0, 10]]] * (3 + self.seed % 4)
        self.tables = [[0, 0, 50, 50]]
        self.paragraphs = [[0, 0, 100, 20], [0, 30, 100, 50]]


def make_checkpoints(model_set="lite"):
    return {k: OrderedDict(w=torch.arange(8, dtype=torch.float32) + i, steps=torch.tensor(7 + i, dtype=torch.int64))
            for i, k in enumerate(("det", "rec", "lay", "tab"))}


def calibrate_heads(sds, device, page):
    return sds


class StubAnalyzer:
    """`serve` of the real analyzer, without the work: checks that the rank's checkpoints arrived intact, sleeps 2 ms per
    page, returns the page seeds in order.  YMK_BENCH_POISON="<rank>:<seed>" makes that page of that rank fail the way
    DocumentAnalyzer.serve reports a failing page: its entry is the exception, the job goes on."""

    def __init__(self, sds):
        self.sds = sds
        self.truth = None

    def serve(self, imgs, wave=8, in_flight=3):
        want = make_checkpoints()
        for k, sd in want.items():  # the broadcast kept every tensor intact
            for name, t in sd.items():
                got = self.sds[k][name]
                assert torch.equal(torch.as_tensor(got).to(t.dtype), t), (k, name)
        poison = os.environ.get("YMK_BENCH_POISON", "")
        bad = int(poison.split(":")[1]) if poison and int(poison.split(":")[0]) == int(os.environ.get("RANK", "0")) else None
        out = []
        for start in range(0, len(imgs), wave):
            chunk = imgs[start : start + wave]
            time.sleep(0.002 * len(chunk))
            for img in chunk:
                seed = int(img[0, 0, 0]) * 251 + int(img[1, 1, 1])
                out.append(RuntimeError(f"poisoned page {seed}") if seed == bad else seed)
        return out

    def close(self):
        pass


def build_analyzer(device, sds, model_set="lite"):
    return StubAnalyzer(sds)


def install(namespace):
    namespace.update(Page=Page, make_checkpoints=make_checkpoints, calibrate_heads=calibrate_heads, build_analyzer=build_analyzer)

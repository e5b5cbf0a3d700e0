"""Orchestration of DocumentAnalyzer.serve (yomitoku_amd/serving.py) with stub stages on the host: page order, waves,
back-pressure, per-page failure isolation (cli/main.py:555-564: a failing page is reported, the job goes on).  The
numerics of the real stages are tests/test_serving_gpu.py's."""
import threading
import time
from types import SimpleNamespace

import numpy as np
import pytest

from yomitoku_amd.serving import PagePipeline


class StubAnalyzer:
    """Stages that only move page ids along; a page whose first pixel is 255 poisons the stage named in its second."""

    STAGES = {1: "detect", 2: "boxes", 3: "recognize", 4: "layout", 5: "finish", 6: "crops", 7: "decode", 8: "tables", 9: "cells"}

    def __init__(self, split=False, delay=0.0):
        self.text_detector = SimpleNamespace(device="cpu")
        self.split_text_across_cells = split
        self.delay = delay
        self.log = []
        self.in_flight_seen = 0
        self.lanes_seen = set()
        self._live = set()
        self._lock = threading.Lock()

    def _poison(self, wave, stage):
        for img in wave.imgs:
            if img[0, 0, 0] == 255 and self.STAGES[int(img[0, 0, 1])] == stage:
                raise RuntimeError(f"poisoned page in {stage}")

    def _stage_detect(self, wave):
        with self._lock:
            self._live.add(wave.seq)
            self.in_flight_seen = max(self.in_flight_seen, len(self._live))
        time.sleep(self.delay)
        self._poison(wave, "detect")
        wave.maps = [int(img[0, 0, 2]) for img in wave.imgs]

    def _stage_boxes(self, wave):
        self._poison(wave, "boxes")
        wave.dets = [m + 1000 for m in wave.maps]

    def _stage_split(self, wave):
        assert wave.lays is not None  # the layout chain of this wave has finished
        wave.dets = [d + 5000 for d in wave.dets]

    def _stage_crops(self, wave):
        self._poison(wave, "crops")
        wave.rec_plan = list(wave.dets)

    def _stage_recognize(self, wave, lane=0):
        self.lanes_seen.add(lane)
        time.sleep(self.delay)
        self._poison(wave, "recognize")
        wave.rec_plan = [d * 2 for d in wave.rec_plan]

    def _stage_decode(self, wave):
        self._poison(wave, "decode")
        wave.recs = wave.rec_plan

    def _stage_layout(self, wave):
        time.sleep(self.delay)
        self._poison(wave, "layout")
        wave.lay_raw = len(wave)

    def _stage_tables(self, wave):
        self._poison(wave, "tables")
        wave.tab_raw = wave.lay_raw

    def _stage_cells(self, wave):
        self._poison(wave, "cells")
        wave.lays = [wave.tab_raw] * len(wave)

    def _stage_finish(self, wave, k):
        if wave.imgs[k][0, 0, 0] == 255 and self.STAGES[int(wave.imgs[k][0, 0, 1])] == "finish":
            raise ValueError("poisoned page in finish")
        with self._lock:
            self._live.discard(wave.seq)
        self.log.append((wave.seq, tuple(wave.ids)))
        return (wave.ids[k], wave.recs[k], wave.lays[k])


def page(tag, poison_stage=0):
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    img[0, 0] = (255 if poison_stage else 0, poison_stage, tag)
    return img


def test_results_in_page_order_and_waves_of_the_asked_size():
    an = StubAnalyzer(delay=0.002)
    pipe = PagePipeline(an, wave=4, in_flight=3)
    out = pipe.serve([page(i) for i in range(18)])
    assert [o[0] for o in out] == list(range(18))
    assert [o[1] for o in out] == [(i + 1000) * 2 for i in range(18)]
    assert [o[2] for o in out] == [4] * 16 + [2] * 2  # 4 full waves and a last wave of 2
    assert pipe.last_job == {"pages": 18, "waves": 5, "retried_pages": 0}
    assert 1 <= an.in_flight_seen <= 3
    assert an.lanes_seen == {0, 1}  # two recogniser lanes took waves (results still in page order)
    assert pipe.serve([]) == []
    pipe.close()
    one = PagePipeline(StubAnalyzer(), wave=4, in_flight=3, rec_lanes=1)
    assert [o[0] for o in one.serve([page(i) for i in range(9)])] == list(range(9)) and one.analyzer.lanes_seen == {0}
    one.close()


@pytest.mark.parametrize("stage", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_a_poisoned_page_fails_alone(stage):
    an = StubAnalyzer()
    pipe = PagePipeline(an, wave=4, in_flight=2)
    pages = [page(i) for i in range(10)]
    pages[5] = page(5, poison_stage=stage)
    out = pipe.serve(pages)
    assert isinstance(out[5], (RuntimeError, ValueError)) and "poisoned" in str(out[5])
    good = [i for i in range(10) if i != 5]
    assert [out[i][0] for i in good] == good and [out[i][1] for i in good] == [(i + 1000) * 2 for i in good]
    if stage != 5:  # the wave {4..7} was re-run page by page; aggregation failures need no re-run
        assert pipe.last_job["retried_pages"] == 4 and pipe.last_job["waves"] == 3 + 4
        assert [out[i][2] for i in (4, 6, 7)] == [1, 1, 1]
    # the pipeline is still usable
    assert [o[0] for o in pipe.serve([page(i) for i in range(3)])] == [0, 1, 2]
    pipe.close()


def test_bad_inputs_are_per_page_failures(tmp_path):
    from PIL import Image

    an = StubAnalyzer()
    pipe = PagePipeline(an, wave=2, in_flight=2)
    out = pipe.serve([page(0), str(tmp_path / "missing.png"), page(2)])
    assert out[0][0] == 0 and out[2][0] == 2 and isinstance(out[1], Exception)
    # a multi-frame file contributes one page per frame; with_source names the file and the frame of every entry
    frames = [Image.fromarray(np.full((40, 40, 3), 40 * k, dtype=np.uint8)) for k in range(3)]
    tiff = str(tmp_path / "three.tiff")
    frames[0].save(tiff, save_all=True, append_images=frames[1:])
    tagged = pipe.serve([page(0), tiff, str(tmp_path / "missing.png"), page(5)], with_source=True)
    assert [(s, f) for s, f, _ in tagged] == [(0, 0), (1, 0), (1, 1), (1, 2), (2, 0), (3, 0)]
    assert isinstance(tagged[4][2], Exception) and tagged[5][2][0] == 5 and tagged[0][2][0] == 0
    pipe.close()


def test_split_text_across_cells_waits_for_the_layout_chain():
    an = StubAnalyzer(split=True, delay=0.002)
    pipe = PagePipeline(an, wave=3, in_flight=2)
    out = pipe.serve([page(i) for i in range(7)])
    assert [o[1] for o in out] == [(i + 1000 + 5000) * 2 for i in range(7)]
    pipe.close()


def test_full_collections_are_deferred_during_a_job_and_restored_after():
    """A generation-2 pass of the cyclic collector holds the GIL while it walks every live container (100+ ms with a
    few hundred page results alive): serve() postpones it for the duration of the job and puts the thresholds back."""
    import gc

    before = gc.get_threshold()
    seen = []

    class Spy(StubAnalyzer):
        def _stage_detect(self, wave):
            seen.append(gc.get_threshold())
            super()._stage_detect(wave)

    pipe = PagePipeline(Spy(), wave=2, in_flight=2)
    pipe.serve([page(i) for i in range(4)])
    assert all(t[:2] == before[:2] and t[2] >= 1 << 30 for t in seen) and len(seen) == 2
    assert gc.get_threshold() == before
    pipe.defer_full_gc = False
    seen.clear()
    pipe.serve([page(i) for i in range(2)])
    assert seen == [before]
    pipe.close()


def test_switch_interval_is_shortened_for_the_job_and_restored_after():
    """The stage threads are launch-latency-bound: serve() runs the job with a 0.2 ms thread switch interval (the bench used
    to set it process-wide for the product) and puts the interpreter's own value back, also when a page fails."""
    import sys

    before = sys.getswitchinterval()
    seen = []

    class Spy(StubAnalyzer):
        def _stage_detect(self, wave):
            seen.append(sys.getswitchinterval())
            super()._stage_detect(wave)

    pipe = PagePipeline(Spy(), wave=2, in_flight=2)
    res = pipe.serve([page(0), page(1, poison_stage=3), page(2)])
    assert isinstance(res[1], BaseException) and not isinstance(res[0], BaseException)
    assert seen and all(abs(v - 2e-4) < 1e-9 for v in seen)
    assert sys.getswitchinterval() == before
    pipe.switch_interval = 0  # leave the interpreter alone
    seen.clear()
    pipe.serve([page(3)])
    assert seen == [before]
    pipe.close()


def test_reserve_once_falls_back_to_grow_on_demand(monkeypatch):
    """the worst-case workspace reservation of a live handle is attempted once; a failing hipMalloc is
    logged and the model keeps growing its workspace on demand (not retried per call: a retry frees the working slab); a
    rebuilt handle starts over."""
    from yomitoku_amd import _lib, nets

    calls = []

    class Net(nets.HipNet):
        def reserve(self, n, h, w, device=None):
            calls.append((n, h, w))
            if len(calls) == 1:
                raise _lib.YmkError("hipMalloc failed: out of memory")

    net = Net()
    net._h = 1234  # a live handle
    assert net.reserve_once(8, 1600, 1280) is False and net.reserve_once(8, 1600, 1280) is False and len(calls) == 1
    net._h, net._reserve_tried_for = 5678, None  # what _build() does: a new handle has no reservation
    assert net.reserve_once(8, 1600, 1280) is True and net.reserve_once(8, 1600, 1280) is True and len(calls) == 2
    net._h = None  # nothing to destroy


def test_two_jobs_at_once_restore_the_switch_interval_once_the_last_one_leaves():
    """the switch interval is process-wide; two pipelines serving concurrently must not restore it out of order
    (the first one out putting the long interval back under the second, or the short one staying behind for good)."""
    import sys

    from yomitoku_amd.serving import _switch_interval

    before = sys.getswitchinterval()
    inside = {}
    a_in, b_in, a_out = threading.Event(), threading.Event(), threading.Event()

    def job_a():
        with _switch_interval(2e-4):
            a_in.set()
            b_in.wait(5)
        inside["after_a_left"] = sys.getswitchinterval()  # B is still serving: the short interval must still be in force
        a_out.set()

    def job_b():
        a_in.wait(5)
        with _switch_interval(2e-4):
            b_in.set()
            a_out.wait(5)
            inside["b_still_in"] = sys.getswitchinterval()

    ta, tb = threading.Thread(target=job_a), threading.Thread(target=job_b)
    ta.start(), tb.start()
    ta.join(10), tb.join(10)
    assert inside["after_a_left"] == pytest.approx(2e-4, rel=0.02) and inside["b_still_in"] == pytest.approx(2e-4, rel=0.02)  # (the interpreter stores microseconds)
    assert sys.getswitchinterval() == before


def test_handover_objects_replace_only_what_they_say():
    """yomitoku_amd.testing: the identity hook returns what it is given; TruthHandover puts the pages' truth into the discrete
    hand-overs and keeps the PRODUCT'S table structures; NetOutputHandover writes one query per true unit into the layout
    net's raw output and touches nothing else."""
    from types import SimpleNamespace as NS

    import numpy as np

    from yomitoku_amd import schemas as sch
    from yomitoku_amd.testing import Handover, NetOutputHandover, TruthHandover

    truth = [NS(quads=[[[1, 1], [9, 1], [9, 5], [1, 5]]], tables=[[10, 10, 50, 40]], paragraphs=[[0, 0, 60, 8], [0, 50, 60, 58]],
                truth_map=np.full((4, 6), 0.5, np.float32)),
             NS(quads=[], tables=[], paragraphs=[[5, 5, 20, 20]], truth_map=np.zeros((4, 6), np.float32))]
    table = sch.TableStructureRecognizerSchema(box=[10, 10, 50, 40], n_row=1, n_col=1, rows=[], cols=[], spans=[], cells=[], order=0)
    lay = sch.LayoutAnalyzerSchema(paragraphs=[sch.Element(id=None, box=[1, 2, 3, 4], score=0.6, role=None, contents=None)], tables=[table], figures=[])
    parsed = [sch.LayoutParserSchema(paragraphs=lay.paragraphs, tables=[], figures=[]), sch.LayoutParserSchema(paragraphs=[], tables=[], figures=[])]
    wave = NS(ids=[2, 5], lay_parsed=parsed)  # page ids 2 and 5 -> truth[0] and truth[1]
    maps = [np.ones((4, 6), np.float32), np.ones((4, 6), np.float32)]
    dets = [sch.TextDetectorSchema(points=[], scores=[]), sch.TextDetectorSchema(points=[], scores=[])]
    ident = Handover()
    assert ident.maps(wave, maps) is maps and ident.boxes(wave, dets) is dets and ident.layouts(wave, [lay]) == [lay]
    stats = {}
    th = TruthHandover(truth, stats)
    assert [m.tolist() for m in th.maps(wave, maps)] == [truth[0].truth_map.tolist(), truth[1].truth_map.tolist()]
    got = th.boxes(wave, dets)
    assert got[0].points == truth[0].quads and got[0].scores == [1.0] and got[1].points == []
    assert th.table_boxes(wave, [[[0, 0, 1, 1]], []]) == [[[10, 10, 50, 40]], []]
    out = th.layouts(wave, [lay, sch.LayoutAnalyzerSchema(paragraphs=[], tables=[], figures=[])])
    assert [p.box for p in out[0].paragraphs] == truth[0].paragraphs and out[0].tables == [table] and out[1].tables == []
    assert stats == {"det_boxes": [0, 0], "layout_boxes": [1, 0], "cells": [0, 0]}
    nh = NetOutputHandover(truth, categories={"tables": 0, "paragraphs": 2})
    raw = [(np.zeros((1, 300, 6), np.float32), np.zeros((1, 300, 4), np.float32), (100, 200)) for _ in range(2)]
    lg, bx, hw = nh.layout_raw(wave, raw)[0]
    assert hw == (100, 200) and (lg[0, :3] > 0).sum() == 3 and lg[0, 0, 2] == 4.0 and lg[0, 2, 0] == 4.0 and (lg[0, 3:] == -12.0).all()
    assert np.allclose(bx[0, 2], (30 / 200, 25 / 100, 40 / 200, 30 / 100))
    assert nh.boxes(wave, dets) is dets and nh.table_boxes(wave, [[], []]) == [[], []]


def test_sources_are_pulled_only_when_a_wave_slot_is_free():
    """serve() reads `sources` lazily and takes a wave slot BEFORE it pulls that wave's first source: a generator that claims
    work from a shared counter (distributed.PageDealer) is never asked for more than `in_flight` waves beyond what has
    finished - the end of a sharded job is not as ragged as a static deal."""
    an = StubAnalyzer(delay=0.02)
    pipe = PagePipeline(an, wave=2, in_flight=2)
    pulled, ahead = [0], []

    def feed(n):
        for i in range(n):
            pulled[0] += 1
            ahead.append(pulled[0] - 2 * len(an.log))  # sources pulled minus sources whose wave has been aggregated
            yield page(i)

    out = pipe.serve(feed(20))
    assert [o[0] for o in out] == list(range(20)) and pipe.last_job["waves"] == 10
    # at most `in_flight` waves out plus the wave being collected (its slot is already held): (2 + 1) x 2 pages; the finish
    # stage logs a wave a moment after its slot is free, hence one wave of slack
    assert max(ahead) <= (2 + 1 + 1) * 2, max(ahead)
    assert pipe.serve(iter([])) == [] and pipe.serve(feed(4))[-1][0] == 3  # the slots all came back (an exact multiple of the wave too)
    pipe.close()

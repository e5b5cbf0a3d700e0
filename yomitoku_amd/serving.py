"""Multi-page serving on ONE GPU from ONE process: host pages in, every page's result out, in page order.

The reference's unit of use is "file in -> all page results out" (cli/main.py:105-137): the page loop calls the
analyzer once per page and a failing file is logged and skipped (cli/main.py:555-564).  On an MI355X one page at a time
leaves the device mostly idle, so `DocumentAnalyzer.serve` runs the page loop as a STAGE PIPELINE over waves of pages:

    caller thread   decode (paths) -> pinned staging ring -> H2D on the copy stream              (PageStager)
    detect          DBNet forwards over the wave, maps back in one DMA per forward               HIP stream
    boxes           DB box extraction, C++, GIL released                                         host
    crops           per-page mini-batches (bucketing, width budget) + the crop kernels           HIP stream
    recognize       ONE grouped PARSeq forward with one greedy loop per wave; TWO lanes (below)   HIP stream x 2
    decode          token decode, un-permutation                                                 host
    layout          RT-DETRv2 layout forward over the wave                                       HIP stream
    tables          layout boxes (host), table-structure forward over all table crops            HIP stream
    cells           row / column / span filters, cell grids                                      host
    finish          word -> cell / paragraph aggregation and reading order, per page             host

The host halves are their own stages on purpose: a thread that owns a network only ever launches - it never sits in box
logic or string decoding while its stream runs dry (measured: with crop planning / decode inside the recognise stage
and the table grid logic inside the layout stage the GPU was idle 16 % of a job, in 10-20 ms holes).
Every network exists once, except the recogniser, which has a second PARSeq handle (a replica of the first's weights,
~50 MB for the --lite model): the grouped forward is the longest stage and half of it is the greedy loop - ~100 dependent,
nearly empty launches - so two lanes let wave k + 1's encoder fill the chip while wave k's loop idles through its steps.  A
model handle is only ever used by one thread, which is what include/ymk.h asks for; six compute streams plus the copy
stream fit the eight hardware queues the package asks the HIP runtime for.  Wave k + 1 is in the detector while wave k decodes text and its tables are parsed; up to `in_flight`
waves are between upload and aggregation (a ring of pinned map buffers per wave slot), which bounds host and device
memory.

Garbage collection: a full (generation-2) pass of CPython's cyclic collector walks every live container object of the
process while holding the GIL - measured 100-180 ms with the results of a few hundred pages alive, six times in a
4 s job - and every stage thread that needs the interpreter to issue its next launch stalls behind it (the GPU idles;
this, not the launch rate, is what held one process at 48 pages/s where two reached 66).  `serve` therefore defers
generation-2 collections for the duration of a job (`defer_full_gc=True`: the young generations keep running, the
previous thresholds come back when the job ends).

Failure isolation: a stage that raises marks its wave failed; the pages of a failed wave are re-run as single-page
waves through the same pipeline, and a page that fails alone gets the exception object as its result - the job goes
on (the reference's per-file `except: log, continue`).  Results equal `DocumentAnalyzer.__call__` page by page
(tests/test_serving_gpu.py)."""

from __future__ import annotations

import contextlib
import gc
import logging
import os
import queue
import sys
import threading
import time
from collections import deque
from typing import Iterable, List, Optional

import numpy as np
import torch

from .data.staging import PageStager

logger = logging.getLogger(__name__)

_STOP = object()


class Wave:
    """The pages that share device batches, and what the stages have produced for them so far."""

    __slots__ = ("seq", "ids", "imgs", "pages", "ring", "uploaded", "maps", "sizes", "dets", "rec_plan", "recs", "lay_raw",
                 "lay_parsed", "tab_raw", "lays", "error", "failed_stage", "layout_done", "joined", "retry", "job")

    def __init__(self, seq=0, ids=(), imgs=(), pages=(), ring=0, uploaded=None, retry=False, job=None):
        self.seq, self.ids, self.imgs, self.pages, self.ring, self.uploaded = seq, list(ids), list(imgs), list(pages), ring, uploaded
        self.job = job  # the serve() call this wave belongs to: a late wave of an aborted job must not write into the next one
        self.sizes = [tuple(int(v) for v in p.shape[:2]) for p in self.pages]
        self.maps = self.dets = self.rec_plan = self.recs = self.lay_raw = self.lay_parsed = self.tab_raw = self.lays = None
        self.error: Optional[BaseException] = None
        self.failed_stage = None
        self.layout_done = threading.Event()
        self.joined = 0
        self.retry = retry

    def __len__(self):
        return len(self.ids)

    def fail(self, stage, exc):
        if self.error is None:
            self.error, self.failed_stage = exc, stage


class _Job:
    """One serve() call: where results go and how many waves are still out."""

    def __init__(self, n_hint=0):
        self.results = {}
        self.cond = threading.Condition()
        self.outstanding = 0
        self.retries: "deque" = deque()  # (page id, host page) to re-run alone
        self.waves = 0
        self.retried_pages = 0


@contextlib.contextmanager
def _full_gc_deferred(on: bool):
    """Generation-2 collections of the cyclic collector postponed while the block runs (module doc)."""
    if not on or not gc.isenabled():
        yield
        return
    t0, t1, t2 = gc.get_threshold()
    gc.set_threshold(t0, t1, 1 << 30)
    try:
        yield
    finally:
        gc.set_threshold(t0, t1, t2)


_switch_lock = threading.Lock()
_switch_users = 0
_switch_saved = None


@contextlib.contextmanager
def _switch_interval(seconds):
    """CPython's thread switch interval shortened while the block runs.  The stage threads are launch-latency-bound: one
    returning from a 50 us library call must not wait 5 ms (the default interval) behind another stage's Python loop before
    it can issue its next launch.  None / 0 leaves the interpreter alone.  The setting is process-wide, so it is reference
    counted: the first job in saves the interpreter's value, the LAST job out restores it - two pipelines serving at once
    (two analyzers, two threads) cannot restore out of order and leave the short interval behind."""
    global _switch_users, _switch_saved
    if not seconds:
        yield
        return
    with _switch_lock:
        if _switch_users == 0:
            _switch_saved = sys.getswitchinterval()
        _switch_users += 1
        sys.setswitchinterval(min(float(seconds), sys.getswitchinterval()))
    try:
        yield
    finally:
        with _switch_lock:
            _switch_users -= 1
            if _switch_users == 0 and _switch_saved is not None:
                sys.setswitchinterval(_switch_saved)
                _switch_saved = None


class PagePipeline:
    def __init__(self, analyzer, wave: int = 16, in_flight: int = 4, defer_full_gc: bool = True, stage_priority=None,
                 rec_lanes: int = 2):
        """rec_lanes: recogniser forwards in flight.  The grouped PARSeq forward is two phases with opposite appetites - the
        ViT encoder (large GEMMs) and the greedy loop (~100 dependent, nearly empty launches) - and it is the longest stage;
        with two lanes (each its own PARSeq handle and HIP stream, the second a replica of the first's weights) wave k + 1's
        encoder fills the chip while wave k's loop idles through its steps.  Forced to 1 with `rec_orientation_fallback`
        (its retry forwards run on the module's own model from the decode path).
        stage_priority: {"detect" | "recognize" | "layout": HIP stream priority (0 default, -1 high)}; also read from
        YMK_STAGE_PRIORITY="recognize:-1,..." (measurement knob)."""
        self.analyzer = analyzer
        self.defer_full_gc = bool(defer_full_gc)
        # thread switch interval for the duration of a job (YMK_SWITCH_INTERVAL overrides; 0 = leave the interpreter's)
        self.switch_interval = float(os.environ.get("YMK_SWITCH_INTERVAL", 2e-4))
        self.stage_priority = dict(stage_priority or {})
        rec = getattr(analyzer, "text_recognizer", None)
        self.rec_lanes = 1 if getattr(rec, "rec_orientation_fallback", False) else max(1, int(os.environ.get("YMK_REC_LANES", rec_lanes)))
        for item in filter(None, os.environ.get("YMK_STAGE_PRIORITY", "").split(",")):
            name, _, value = item.partition(":")
            self.stage_priority.setdefault(name.strip(), int(value))
        self.wave = max(1, int(wave))
        self.in_flight = max(1, int(in_flight))
        self.device = torch.device(analyzer.text_detector.device)
        # a HIP device always, for DocumentAnalyzer (BaseModule refuses anything else); the host-only form exists so that
        # the orchestration (ordering, back-pressure, retries) can be tested with stub stages on a box without a GPU
        self._gpu = self.device.type == "cuda"
        self.stager = PageStager(self.device, slots=self.wave * (self.in_flight + 1)) if self._gpu else None
        self._rings: "queue.Queue" = queue.Queue()
        for r in range(self.in_flight):
            self._rings.put(r)
        names = ("detect", "boxes", "crops", "recognize", "decode", "layout", "tables", "cells", "finish")
        self._q = {name: queue.Queue() for name in names}
        self._job: Optional[_Job] = None
        self._seq = 0
        self.trace = None  # a list: every stage appends (stage, wave seq, pages, t_start, t_end) - tools/serve_trace.py
        self._serve_lock = threading.Lock()
        a = analyzer
        # (stage, function, launches on the GPU?, next stages); the recognise chain and the layout chain join in `finish`
        plan = (("detect", a._stage_detect, True, ("boxes",)), ("boxes", self._boxes, False, ("crops",)),
                ("crops", a._stage_crops, True, ("recognize",)), ("recognize", a._stage_recognize, True, ("decode",)),
                ("decode", a._stage_decode, False, ("finish",)),
                ("layout", a._stage_layout, True, ("tables",)), ("tables", a._stage_tables, True, ("cells",)),
                ("cells", a._stage_cells, False, ("finish",)), ("finish", self._finish, False, ()))
        self._workers = {spec[0]: (self.rec_lanes if spec[0] == "recognize" else 1) for spec in plan}
        self._threads = [threading.Thread(target=self._loop, args=spec + (lane,), name=f"ymk-{spec[0]}-{lane}" if self._workers[spec[0]] > 1 else f"ymk-{spec[0]}",
                                          daemon=True) for spec in plan for lane in range(self._workers[spec[0]])]
        for t in self._threads:
            t.start()

    # ------------------------------------------------------------------ stage threads
    def _loop(self, name, fn, on_gpu, outs, lane=0):
        stream = None
        if self._workers[name] > 1:  # a stage with several lanes tells its function which one is calling
            inner = fn

            def fn(wave):
                return inner(wave, lane)

        if on_gpu and self._gpu:
            torch.cuda.set_device(self.device)
            stream = torch.cuda.Stream(device=self.device, priority=self.stage_priority.get(name, 0))
        q_in = self._q[name]
        while True:
            wave = q_in.get()
            if wave is _STOP:
                return
            if name == "finish":
                wave.joined += 1
                if wave.joined < 2:  # arrives once from the recognise chain and once from the layout chain
                    continue
            if wave.error is None or name == "finish":
                t_start = time.perf_counter()
                try:
                    if stream is not None:
                        with torch.cuda.stream(stream):
                            stream.wait_event(wave.uploaded)
                            fn(wave)
                            stream.synchronize()
                    else:
                        fn(wave)
                except BaseException as exc:  # noqa: BLE001 - delivered to the page's result slot
                    wave.fail(name, exc)
                    if stream is not None:
                        # kernels the failed stage already queued may still read the wave's pages / pinned maps / crops, which
                        # the finish stage is about to hand back: drain them first (best effort - the device may be the problem)
                        try:
                            stream.synchronize()
                        except Exception:  # noqa: BLE001
                            pass
                    if name == "finish":  # cannot happen short of a bug in _finish itself: never lose a page or a slot
                        for idx in wave.ids:
                            wave.job.results.setdefault(idx, exc)
                        if wave.pages is not None:
                            self._release(wave)
                if self.trace is not None:
                    self.trace.append((name, wave.seq, len(wave), t_start, time.perf_counter()))
            if name == "cells":
                wave.layout_done.set()
            for out in outs:
                self._q[out].put(wave)

    def _boxes(self, wave):
        a = self.analyzer
        a._stage_boxes(wave)
        if a.split_text_across_cells:
            wave.layout_done.wait()
            if wave.error is None:
                a._stage_split(wave)

    def _release(self, wave):
        job = wave.job
        wave.pages = wave.maps = wave.rec_plan = None  # the device pages, the pinned map views and the crop tensors go back
        self._rings.put(wave.ring)
        with job.cond:
            job.outstanding -= 1
            job.cond.notify_all()

    def _finish(self, wave):
        job = wave.job
        if wave.error is not None:
            if len(wave) > 1:
                logger.warning("wave of %d pages failed in stage %s (%s: %s); re-running its pages one by one", len(wave),
                               wave.failed_stage, type(wave.error).__name__, wave.error)
                with job.cond:
                    job.retries.extend(zip(wave.ids, wave.imgs))
            else:
                logger.error("page %d failed in stage %s: %s: %s", wave.ids[0], wave.failed_stage, type(wave.error).__name__,
                             wave.error)
                job.results[wave.ids[0]] = wave.error
        else:
            for k, idx in enumerate(wave.ids):
                try:
                    job.results[idx] = self.analyzer._stage_finish(wave, k)
                except Exception as exc:  # noqa: BLE001 - aggregation is per page: only this page fails
                    logger.error("page %d failed in aggregation: %s: %s", idx, type(exc).__name__, exc)
                    job.results[idx] = exc
        self._release(wave)

    # ------------------------------------------------------------------ caller side
    def _launch(self, job, ids, imgs, retry=False, ring=None, waited=0.0):
        """`ring`: a wave slot the caller already holds (serve() takes the slot BEFORE it pulls the wave's sources; `waited`:
        how long it waited for it), or None: taken here."""
        t_start = time.perf_counter() - waited
        if ring is None:
            ring = self._rings.get()  # blocks while `in_flight` waves are out: back-pressure on decoding and staging
        t_ring = time.perf_counter()
        try:
            if self._gpu:
                pages = [self.stager.upload(img, wait=False) for img in imgs]
                uploaded = torch.cuda.Event()
                uploaded.record(self.stager.stream)
            else:
                pages, uploaded = list(imgs), None
        except BaseException:
            self._rings.put(ring)
            raise
        self._seq += 1
        if self.trace is not None:
            self.trace.append(("wait_slot", self._seq, len(ids), t_start, t_ring))
            self.trace.append(("stage_h2d", self._seq, len(ids), t_ring, time.perf_counter()))
        wave = Wave(self._seq, ids, imgs, pages, ring, uploaded, retry, job)
        with job.cond:
            job.outstanding += 1
            job.waves += 1
        self._q["detect"].put(wave)
        self._q["layout"].put(wave)

    def serve(self, sources: Iterable, with_source: bool = False) -> List:
        """sources: uint8 H x W x 3 BGR arrays and / or image file paths (a multi-frame file contributes one entry per
        frame).  Returns one entry per page, in order: the DocumentAnalyzerSchema, or the exception that page (or
        file) raised.  with_source=True: (source index, frame index within the source, entry) triples instead - what a
        caller that writes `<file>_p<page>.json` per page needs (cli/main.py:122-137)."""
        from .data.functions import load_image

        with self._serve_lock, _full_gc_deferred(self.defer_full_gc), _switch_interval(self.switch_interval):
            job = self._job = _Job()
            n = 0
            origin = []  # page id -> (source index, frame index)
            pend_ids, pend_imgs = [], []

            held = [None, 0.0]  # the wave slot taken for the wave being collected, and how long the caller waited for it

            def flush():
                nonlocal pend_ids, pend_imgs
                if pend_ids:
                    ids, imgs, pend_ids, pend_imgs = pend_ids, pend_imgs, [], []
                    ring, held[0] = held[0], None
                    try:
                        self._launch(job, ids, imgs, ring=ring, waited=held[1] if ring is not None else 0.0)
                    except Exception as exc:  # noqa: BLE001 - e.g. a page that is not uint8 H x W x 3
                        if len(ids) == 1:
                            job.results[ids[0]] = exc
                        else:
                            with job.cond:
                                job.retries.extend(zip(ids, imgs))

            def drain_retries():
                while True:
                    with job.cond:
                        if not job.retries:
                            return
                        idx, img = job.retries.popleft()
                        job.retried_pages += 1
                    try:
                        self._launch(job, [idx], [img], retry=True)
                    except Exception as exc:  # noqa: BLE001
                        job.results[idx] = exc

            # The slot of a wave is taken BEFORE the wave's first source is pulled: `sources` may be a generator that claims
            # work from a shared counter (distributed.PageDealer) - a rank must not claim pages it has no room for yet, or the
            # end of a sharded job is as ragged as a static deal.  (A slot held while nothing is pending is given back below.)
            it = enumerate(sources)
            while True:
                if not pend_ids and held[0] is None:
                    drain_retries()
                    t_wait = time.perf_counter()
                    held[0] = self._rings.get()
                    held[1] = time.perf_counter() - t_wait
                try:
                    si, src = next(it)
                except StopIteration:
                    break
                if isinstance(src, np.ndarray):
                    frames = [src]
                else:
                    try:
                        frames = list(load_image(src))
                    except Exception as exc:  # noqa: BLE001 - cli/main.py:555-564: log, go on with the next file
                        logger.error("cannot load %s: %s: %s", src, type(exc).__name__, exc)
                        job.results[n] = exc
                        origin.append((si, 0))
                        n += 1
                        continue
                for fi, frame in enumerate(frames):
                    origin.append((si, fi))
                    pend_ids.append(n)
                    pend_imgs.append(frame)
                    n += 1
                    if len(pend_ids) == self.wave:
                        flush()
            flush()
            if held[0] is not None:  # the source list ended on a wave boundary
                self._rings.put(held[0])
                held[0] = None
            while True:
                drain_retries()
                with job.cond:
                    if job.outstanding == 0 and not job.retries:
                        break
                    job.cond.wait(timeout=0.05)
            self.last_job = {"pages": n, "waves": job.waves, "retried_pages": job.retried_pages}
            if with_source:
                return [(origin[i][0], origin[i][1], job.results[i]) for i in range(n)]
            return [job.results[i] for i in range(n)]

    def close(self):
        for name, q in self._q.items():
            for _ in range(self._workers[name]):
                q.put(_STOP)
        for t in self._threads:
            t.join(timeout=10)

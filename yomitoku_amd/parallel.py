"""Several pages in flight on ONE GPU.

A single page cannot fill an MI355X: the recogniser's greedy decode is a chain of small dependent
launches, the RT-DETR forward at batch 1 is latency bound, and the host post-processing leaves the
device idle.  Pages are independent (cli/main.py:116-120), so a rank keeps K analyzer replicas
(K x ~0.5 GB of weights out of 288 GB) and K worker threads; each worker owns its replicas' HIP
streams, ctypes releases the GIL inside every library call, and results come back in page order.
This is the intra-GPU half of the page sharding in `yomitoku_amd.distributed` (the inter-GPU half).
Per-page results are identical to the serial path: replicas share nothing mutable.
"""

from __future__ import annotations

import queue
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List


class PageParallel:
    def __init__(self, make_worker: Callable[[int], Callable], n_workers: int = 4):
        """make_worker(i) -> callable(item) that owns everything it mutates (its own analyzer); an item is a page
        or a wave of pages (DocumentAnalyzer.analyze_pages).

        Note for applications: every library call returns into Python by re-taking the GIL; with CPython's default
        5 ms switch interval a worker coming back from a 50 us kernel launch can wait 5 ms behind another worker's
        post-processing loop.  `sys.setswitchinterval(2e-4)` in the APPLICATION helps (bench.py does it); this
        library does not touch process-wide interpreter settings."""
        self.n_workers = int(n_workers)
        self._free: "queue.Queue" = queue.Queue()
        self.workers = [make_worker(i) for i in range(self.n_workers)]
        for w in self.workers:
            self._free.put(w)
        self._pool = ThreadPoolExecutor(max_workers=self.n_workers, thread_name_prefix="ymk-page")

    def _run(self, item):
        w = self._free.get()
        try:
            return w(item)
        finally:
            self._free.put(w)

    def map(self, items: Iterable) -> List:
        """Process every item; results in input order.  A failing page raises after the others finish
        being scheduled (the caller may catch per page, like cli/main.py:555-564 does per file)."""
        return list(self._pool.map(self._run, list(items)))

    def close(self):
        self._pool.shutdown(wait=True)


# --------------------------------------------------------------------------------------------------
# Processes: the host half of a page (box geometry, token decoding, reading order) is Python, and the
# threads of one interpreter share one GIL.  Measured on MI355X: one process with 8 pages in flight
# reaches ~40 pages/s while the device still has idle gaps; two processes with 4 pages each reach ~50.
# So a rank may put helper PROCESSES next to itself on the same GPU - each with its own interpreter,
# HIP context, weights (handed over in CPU shared memory) and PageParallel.  Nothing is shared on the
# device; a page's result does not depend on which process ran it.
def _helper_loop(conn, init_fn, init_args, index):
    try:
        state = init_fn(*init_args, index)
        conn.send(("ready", None))
    except BaseException as exc:  # noqa: BLE001 - the parent re-raises
        conn.send(("error", f"{type(exc).__name__}: {exc}"))
        return
    while True:
        cmd, payload = conn.recv()
        if cmd == "stop":
            break
        try:
            conn.send(("ok", state(payload)))
        except BaseException as exc:  # noqa: BLE001
            conn.send(("error", f"{type(exc).__name__}: {exc}"))
    conn.close()


class PageProcesses:
    """n helper processes; helper i runs `state = init_fn(*init_args, i)` once, then `state(payload)` per
    request.  `init_fn` must be a module-level function (the helpers are spawned, not forked: a forked HIP
    context is unusable).  start()/finish() bracket one round so that the caller can work meanwhile."""

    def __init__(self, init_fn: Callable, init_args: tuple = (), n_procs: int = 1, first_index: int = 1):
        import torch.multiprocessing as mp

        ctx = mp.get_context("spawn")
        self._conns, self._procs = [], []
        for i in range(int(n_procs)):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_helper_loop, args=(child, init_fn, init_args, first_index + i), daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)
        for c in self._conns:  # all helpers initialise concurrently; fail early if one cannot
            self._expect(c, "ready")
        self._pending = False

    @staticmethod
    def _expect(conn, want):
        kind, value = conn.recv()
        if kind == "error":
            raise RuntimeError(f"page helper process failed: {value}")
        assert kind == want, (kind, want)
        return value

    def __len__(self):
        return len(self._procs)

    def start(self, payloads):
        assert not self._pending and len(payloads) == len(self._conns)
        for c, p in zip(self._conns, payloads):
            c.send(("call", p))
        self._pending = True

    def finish(self) -> List:
        assert self._pending
        self._pending = False
        replies = [c.recv() for c in self._conns]  # drain every helper before reporting a failure
        for kind, value in replies:
            if kind == "error":
                raise RuntimeError(f"page helper process failed: {value}")
        return [value for _, value in replies]

    def call(self, payloads) -> List:
        self.start(payloads)
        return self.finish()

    def close(self):
        for c in self._conns:
            try:
                c.send(("stop", None))
            except (BrokenPipeError, OSError):
                pass
        for p in self._procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        self._conns, self._procs = [], []

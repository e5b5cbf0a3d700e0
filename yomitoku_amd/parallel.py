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

import os
import queue
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List


class PageParallel:
    def __init__(self, make_worker: Callable[[int], Callable], n_workers: int = 4, switch_interval: float = 2e-4):
        """make_worker(i) -> callable(page_item) that owns everything it mutates (its own analyzer).

        switch_interval: every library call returns into Python by re-taking the GIL; with CPython's default
        5 ms switch interval a worker coming back from a 50 us kernel launch can wait 5 ms behind another
        worker's post-processing loop, and the device drains meanwhile.  The workers are launch-latency
        bound, so the interval is lowered for the process (None leaves it alone)."""
        if switch_interval is not None:
            sys.setswitchinterval(float(os.environ.get("YMK_SWITCH_INTERVAL", switch_interval)))
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

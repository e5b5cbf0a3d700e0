"""Host-side page parallelism (yomitoku_amd/parallel.py) without a GPU: worker threads keep page order and
worker ownership; helper processes answer rounds in order and surface failures."""
import threading

import pytest

from yomitoku_amd.parallel import PageParallel, PageProcesses


def test_threads_keep_order_and_ownership():
    seen = {}

    def make(i):
        def work(item):
            seen.setdefault(i, set()).add(threading.get_ident())
            return item * item

        return work

    pool = PageParallel(make, n_workers=3)
    assert pool.map(range(20)) == [i * i for i in range(20)]
    pool.close()
    assert all(len(threads) >= 1 for threads in seen.values())


def test_thread_failure_propagates():
    def make(i):
        def work(item):
            if item == 3:
                raise ValueError("bad page")
            return item

        return work

    pool = PageParallel(make, n_workers=2)
    with pytest.raises(ValueError):
        pool.map(range(6))
    pool.close()


def _init_helper(offset, index):
    def handle(payload):
        if payload == "boom":
            raise ValueError("helper failure")
        return [offset + index + v for v in payload]

    return handle


def _init_broken(index):
    raise RuntimeError("cannot start")


def test_helper_processes_round_trip():
    procs = PageProcesses(_init_helper, (100,), n_procs=2, first_index=1)
    assert len(procs) == 2
    assert procs.call([[1, 2], [3]]) == [[102, 103], [105]]
    procs.start([[0], [0]])
    assert procs.finish() == [[101], [102]]
    with pytest.raises(RuntimeError, match="helper failure"):
        procs.call(["boom", [1]])
    procs.close()


def test_helper_init_failure_is_loud():
    with pytest.raises(RuntimeError, match="cannot start"):
        PageProcesses(_init_broken, (), n_procs=1)

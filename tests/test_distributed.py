"""world_size-2 gloo tests (CPU) of the page-sharding helpers: weight broadcast from rank 0 and
ordered gather of per-page results."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yomitoku_amd import distributed as yd

    r, lr, w = yd.init(backend="gloo")
    assert (r, w) == (rank, world)
    sd = None
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        sd = {"a.weight": torch.randn(4, 3, 2, 2, generator=g), "a.bias": torch.randn(4, generator=g),
              "bn.num_batches_tracked": torch.tensor(7, dtype=torch.int64),
              "h.bf16": torch.tensor([0.3359375, -2.5, 1e-3], dtype=torch.bfloat16),
              "h.f64": torch.tensor([1.0 + 2.0 ** -40, -3.0], dtype=torch.float64)}
    got = yd.broadcast_state_dict(sd, src=0, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(3)
    exp_w = torch.randn(4, 3, 2, 2, generator=g)
    exp_b = torch.randn(4, generator=g)
    ok = torch.equal(got["a.weight"], exp_w) and torch.equal(got["a.bias"], exp_b)
    ok = ok and int(got["bn.num_batches_tracked"]) == 7
    ok = ok and list(got) == ["a.weight", "a.bias", "bn.num_batches_tracked", "h.bf16", "h.f64"]
    # half-precision tensors are not truncated to integers, doubles are not narrowed to fp32 on the way
    ok = ok and got["h.bf16"].dtype == torch.bfloat16 and torch.equal(got["h.bf16"], torch.tensor([0.3359375, -2.5, 1e-3], dtype=torch.bfloat16))
    ok = ok and got["h.f64"].dtype == torch.float64 and torch.equal(got["h.f64"], torch.tensor([1.0 + 2.0 ** -40, -3.0], dtype=torch.float64))
    # every rank holds rank 0's bytes, and the report counts the ranks the collective saw
    rep = yd.replica_report({"ck": got}, device=torch.device("cpu"))
    ok = ok and rep["ranks"] == world and rep["backend"] == "gloo" and rep["weights_crc_equal"] is True
    bad = dict(got)
    if rank == 1:
        bad["a.bias"] = got["a.bias"] + 1
    ok = ok and yd.replica_report({"ck": bad}, device=torch.device("cpu"))["weights_crc_equal"] is False
    ok = ok and yd.all_gather_scalars([rank + 0.5], torch.device("cpu")) == [[0.5], [1.5]]
    n_items = 7
    mine = yd.shard_indices(n_items, rank, world)
    merged = yd.gather_in_order([f"page{i}" for i in mine], n_items, rank, world)
    if rank == 0:
        ok = ok and merged == [f"page{i}" for i in range(n_items)]
    else:
        ok = ok and merged is None
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_shard_indices_cover_all_pages():
    from yomitoku_amd.distributed import shard_indices

    for n in (0, 1, 5, 64, 513):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


class _StubAnalyzer:
    """serve() contract of DocumentAnalyzer without a GPU: an entry per page (frame), exceptions in place."""

    def __init__(self, device, checkpoints, budget):
        self.bias = float(checkpoints["net"]["w"].sum())  # proves the broadcast reached make_analyzer
        self.budget = budget
        self.closed = False

    def serve(self, sources, wave=8, in_flight=4, with_source=False, rec_lanes=2):
        out = []
        for si, src in enumerate(sources):
            frames = src if isinstance(src, list) else [src]
            for fi, frame in enumerate(frames):
                entry = ValueError(f"bad page {frame}") if frame < 0 else (frame, self.bias, rec_lanes)
                out.append((si, fi, entry) if with_source else entry)
        return out

    def close(self):
        self.closed = True


def _sharded_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yomitoku_amd import distributed as yd

    ck = (lambda: {"net": {"w": torch.arange(6, dtype=torch.float32)}}) if rank == 0 else None
    # sources: plain pages, one multi-frame file (a list) and one failing page; 7 sources over 2 ranks
    sources = [10, 11, [20, 21, 22], -1, 13, 14, 15]
    server = yd.ShardedServer(_StubAnalyzer, ck, backend="gloo", device="cpu")
    ok = server.analyzer.bias == 15.0 and server.replicas["ranks"] == world and server.replicas["weights_crc_equal"]
    ok = ok and server.shard(len(sources)) == list(range(rank, 7, world))
    ok = ok and server.budget["stage_threads"] == 9 and 1 <= server.budget["box_threads"] <= 4 and server.cores >= 1
    res = server.run(sources, gather="objects", wave=2, in_flight=1)  # dealt dynamically, two sources per claim
    ok = ok and server.last_run["assign"] == "dynamic" and server.last_run["claims"] >= 1
    if rank == 0:
        lanes = server.budget["rec_lanes"]
        ok = ok and [r if isinstance(r, tuple) else type(r).__name__ for r in res] == [
            (10, 15.0, lanes), (11, 15.0, lanes), (20, 15.0, lanes), (21, 15.0, lanes), (22, 15.0, lanes), "ValueError", (13, 15.0, lanes),
            (14, 15.0, lanes), (15, 15.0, lanes)]
    else:
        ok = ok and res is None
    server.close()
    ok = ok and server.analyzer.closed and not torch.distributed.is_initialized()
    q.put((rank, bool(ok)))


def test_sharded_server_world2():
    """ShardedServer / serve_sharded: init -> core slice -> broadcast -> analyzer -> shard by source -> serve -> ordered gather
    with a failing page and a multi-frame source, two gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_serve_sharded_single_process():
    from yomitoku_amd import distributed as yd

    env = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    try:
        res = yd.serve_sharded([1, 2, [3, 4]], _StubAnalyzer, {"net": {"w": torch.ones(3)}}, backend="gloo", wave=4, gather="objects")
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
    assert [r[0] for r in res] == [1, 2, 3, 4] and res[0][1] == 3.0


def test_thread_budget_follows_the_core_slice():
    from yomitoku_amd.distributed import thread_budget

    assert thread_budget(128) == {"stage_threads": 9, "box_threads": 4, "rec_lanes": 2}
    assert thread_budget(16) == {"stage_threads": 9, "box_threads": 4, "rec_lanes": 2}   # 8 ranks on a 128-core host
    assert thread_budget(6) == {"stage_threads": 9, "box_threads": 1, "rec_lanes": 1}


class _PickyError(Exception):
    """An exception whose constructor wants two arguments: pickles, then fails to UNpickle (args holds one string)."""

    def __init__(self, code, what):
        super().__init__(f"{what} ({code})")
        self.code = code


class _FailingAnalyzer(_StubAnalyzer):
    def __init__(self, device, checkpoints, budget):
        super().__init__(device, checkpoints, budget)
        mode = os.environ["YMK_TEST_FAIL"]
        if mode == "build" and int(os.environ["RANK"]) == 1:
            raise _PickyError(7, "no such device")
        self.mode = mode

    def serve(self, sources, **kw):
        if self.mode == "serve" and int(os.environ["RANK"]) == 1:
            raise _PickyError(9, "serve blew up")
        out = super().serve(sources, **kw)
        if self.mode == "entry":
            out = [(si, fi, _PickyError(3, "page") if not isinstance(e, tuple) else e) for si, fi, e in out]
        return out


def _failing_worker(rank, world, port, q, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), YMK_TEST_FAIL=mode)
    from yomitoku_amd import distributed as yd

    def ck():
        if mode == "checkpoints":
            raise FileNotFoundError("model.safetensors")
        return {"net": {"w": torch.arange(6, dtype=torch.float32)}}

    sources = [10, 11, -1, 13]
    try:
        res = yd.serve_sharded(sources, _FailingAnalyzer, ck if rank == 0 else None, backend="gloo", wave=2, gather="objects")
        outcome = ("ok", [e if isinstance(e, tuple) else f"{type(e).__name__}: {e}" for e in res] if res is not None else None)
    except yd.ShardedJobError as exc:
        outcome = ("job-error", sorted(exc.failures), str(exc))
    q.put((rank, outcome, torch.distributed.is_initialized()))


def _run_failing(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]  # a rank left waiting in a collective would time this out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return {r: (o, init) for r, o, init in got}


def test_a_failing_rank_reaches_every_collective_and_all_ranks_raise():
    """whatever fails on one rank - rank 0's checkpoint source, a rank's analyzer constructor, `serve` itself -
    no rank is left blocked in a collective; every rank raises ShardedJobError naming the culprit and leaves the group."""
    res = _run_failing("checkpoints")
    for rank in (0, 1):
        (kind, ranks, text), still_init = res[rank]
        assert kind == "job-error" and ranks == [0] and "FileNotFoundError" in text and not still_init
    res = _run_failing("build")
    for rank in (0, 1):
        (kind, ranks, text), still_init = res[rank]
        assert kind == "job-error" and ranks == [1] and "_PickyError: no such device (7)" in text and not still_init
    res = _run_failing("serve")
    for rank in (0, 1):
        (kind, ranks, text), still_init = res[rank]
        assert kind == "job-error" and ranks == [1] and "serve blew up (9)" in text and not still_init


def test_an_exception_entry_that_cannot_be_unpickled_still_arrives():
    res = _run_failing("entry")
    (kind, entries), still_init = res[0]
    assert kind == "ok" and not still_init
    assert len(entries) == 4 and entries[2] == "RuntimeError: _PickyError: page (3)"
    assert [e[0] for i, e in enumerate(entries) if i != 2] == [10, 11, 13]
    assert res[1][0] == ("ok", None)


def test_portable_entry_keeps_what_pickles():
    from yomitoku_amd.distributed import portable_entry

    e = ValueError("bad page")
    assert portable_entry(e) is e and portable_entry({"a": 1}) == {"a": 1}
    p = portable_entry(_PickyError(3, "page"))
    assert type(p) is RuntimeError and str(p) == "_PickyError: page (3)"


def _forms_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import json

    from yomitoku_amd import distributed as yd

    ck = (lambda: {"net": {"w": torch.arange(6, dtype=torch.float32)}}) if rank == 0 else None
    sources = [10, 11, -1, 13, 14]
    server = yd.ShardedServer(_StubAnalyzer, ck, backend="gloo", device="cpu")
    as_json = server.run(sources, wave=2)  # the default form: JSON text to rank 0
    mine = server.run(sources, gather=None, assign="static", wave=2)
    server.close()
    ok = True
    if rank == 0:
        decoded = [json.loads(t) for t in as_json]
        ok = ok and [d[0] if isinstance(d, list) else d for d in decoded] == [10, 11, {"error": "ValueError: bad page -1"}, 13, 14]
    else:
        ok = ok and as_json is None
    ok = ok and [(si, fi) for si, fi, _ in mine] == [(i, 0) for i in range(rank, 5, world)]
    q.put((rank, bool(ok)))


def test_gather_forms_world2():
    """ShardedServer.run(gather="json"): rank 0 receives text it does not have to rebuild; gather=None: every rank keeps its own
    triples and no result travels (the per-page path of an 8-GPU node without rank 0 in it)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_forms_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------- dynamic page assignment
class _TimedAnalyzer:
    """A stub whose pages COST something: `heavy` pages (multiples of the world size: the whole static share of rank 0) take
    ten times as long as the others - the 8 ... 30 ms spread of real pages with and without tables.  Reads its sources
    lazily, a wave at a time, as DocumentAnalyzer.serve does."""

    def __init__(self, device, checkpoints, budget):
        self.world = int(os.environ["WORLD_SIZE"])
        self.light = float(os.environ.get("YMK_TEST_LIGHT_S", "0.004"))

    def serve(self, sources, wave=8, in_flight=4, with_source=False, rec_lanes=2):
        import time

        out = []
        for si, page in enumerate(sources):
            time.sleep(self.light * (10 if page % self.world == 0 else 1))
            out.append((si, 0, {"page": page}))
        self.finished = time.time()
        return out

    def close(self):
        pass


def _dealer_worker(rank, world, port, q, n_sources, wave):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import json
    import time

    from yomitoku_amd import distributed as yd

    server = yd.ShardedServer(_TimedAnalyzer, None, backend="gloo", device="cpu", pin_cores=False)
    sources = list(range(n_sources))
    out = {}
    for assign in ("dynamic", "static"):
        server.barrier()
        t0 = time.time()
        res = server.run(sources, assign=assign, wave=wave)
        mine = server.last_run["sources"]
        out[assign] = {"seconds": server.analyzer.finished - t0, "sources": mine, "claims": server.last_run["claims"],
                       "pages": [json.loads(t)["page"] for t in res] if rank == 0 else None}
    again = server.run(sources, wave=wave)  # a second dynamic job deals from a fresh counter
    out["again"] = [json.loads(t)["page"] for t in again] if rank == 0 else None
    server.close()
    q.put((rank, out))


def _run_dealer(world, n_sources, wave):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dealer_worker, args=(r, world, port, q, n_sources, wave)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _check_dealing(world, n_sources, wave, light=0.004):
    got = _run_dealer(world, n_sources, wave)
    for assign in ("dynamic", "static"):
        assert got[0][assign]["pages"] == list(range(n_sources))          # every page once, in order, on rank 0
        assert sum(got[r][assign]["sources"] for r in range(world)) == n_sources
    assert got[0]["again"] == list(range(n_sources))
    dyn = [got[r]["dynamic"]["seconds"] for r in range(world)]
    sta = [got[r]["static"]["seconds"] for r in range(world)]
    # static: rank 0 owns every heavy page (10 x) and finishes long after the others; dynamic: the ranks finish within about
    # one chunk of each other - a chunk is `wave` pages, at most ceil(wave / world) of them heavy
    heavy_per_chunk = -(-wave // world)
    one_chunk = light * (heavy_per_chunk * 10 + (wave - heavy_per_chunk))
    assert max(dyn) - min(dyn) <= 2.0 * one_chunk + 0.15, (dyn, one_chunk)
    assert max(sta) - min(sta) > 3.0 * (max(dyn) - min(dyn)) and max(dyn) < 0.8 * max(sta), (dyn, sta)
    # the busy rank asked less often: claims differ, and every rank claimed once more than it was served (the miss that ends its loop)
    claims = [got[r]["dynamic"]["claims"] for r in range(world)]
    assert sum(claims) == -(-n_sources // wave) + world, claims


def test_pages_are_pulled_not_dealt_up_front_world2():
    _check_dealing(2, 96, 4)


def test_pages_are_pulled_not_dealt_up_front_world8():
    _check_dealing(8, 512, 8)

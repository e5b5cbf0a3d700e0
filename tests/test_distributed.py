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

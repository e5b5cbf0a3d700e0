"""Page sharding across the GPUs of one node.

The reference has no distributed code: pages are independent units (cli/main.py:116-120 loops
them sequentially).  Here every rank (one process per GPU) holds a full replica of the models and
takes a strided share of the pages; the only collective is a one-off broadcast of the packed
checkpoint from rank 0 (RCCL over xGMI when the backend is "nccl"), so that weights are read /
generated once.  There is no collective on the per-page path.
"""

from __future__ import annotations

import os
from collections import OrderedDict
from typing import List, Mapping, Sequence

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin page assignment: item i goes to rank i % world (keeps ranks within one page
    of each other for any n_items)."""
    return list(range(rank, n_items, world))


def gather_in_order(local: Sequence, n_items: int, rank: int, world: int) -> list | None:
    """Inverse of shard_indices for picklable per-page results: rank 0 gets the full list in page
    order (host-side gather of Python objects; not a data-path collective)."""
    if world == 1:
        return list(local)
    out = [None] * world if rank == 0 else None
    dist.gather_object(list(local), out, dst=0)
    if rank != 0:
        return None
    merged = [None] * n_items
    for r, part in enumerate(out):
        for j, idx in enumerate(shard_indices(n_items, r, world)):
            merged[idx] = part[j]
    return merged


def broadcast_state_dict(sd: Mapping[str, torch.Tensor] | None, src: int = 0, device=None):
    """Broadcast a checkpoint from `src` as ONE flat fp32 message (+ a tiny int64 one).

    Rank `src` passes the state dict, the others pass None and receive an identical copy (CPU
    tensors).  The flat buffer lives on `device` during the collective: with the nccl backend that
    is the rank's GPU, so the ~0.1-0.5 GB message travels over xGMI, not through the host.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sd
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    def kind(d):  # which flat message a tensor travels in: fp32 (incl. f16 / bf16, widened losslessly), fp64, or int64
        dt = getattr(torch, d)
        return "f64" if dt == torch.float64 else ("f32" if dt.is_floating_point else "i64")

    wire = {"f32": torch.float32, "f64": torch.float64, "i64": torch.int64}
    numel = {k: sum(int(torch.Size(s).numel()) for _, s, d in meta if kind(d) == k) for k in wire}
    bufs = {k: torch.empty(max(numel[k], 1), dtype=dt, device=device) for k, dt in wire.items()}
    if rank == src:
        off = dict.fromkeys(wire, 0)
        for k, s, d in meta:
            v, c = sd[k], kind(d)
            n = v.numel()
            bufs[c][off[c] : off[c] + n] = v.reshape(-1).to(device=device, dtype=wire[c])
            off[c] += n
    for c in wire:
        if numel[c]:
            dist.broadcast(bufs[c], src=src)
    if rank == src:
        return sd
    out = OrderedDict()
    host = {c: b.cpu() for c, b in bufs.items()}
    off = dict.fromkeys(wire, 0)
    for k, s, d in meta:
        n, c = int(torch.Size(s).numel()), kind(d)
        out[k] = host[c][off[c] : off[c] + n].reshape(s).to(getattr(torch, d)).clone()
        off[c] += n
    return out


def state_dict_crc(sd: Mapping[str, torch.Tensor]) -> int:
    """CRC-32 over the bytes of every tensor in key order - of what this rank actually HOLDS after the broadcast."""
    import zlib

    crc = 0
    for k, v in sd.items():
        crc = zlib.crc32(k.encode(), crc)
        t = v.detach().cpu().contiguous()
        if t.dtype == torch.bfloat16:  # numpy has no bfloat16: hash its bit pattern
            t = t.view(torch.int16)
        crc = zlib.crc32(t.numpy().tobytes(), crc)
    return crc


def all_gather_scalars(values: Sequence[float], device=None) -> List[List[float]]:
    """Every rank's `values` on every rank (one tiny all-gather; world size 1: [[values]]).  float64 on the wire, so
    32-bit CRCs and counters travel exactly."""
    vals = [float(v) for v in values]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [vals]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(vals, dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [p.cpu().tolist() for p in parts]


def replica_report(sds: Mapping[str, Mapping[str, torch.Tensor]], device=None) -> dict:
    """Proof that the collective saw every rank and that every rank holds rank 0's weights: world size and backend as
    torch.distributed reports them, and whether the per-checkpoint CRCs of all ranks agree."""
    names = sorted(sds)
    crcs = all_gather_scalars([state_dict_crc(sds[k]) for k in names], device)
    world = len(crcs)
    return {"ranks": world, "backend": dist.get_backend() if dist.is_initialized() else None,
            "weights_crc_equal": all(c == crcs[0] for c in crcs),
            "weights_crc": {k: f"{int(v):08x}" for k, v in zip(names, crcs[0])}}

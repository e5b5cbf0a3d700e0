"""Page sharding across the GPUs of one node.

The reference has no distributed code: pages are independent units (cli/main.py:116-120 loops
them sequentially).  Here every rank (one process per GPU) holds a full replica of the models and
PULLS its pages in chunks of one wave from a counter that rank 0 hosts (a TCPStore word: one atomic
add per chunk, host sockets only) - a page costs anything between 8 and 30 ms depending on its
tables, so a fixed share per rank would end ragged; the only collective is a one-off broadcast of
the packed checkpoint from rank 0 (RCCL over xGMI when the backend is "nccl"), so that weights
are read / generated once.  There is no collective on the per-page path.
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
    of each other for any n_items).  The static form (`ShardedServer.run(assign="static")`): page COUNTS are level, page
    costs are not - `PageDealer` is what the sharded job uses by default."""
    return list(range(rank, n_items, world))


class PageDealer:
    """Dynamic page assignment for the ranks of one node: chunk k of the source list belongs to whichever rank asks for
    "the next chunk" k-th.  The counter is one word of a TCPStore that rank 0 hosts (host sockets; `add` is atomic on the
    server): no collective, nothing on the GPU path, and a rank that is busy with table-heavy pages simply asks less often.

        dealer = PageDealer(rank, world)            # once per process group (rank 0 hosts, the port travels by broadcast)
        for first, stop in dealer.chunks(job, n_sources, chunk):   # on every rank, lazily: a claim per chunk
            ...

    World size 1 needs no store: the chunks are dealt in order by a local counter."""

    def __init__(self, rank: int, world: int, host: str | None = None):
        self.rank, self.world = rank, world
        self.store = None
        self._local = {}
        self.claims = 0
        self.error = None  # a rank that could not reach the counter says so here (ShardedServer agrees on it): it never raises alone
        if world > 1:
            from datetime import timedelta

            host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = [None]
            if rank == 0:
                try:  # port 0: the server picks a free one; the number travels to the other ranks through the process group
                    self.store = dist.TCPStore(host, 0, world, True, timeout=timedelta(seconds=300), wait_for_workers=False)
                    port[0] = ("port", self.store.port)
                except Exception as exc:  # noqa: BLE001 - told to the ranks waiting in the broadcast below
                    port[0] = ("error", _describe(exc))
            dist.broadcast_object_list(port, src=0)
            if port[0][0] == "error":  # the same on every rank
                raise RuntimeError("rank 0 could not host the page counter: " + port[0][1])
            if rank != 0:
                try:
                    self.store = dist.TCPStore(host, int(port[0][1]), world, False, timeout=timedelta(seconds=300))
                except Exception as exc:  # noqa: BLE001
                    self.error = _describe(exc)

    def claim(self, job: int, chunk: int) -> int:
        """First source index of the next unclaimed chunk of `job` (may lie beyond the job's end: then there is none left)."""
        self.claims += 1
        if self.store is None:
            first = self._local.get(job, 0)
            self._local[job] = first + chunk
            return first
        return int(self.store.add(f"ymk/job{job}/next", chunk)) - chunk

    def chunks(self, job: int, n_sources: int, chunk: int):
        chunk = max(1, int(chunk))
        while True:
            first = self.claim(job, chunk)
            if first >= n_sources:
                return
            yield first, min(first + chunk, n_sources)


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


_WIRE = {"f32": torch.float32, "f64": torch.float64, "i64": torch.int64}


def _wire_kind(dtype_name: str) -> str:
    """Which flat message a tensor travels in: fp32 (incl. f16 / bf16, widened losslessly), fp64, or int64."""
    dt = getattr(torch, dtype_name)
    return "f64" if dt == torch.float64 else ("f32" if dt.is_floating_point else "i64")


def _checkpoint_meta(sd: Mapping[str, torch.Tensor]) -> list:
    return [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]


def _flat_buffers(meta, device, sd: Mapping[str, torch.Tensor] | None = None) -> dict:
    """One flat buffer per wire type on `device`, sized from `meta`; filled from `sd` where one is given (the sender)."""
    numel = {c: sum(int(torch.Size(s).numel()) for _, s, d in meta if _wire_kind(d) == c) for c in _WIRE}
    bufs = {c: torch.empty(max(numel[c], 1), dtype=dt, device=device) for c, dt in _WIRE.items()}
    if sd is not None:
        off = dict.fromkeys(_WIRE, 0)
        for k, s, d in meta:
            v, c = sd[k], _wire_kind(d)
            n = v.numel()
            bufs[c][off[c] : off[c] + n] = v.reshape(-1).to(device=device, dtype=_WIRE[c])
            off[c] += n
    return {c: (b, numel[c]) for c, b in bufs.items()}


def _unflatten(bufs: dict, meta) -> "OrderedDict[str, torch.Tensor]":
    """The receiver's side: CPU tensors of the sender's shapes and dtypes out of the flat buffers."""
    out = OrderedDict()
    host = {c: b.cpu() for c, (b, _) in bufs.items()}
    off = dict.fromkeys(_WIRE, 0)
    for k, s, d in meta:
        n, c = int(torch.Size(s).numel()), _wire_kind(d)
        out[k] = host[c][off[c] : off[c] + n].reshape(s).to(getattr(torch, d)).clone()
        off[c] += n
    return out


def _collective_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_state_dict(sd: Mapping[str, torch.Tensor] | None, src: int = 0, device=None):
    """Broadcast a checkpoint from `src` as ONE flat fp32 message (+ a tiny int64 one).

    Rank `src` passes the state dict, the others pass None and receive an identical copy (CPU
    tensors).  The flat buffer lives on `device` during the collective: with the nccl backend that
    is the rank's GPU, so the ~0.1-0.5 GB message travels over xGMI, not through the host.
    Without a process group there is nobody to send to and `sd` comes straight back; a group of ONE
    rank still runs the collective (that is how the RCCL path is exercised on a single-GPU box).
    """
    if not dist.is_initialized():
        return sd
    rank = dist.get_rank()
    meta = [_checkpoint_meta(sd) if rank == src else None]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    if device is None:
        device = _collective_device()
    bufs = _exchange(_flat_buffers(meta, device, sd if rank == src else None), src)
    return sd if rank == src else _unflatten(bufs, meta)


def _exchange(bufs: dict, src: int) -> dict:
    """The collectives proper: nothing here can fail on one rank alone (buffers exist on every rank when it is entered)."""
    for c, (buf, numel) in bufs.items():
        if numel:
            dist.broadcast(buf, src=src)
    return bufs


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
    if not dist.is_initialized():
        return [vals]
    if device is None:
        device = _collective_device()
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


# ------------------------------------------------------------------------------------------------ the sharded job
def claim_core_slice(local_rank: int, local_world: int):
    """Pin this process to its 1 / local_world share of the cores it may run on (Linux; elsewhere a no-op) and return the
    share's size.  One rank keeps ~15 threads runnable (nine pipeline stages, two recogniser lanes, the box-extraction pool):
    eight ranks wandering over each other's cores is what this prevents."""
    if not hasattr(os, "sched_getaffinity"):
        return os.cpu_count() or 1
    cores = sorted(os.sched_getaffinity(0))
    if local_world <= 1:
        return len(cores)
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per : (local_rank + 1) * per] or cores
    os.sched_setaffinity(0, set(mine))
    return len(mine)


def thread_budget(cores: int) -> dict:
    """Host threads of one rank for a share of `cores` cores.  The nine stage threads of DocumentAnalyzer.serve mostly sleep
    in library calls (GIL released), so they stay; what scales with the share is the C++ box-extraction pool (4 at >= 16
    cores, never more than a quarter of the share) and the second recogniser lane (dropped below 8 cores, where its extra
    stream only adds runnable threads)."""
    return {"stage_threads": 9, "box_threads": max(1, min(4, cores // 4)), "rec_lanes": 2 if cores >= 8 else 1}


class ShardedJobError(RuntimeError):
    """A sharded job that failed as a whole (not a page: pages fail in place): raised on EVERY rank after the collective the
    failure was carried through, so no rank is left waiting in one.  `failures`: {rank: "ExceptionType: message"}."""

    def __init__(self, what: str, failures: Mapping[int, str]):
        self.failures = dict(failures)
        super().__init__(f"{what}: " + "; ".join(f"rank {r}: {m}" for r, m in sorted(self.failures.items())))

    def __reduce__(self):
        return (RuntimeError, (str(self),))


def _describe(exc: BaseException) -> str:
    return f"{type(exc).__name__}: {exc}"


def portable_entry(entry):
    """A per-page entry as it can cross a process boundary: schemas pickle; an exception object is kept when it survives a
    pickle round trip and becomes RuntimeError("Type: message") otherwise (an exception class with a custom __init__ -
    a C-ABI error carrying a code, say - may pickle and then fail to UNpickle on rank 0, inside gather_object)."""
    if not isinstance(entry, BaseException):
        return entry
    import pickle

    try:
        back = pickle.loads(pickle.dumps(entry))
        if type(back) is type(entry):
            return entry
    except Exception:  # noqa: BLE001 - anything the round trip throws means "not portable"
        pass
    return RuntimeError(_describe(entry))


def _entry_json(entry) -> str:
    """A page's entry as JSON text: the schema's own serialisation, or {"error": "Type: message"} for a failed page."""
    if isinstance(entry, BaseException):
        import json

        return json.dumps({"error": _describe(entry)}, ensure_ascii=False)
    dump = getattr(entry, "model_dump_json", None)
    if dump is not None:
        return dump()
    import json

    return json.dumps(entry, ensure_ascii=False, default=str)


class ShardedServer:
    """The page loop of cli/main.py:116-120 over the GPUs of one node - what north_star calls "pages shard naturally (one page
    per GPU) ... with RCCL broadcast of weights": ONE process per GPU (torchrun, or any launcher that sets RANK / LOCAL_RANK /
    WORLD_SIZE), each a full replica.

        server = ShardedServer(make_analyzer, checkpoints)      # init -> core slice -> broadcast -> replica report -> analyzer
        results = server.run(sources, wave=16, in_flight=4)      # deal (PageDealer) -> DocumentAnalyzer.serve -> ordered gather (rank 0)
        server.close()

    make_analyzer(device, checkpoints, budget) builds this rank's DocumentAnalyzer from the broadcast checkpoints ({name:
    state dict}; `checkpoints` is that mapping - or a callable returning it - on rank 0 and ignored elsewhere).  Sources are
    dealt BY SOURCE (a multi-frame file stays on one rank), a wave's worth at a time, to whichever rank asks next; an entry that failed stays an exception object in
    its place, exactly as `serve` reports it, and cannot hold the other ranks (nothing on the per-page path communicates).

    Failures of the JOB (rank 0 cannot produce the checkpoints, a rank cannot build its analyzer, `serve` itself raises on a
    rank) are carried THROUGH the next collective instead of skipping it: every rank reaches the same broadcast / all-gather /
    gather, then every rank raises `ShardedJobError` naming the ranks that failed - nobody is left blocked in a collective
    its peer never enters, and `close()` after such a failure tears the group down without a barrier."""

    def __init__(self, make_analyzer, checkpoints=None, backend: str | None = None, device=None, pin_cores: bool = True):
        self.rank, self.local_rank, self.world = init(backend)
        self.failed = False
        self.analyzer = None
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
        self.cores = claim_core_slice(self.local_rank, local_world) if pin_cores else (os.cpu_count() or 1)
        self.budget = thread_budget(self.cores)
        if device is None:
            device = torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        sds, head, prepared = None, [None], {}
        if self.rank == 0:
            try:
                sds = checkpoints() if callable(checkpoints) else checkpoints
                if sds is not None:  # everything that can fail on the sender alone happens BEFORE the first collective: the
                    for k in sds:    # flat messages are packed here, and a failure travels in `head` like a missing file does
                        meta = _checkpoint_meta(sds[k])
                        prepared[k] = (meta, _flat_buffers(meta, self.device, sds[k]))
                head[0] = ("names", [(k, prepared[k][0]) for k in sds] if sds is not None else None)
            except Exception as exc:  # noqa: BLE001 - carried to the other ranks through the broadcast they are waiting in
                head[0] = ("error", _describe(exc))
        if self.world > 1:
            dist.broadcast_object_list(head, src=0)
        if head[0][0] == "error":
            self.failed = True
            raise ShardedJobError("rank 0 could not produce the checkpoints", {0: head[0][1]})
        names = head[0][1]
        self.checkpoints = None
        if names is not None:
            error = None
            if self.rank != 0:
                try:
                    prepared = {k: (meta, _flat_buffers(meta, self.device, None)) for k, meta in names}
                except Exception as exc:  # noqa: BLE001 - a receiver that cannot hold the message says so before anybody sends
                    error = _describe(exc)
            self._agree("allocating the broadcast buffers", error)
            if dist.is_initialized():
                for k, _ in names:
                    _exchange(prepared[k][1], 0)
            self.checkpoints = OrderedDict((k, sds[k] if self.rank == 0 else _unflatten(prepared[k][1], meta)) for k, meta in names)
            prepared = None
        self.replicas = replica_report(self.checkpoints, self.device) if self.checkpoints else None
        error = None
        try:
            self.analyzer = make_analyzer(self.device, self.checkpoints, self.budget)
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            error = _describe(exc)
        try:
            self._agree("building the analyzer", error)
        except ShardedJobError:
            self._close_analyzer()  # a rank that DID build one gives its workspaces (tens of GB) back before it leaves
            raise
        self.dealer = PageDealer(self.rank, self.world)
        self._agree("connecting to the page counter", self.dealer.error)
        self._jobs = 0
        self.last_run = None

    def _close_analyzer(self):
        close = getattr(self.analyzer, "close", None)
        if close is not None:
            try:
                close()
            except Exception:  # noqa: BLE001 - on the way out of a failed job
                pass

    def _agree(self, what: str, error: str | None):
        """One tiny all-gather of every rank's (ok | error text): raises ShardedJobError on ALL ranks when any rank failed."""
        reports = [error]
        if self.world > 1:
            reports = [None] * self.world
            dist.all_gather_object(reports, error)
        failures = {r: m for r, m in enumerate(reports) if m is not None}
        if failures:
            self.failed = True
            raise ShardedJobError(f"{what} failed", failures)

    def shard(self, n_sources: int) -> List[int]:
        return shard_indices(n_sources, self.rank, self.world)

    def serve_local(self, sources: Sequence, assign: str = "dynamic", chunk: int | None = None, **serve_kwargs) -> list:
        """This rank's share through DocumentAnalyzer.serve: [(global source index, frame index, entry)], in order.
        assign "dynamic" (default): the share is whatever this rank PULLS - `serve` reads its sources lazily (a new wave
        only when a wave slot is free), and every `chunk` sources (default: one wave) the generator claims the next chunk
        from the node's PageDealer; "static": source i belongs to rank i % world, known up front."""
        if assign not in ("dynamic", "static"):
            raise ValueError(f"assign must be 'dynamic' or 'static', got {assign!r}")
        serve_kwargs.setdefault("rec_lanes", self.budget["rec_lanes"])
        self._jobs += 1  # run() is called by every rank for every job: the job number is the same everywhere
        if assign == "static":
            mine = self.shard(len(sources))
            feed = [sources[i] for i in mine]
            claims0 = self.dealer.claims
        else:
            mine, claims0 = [], self.dealer.claims
            size = int(chunk) if chunk else int(serve_kwargs.get("wave", 16))

            def pull():
                for first, stop in self.dealer.chunks(self._jobs, len(sources), size):
                    for i in range(first, stop):
                        mine.append(i)
                        yield sources[i]

            feed = pull()
        local = self.analyzer.serve(feed, with_source=True, **serve_kwargs)
        self.last_run = {"assign": assign, "sources": len(mine), "claims": self.dealer.claims - claims0}
        return [(mine[si], fi, portable_entry(entry)) for si, fi, entry in local]

    def gather(self, local, form: str = "json") -> list | None:
        """Every rank's (source, frame, entry) triples on rank 0, ordered by (source, frame): the entries alone are returned
        there, None elsewhere.  A host-side gather of Python objects, not a data-path collective.  `local` may also be
        {"error": text} - a rank whose share failed as a whole: the collectives still run on every rank and then all of them
        raise ShardedJobError.

        form "objects": the entries as they are (schemas and exception objects pickle).  Rank 0 then rebuilds every remote
        page's schema - a few thousand pydantic objects, 0.4-0.6 ms per page on an idle core, and the whole gather of a
        1024-page job with 8 ranks measured 6.7 s on an 8-core host (tools/host_rehearsal.py): at 8 GPUs x 110 pages/s that is
        rank 0's whole budget - so "json": every schema travels as its `model_dump_json()` text (0.17 ms per page on the
        sender, NOTHING to rebuild on rank 0: 1.1 s for the same job; a failed page as {"error": "Type: message"}) - the form
        for a caller that writes the pages out (cli/main.py:122-137 writes one JSON file per page).  The cyclic garbage
        collector is held back while rank 0 unpickles (a third of the rebuild time)."""
        if form not in ("objects", "json"):
            raise ValueError(f"gather form must be 'objects' or 'json', got {form!r}")
        failed = local.get("error") if isinstance(local, dict) else None
        if form == "json" and failed is None:
            try:
                local = [(si, fi, _entry_json(entry)) for si, fi, entry in local]
            except Exception as exc:  # noqa: BLE001 - a schema that cannot be serialised fails this rank's share, IN the collective below
                failed = _describe(exc)
                local = {"error": failed}
        parts = [local if isinstance(local, dict) else list(local)]
        if self.world > 1:
            # all ranks learn about a failed rank (all_gather of one short string), rank 0 alone receives the results
            self._agree("serving this rank's share", failed)
            parts = [None] * self.world if self.rank == 0 else None
            import gc

            was_enabled = gc.isenabled()
            gc.disable()
            try:
                dist.gather_object(list(local), parts, dst=0)
            finally:
                if was_enabled:
                    gc.enable()
            if self.rank != 0:
                return None
        elif failed is not None:
            self._agree("serving this rank's share", failed)
        merged = sorted((t for part in parts for t in part), key=lambda t: (t[0], t[1]))
        return [entry for _, _, entry in merged]

    def run(self, sources: Sequence, gather: str | None = "json", assign: str = "dynamic", chunk: int | None = None, **serve_kwargs) -> list | None:
        """deal -> serve -> gather.  assign / chunk: see `serve_local`.  gather: "json" (default: rank 0 gets every page as the
        JSON text of its DocumentAnalyzerSchema - what a caller that writes the pages out wants, and nothing for rank 0 to
        rebuild: see `gather`), "objects" (rank 0 gets the schema / exception objects themselves), or None - no result travels at
        all: EVERY rank returns its own [(global source index, frame index, entry)] and writes / forwards them itself, which is
        how a node of 8 GPUs keeps rank 0 out of the per-page path; a failed rank still fails the job on every rank."""
        try:
            local = self.serve_local(sources, assign=assign, chunk=chunk, **serve_kwargs)
        except Exception as exc:  # noqa: BLE001 - the job failed on this rank: say so IN the collective the others will enter
            local = {"error": _describe(exc)}
        if gather is None:
            self._agree("serving this rank's share", local.get("error") if isinstance(local, dict) else None)
            return local
        return self.gather(local, form=gather)

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def close(self, destroy_group: bool = True):
        close = getattr(self.analyzer, "close", None)
        if close is not None:
            close()
        self.dealer = None  # (rank 0's store server goes with it)
        if destroy_group and self.world > 1 and dist.is_initialized():
            if not self.failed:  # after a job failure the peers may already be gone: no rendezvous, just leave
                dist.barrier()
            dist.destroy_process_group()


def serve_sharded(sources: Sequence, make_analyzer, checkpoints=None, backend: str | None = None, **serve_kwargs):
    """One call per rank: the whole sharded job (ShardedServer: init -> broadcast -> shard -> serve -> ordered gather).
    Returns every page's entry in source order on rank 0, None on the other ranks."""
    server = None
    try:
        server = ShardedServer(make_analyzer, checkpoints, backend=backend)
        return server.run(sources, **serve_kwargs)
    except ShardedJobError:
        if server is not None:
            server.failed = True
        raise
    finally:
        if server is not None:
            server.close()
        elif dist.is_initialized():  # the constructor raised on every rank (ShardedJobError): nobody waits in a barrier
            dist.destroy_process_group()

"""The RCCL leg of the sharded job on the hardware that is there: a process group of ONE rank with backend "nccl" (a GPU box
of the test pool has one GPU, and RCCL refuses two ranks on one device).  What this does exercise: RCCL loading and
initialising a communicator on gfx950 with the environment the launcher sets, the flat device buffers a checkpoint travels
in (pack on the GPU -> ncclBroadcast -> unpack), the float64 all-gather of the replica report, the barrier - everything of
yomitoku_amd.distributed that is backend-specific.  What it cannot: a second rank (tests/test_distributed.py does that over
gloo) and xGMI.  Runs in a subprocess so that the process group never leaks into the test session."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, os, sys
import torch
import torch.distributed as dist

from yomitoku_amd import distributed as yd
from yomitoku_amd.utils.synth import dbnet_state_dict

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
sd = dict(dbnet_state_dict(5))
sd["extra.bf16"] = torch.tensor([0.3359375, -2.5, 1e-3], dtype=torch.bfloat16)
sd["extra.f64"] = torch.tensor([1.0 + 2.0 ** -40, -3.0], dtype=torch.float64)
sd["extra.i64"] = torch.tensor([2 ** 40 + 1, -7], dtype=torch.int64)
meta = yd._checkpoint_meta(sd)
bufs = yd._flat_buffers(meta, dev, sd)
moved = 0
for kind, (buf, numel) in bufs.items():
    assert buf.is_cuda
    if numel:
        dist.broadcast(buf, src=0)
        moved += numel * buf.element_size()
torch.cuda.synchronize()
got = yd._unflatten(bufs, meta)
same = list(got) == list(sd) and all(got[k].dtype == sd[k].dtype and got[k].shape == sd[k].shape and torch.equal(got[k], sd[k]) for k in sd)
passthrough = yd.broadcast_state_dict(sd, src=0) is sd          # the sender's own call, through RCCL as well
report = yd.replica_report({"dbnet": got})
scalars = yd.all_gather_scalars([1.5, 4294967295.0, -0.0])


class Stub:
    def __init__(self, device, checkpoints, budget):
        self.device, self.names = device, list(checkpoints)

    def serve(self, sources, with_source=False, **kw):
        return [(i, 0, f"page:{s}") for i, s in enumerate(sources)]


server = yd.ShardedServer(Stub, {"dbnet": sd}, pin_cores=False)
out = server.run(["a", "b", "c"], wave=2, gather="objects")
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RESULT " + json.dumps({"same": bool(same), "passthrough": bool(passthrough), "report": report, "scalars": scalars, "bytes": moved,
                              "server_replicas": server.replicas, "server_device": str(server.device), "out": out}))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_checkpoint_broadcast_and_replica_report_over_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, "-c", SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = next(ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT "))
    r = json.loads(line[len("RESULT "):])
    assert r["same"] is True and r["passthrough"] is True
    assert r["bytes"] > 90e6  # a DBNet checkpoint, as one fp32 message and two small ones
    assert r["report"]["ranks"] == 1 and r["report"]["backend"] == "nccl" and r["report"]["weights_crc_equal"] is True
    assert r["scalars"] == [[1.5, 4294967295.0, -0.0]]
    assert r["server_replicas"]["backend"] == "nccl" and r["server_replicas"]["weights_crc"] == r["report"]["weights_crc"]
    assert r["server_device"] == "cuda:0" and r["out"] == ["page:a", "page:b", "page:c"]

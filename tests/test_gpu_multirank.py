"""The N > 1 code path of bench.py on the ONE GPU the box has: two ranks under torch.distributed.run,
both on device 0, the exchange over gloo with CPU tensors (MD_BENCH_SHARE_DEVICE=1).  What runs is the real multi-rank
code - weak-scaling seeds per rank, the C4 batch cut by bytes (shard.shard_by_bytes), the gather of per-stream results
(shard.gather_varlen) and of the compressed members' bytes (shard.gather_payload), the max-over-ranks timing - with only
the backend, the device index and the device of the exchanged tensors different from a launch on N devices over RCCL.
Needs an MI355X: `pytest -m gpu`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device():
    env = dict(os.environ, MD_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    n, members = 192, 600
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", str(n),
           "--stream-kib", "64", "--deflate-streams", "64", "--deflate-kib", "64", "--deflate-steps", "1", "--gzip-members", str(members),
           "--no-cpu-baseline", "--no-host-path", "--no-text-leg"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["parity_ok"] is True
    assert d["config"]["results_gathered"] == 2 * n  # every rank's per-stream results reached rank 0
    assert d["deflate"]["parity_ok"] is True
    g = d["gzip"]
    assert g["scaling"] == "strong" and g["parity_ok"] is True
    assert g["members_total"] == members and g["results_gathered"] == members
    assert len(g["members_per_rank"]) == 2 and sum(g["members_per_rank"]) == members and min(g["members_per_rank"]) > 0
    assert g["gathered_bytes"] > 0

"""bench.py's multi-rank flow, rehearsed on the one GPU of the test box: `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2` with VMX_BENCH_LOCAL_RANKS=1 (both ranks on GPU 0, joined over gloo + peer-mapped memory instead of RCCL) and
a two-layer model: the line must be the tensor-parallel job (strong scaling, ONE model) with the N-replica job as a secondary,
the collectives' failure counters at zero, and a per-step time that the driver's clock can check."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(extra):
    env = dict(os.environ, VMX_BENCH_LOCAL_RANKS="1", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--prompt", "16", "--layers", "2", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]   # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_default_is_the_tensor_parallel_job():
    line = _run([])
    assert line["n_gpus"] == 2 and line["steps"] == 8 and line["warmup"] == 2
    assert line["scaling"] == "strong" and line["config"]["parallelism"].startswith("tp2")
    assert line["comm"]["state"] == {"p2p_timeouts": 0, "error_code": 0}
    assert line["value"] > 0 and abs(line["value"] - 8 / (line["ms_per_step"] * 8 / 1e3)) / line["value"] < 1e-3
    rep = line["replicas"]
    assert rep["scaling"] == "weak" and rep["value"] > 0
    assert line["roofline"] is not None and "model" not in line["config"] and "layers" not in line["config"]


def test_two_ranks_replicas_flag():
    line = _run(["--replicas"])
    assert line["scaling"] == "weak" and line["config"]["parallelism"].startswith("dp2")
    assert line["replicas"] is None and line["comm"]["state"] is None

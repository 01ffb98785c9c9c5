"""GPU: bench.py launched the way the driver launches N > 1 (torch.distributed.run, one process per rank), with
two ranks sharing cuda:0 over gloo - rank / sharding / barrier / MAX-reduce plumbing and the sharded fused update."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--particles", "512", "--backend", "gloo", "--device", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints, and only once
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["solver_failures"] == 0
    assert j["config"]["particles_per_gpu"] == 512
    assert abs(j["value"] - 2 * 512 * 32 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6     # whole-job aggregate

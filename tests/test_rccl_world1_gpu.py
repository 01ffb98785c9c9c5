"""GPU: the sharded control iteration as a rank of a multi-GPU run replays it - rollout + record launches, an RCCL
all-gather INSIDE the captured hipGraph, the combine kernel, the env step - on one GPU (world-size-1 RCCL group, a
communicator that claims two ranks): tools/rccl_world1.py, run in a process of its own."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_iteration_with_rccl_in_the_graph():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    s = socket.socket()                 # a free port (a fixed one can still be held by an earlier test's rendezvous)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py")], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert out.stdout.count("with an RCCL all-gather") == 2 and "\nok\n" in out.stdout
    assert out.stdout.count("collectives: library RCCL") == 2          # (the path bench.py would report; no fallback)
    assert out.stdout.count("captured iteration with its RCCL exchanges") == 3        # CEM, DMD-MPC (update_cov), random shooting

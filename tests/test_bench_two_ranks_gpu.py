"""GPU: bench.py at N > 1 - launched the way the driver launches it (torch.distributed.run, one process per rank) and
the way the driver launches N = 1 (plain ``python bench.py --gpus N``: bench.py then starts its own ranks as a fresh child
process tree), with two ranks sharing cuda:0 over gloo - rank / sharding / barrier / MAX-reduce plumbing, the sharded
updates of every controller (MPPI / DMD: one record all-gather, CEM: two) and both scaling modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints, and only once
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--particles", "512", "--backend", "gloo", "--device", "0", "--no-cpu-baseline"]
    j = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT))
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["solver_failures"] == 0
    assert j["config"]["particles_per_gpu"] == 512
    assert abs(j["value"] - 2 * 512 * 32 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6     # whole-job aggregate
    _check_strong_block(j)
    _check_collectives_fields(j)


def _check_strong_block(j, H=32):
    """One --gpus N invocation answers both readings of the metric: `value` is weak scaling (--particles per GPU), and
    the `strong` block is the same loop over --particles IN TOTAL (examples/example_mpc.py:78-79: num_particles is a total)."""
    st = j["strong"]
    assert st["scaling"] == "strong" and st["particles_total"] == 512 and st["particles_per_gpu"] == 256
    assert abs(st["value"] - 512 * H * 6 / (st["ms_per_step"] * 6e-3)) / st["value"] < 1e-6
    assert abs(st["control_loop_hz"] - 1e3 / st["ms_per_step"]) / st["control_loop_hz"] < 1e-6
    assert j["scaling"] == "weak"                       # the headline keeps its label


def _check_collectives_fields(j):
    """the line says which exchange path ran, what the library's communicator cost to make and whether any
    rank fell back - on gloo that is torch.distributed, no communicator, no fallback, and no A/B block (there is no second path)."""
    c = j["config"]
    assert c["collectives"] == "torch.distributed" and c["comm_init_s"] == 0 and c["collectives_fallback"] is False
    assert "collectives_ab" not in j
    if "strong" in j:
        assert j["strong"]["collectives"] == "torch.distributed"


def _self_launched(*flags):
    """``python bench.py --gpus 2 ...`` with no launcher and no WORLD_SIZE in the environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
           "--particles", "512", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + list(flags)
    return _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env))


def test_self_launch_mppi():
    j = _self_launched()
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["config"]["backend"] == "gloo"
    assert j["scaling"] == "weak" and j["config"]["particles_total"] == 1024 and j["config"]["collectives_per_step"] == 1
    assert abs(j["value"] - 2 * 512 * 32 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6
    assert j["solver_failures"] == 0
    _check_strong_block(j)
    _check_collectives_fields(j)
    assert "arm_rollout.hip" in j["config"]["build"]       # which scheduler alternative the kernels were compiled with


def test_self_launch_cem_strong_scaling():
    """BASELINE config 4's shape: CEM full covariance, the population divided over the ranks."""
    j = _self_launched("--controller", "cem", "--scaling", "strong")
    assert j["config"]["ranks_seen"] == 2 and j["scaling"] == "strong" and j["config"]["collectives_per_step"] == 2
    assert j["config"]["particles_per_gpu"] == 256 and j["config"]["particles_total"] == 512
    assert abs(j["value"] - 512 * 32 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6
    assert j["solver_failures"] == 0 and j["final_distance_to_target"] < 1.0
    assert "strong" not in j                            # (the headline already is the strong reading)


def test_self_launch_dmd_on_the_hand_tree():
    """BASELINE config 5's shape: DMD-MPC on the 24-dof tree, sharded."""
    j = _self_launched("--workload", "hand24", "--controller", "dmd", "--horizon", "8")
    assert j["config"]["ranks_seen"] == 2 and j["roofline"]["kernel"].startswith("tree_rollout_kernel")
    assert abs(j["value"] - 2 * 512 * 8 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6
    assert j["solver_failures"] == 0
    _check_strong_block(j, H=8)


def test_sharded_equals_unsharded_action_sequence():
    """Strong scaling must not change the result: the same 512 particles on one rank and on two give the same closed
    loop (device noise is keyed by the global particle index; the records combine in rank order)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "512", "--steps", "10", "--warmup", "2",
            "--no-cpu-baseline", "--scaling", "strong"]
    one = _line(subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env))
    two = _line(subprocess.run(base + ["--gpus", "2", "--backend", "gloo", "--device", "0"], capture_output=True,
                               text=True, timeout=900, cwd=ROOT, env=env))
    assert abs(one["final_distance_to_target"] - two["final_distance_to_target"]) < 1e-9

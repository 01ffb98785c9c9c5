"""GPU: the example driver (examples/example_mpc.py - the counterpart of mjmpc's) runs every controller block of the
shipped configuration for a short episode, with and without dynamics randomization."""
import os
import subprocess
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _short_config(tmp_path, env_name="reacher_7dof-v0"):
    with open(os.path.join(ROOT, "examples", "configs", "reacher_gpu.yml")) as f:
        exp = yaml.safe_load(f)
    exp["n_episodes"], exp["max_ep_length"], exp["env_name"] = 1, 6, env_name
    for block in exp.values():
        if isinstance(block, dict) and "particles_per_cpu" in block:
            block["num_cpu"], block["particles_per_cpu"] = 4, 16
    p = tmp_path / "exp.yml"
    p.write_text(yaml.safe_dump(exp))
    return str(p)


@pytest.mark.parametrize("controller,extra", [
    ("mppi", []), ("cem", ["--noise_mode", "device", "--graph"]), ("dmd", ["--noise_mode", "device"]),
    ("random_shooting", []), ("pfmpc", []),
    ("mppi", ["--dyn_randomize_config", os.path.join(ROOT, "examples", "configs", "reacher_gpu_dyn_randomize.yml")]),
])
def test_example_driver(tmp_path, controller, extra):
    cfg = _short_config(tmp_path)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "example_mpc.py"), "--config", cfg,
                          "--controller", controller] + extra, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Success Metric" in out.stdout and "solver failures 0" in out.stdout


def test_continual_config_parses(tmp_path):
    cfg = _short_config(tmp_path, "continual_reacher-v0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "example_mpc.py"), "--config", cfg,
                          "--controller", "mppi", "--noise_mode", "device"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]


def test_tree_config_runs(tmp_path):
    """examples/configs/hand_tree_gpu.yml: the same driver over the tree engine (24-dof hand), a short DMD-MPC episode."""
    with open(os.path.join(ROOT, "examples", "configs", "hand_tree_gpu.yml")) as f:
        exp = yaml.safe_load(f)
    exp["n_episodes"], exp["max_ep_length"] = 1, 5
    for block in exp.values():
        if isinstance(block, dict) and "particles_per_cpu" in block:
            block["particles_per_cpu"] = 128
    p = tmp_path / "tree.yml"
    p.write_text(yaml.safe_dump(exp))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "example_mpc.py"), "--config", str(p),
                          "--controller", "dmd", "--noise_mode", "device"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Success Metric" in out.stdout and "solver failures 0" in out.stdout

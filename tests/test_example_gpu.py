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


def test_integration_md_ctypes_stub_runs(raw_arm, ref_arm):
    """The ctypes stub INTEGRATION.md section 3 shows a maintainer (HipVecEnv) is executed as written and compared with
    the oracle: the documented binding is a working binding."""
    import numpy as np
    from mjmpc_amd.models.compile import compile_arm
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):text.index("## 4.")]
    code = sec[sec.index("```python") + len("```python"):]
    code = code[:code.index("```")]
    code = code.replace('ctypes.CDLL("libmjmpc_amd.so")', 'ctypes.CDLL(%r)' % os.path.join(ROOT, "mjmpc_amd", "libmjmpc_amd.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md#3", "exec"), ns)
    env = ns["HipVecEnv"](np.ascontiguousarray(compile_arm(raw_arm).blob))
    st = dict(qp=np.array([0.3, 0.5, -0.2, -1.0, 0.4, -0.6, 0.2]), qv=np.zeros(7), target_pos=np.array([0.1, 0.1, 0.1]))
    env.set_env_state(st)
    rs = np.random.RandomState(0)
    mean, noise = 0.1 * rs.randn(8, 7), rs.randn(16, 8, 7)
    obs, rew, act, done, info, nobs = env.rollout(16, 8, mean, noise)
    o = ref_arm.rollout(st["qp"], st["qv"], st["target_pos"], mean, noise)
    np.testing.assert_allclose(rew, o[1], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nobs, o[4], rtol=0, atol=1e-9)

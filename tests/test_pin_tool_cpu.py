"""CPU: tools/pin_with_mujoco.py stays one command away from a verdict.  No MuJoCo exists in this image, so
the tool's `--self-test` mode lets the oracle stand in for it (through the exported MJCF text): the export, the state
generators, the reset case and the per-feature table all run; nothing is pinned by this."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_tool_runs_end_to_end_in_self_test_mode():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_with_mujoco.py"), "--self-test", "reacher", "cartpole",
                          "fourbar", "gripper"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SELF-TEST ONLY" in out.stdout and "NOT PINNED" not in out.stdout
    for feature in ("joint limits", "friction loss", "equality: connect", "fixed tendon + tendon limits",
                    "reset on instability", "elliptic cones + impratio"):
        assert feature in out.stdout, feature
    assert "expected deviations" in out.stdout and "gripper*" in out.stdout       # the own-scheme colliders are named, not judged


def test_pin_tool_without_a_mujoco_says_what_it_would_do():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_with_mujoco.py")], capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    if out.returncode == 2:         # (this image: no mujoco, no mujoco_py)
        assert "Models that WOULD be compared" in out.stdout and "half_cheetah" in out.stdout

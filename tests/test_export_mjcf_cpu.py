"""CPU: RawModel -> MJCF -> RawModel round trip (mjmpc_amd/models/export_mjcf.py through mjmpc_amd/models/mjcf.py) on the
hand-built models and on the random models of tests/test_random_models_gpu.py: the loader reads back exactly the tables
the writer was given (``to_flat()`` equal), which exercises the loader's joints, geoms, pairs with overrides, equalities,
tendons, actuators and per-element solver parameters on a few hundred models."""
import importlib.util
import os

import numpy as np
import pytest

from mjmpc_amd.models.export_mjcf import to_mjcf
from mjmpc_amd.models.mjcf import load_mjcf

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("random_models", os.path.join(HERE, "test_random_models_gpu.py"))
_rm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_rm)


def _round_trip(raw, tmp_path, name):
    p = tmp_path / name
    p.write_text(to_mjcf(raw))
    back = load_mjcf(str(p), frame_skip=raw.frame_skip, task=raw.task, ctrl_cost=raw.ctrl_cost, obs_skip=raw.obs_skip, self_collision=False)
    back.site_axis, back.target_dir, back.capsule_cap_factor = raw.site_axis, raw.target_dir, raw.capsule_cap_factor
    return back


@pytest.mark.parametrize("seed", range(0, 120))
def test_random_model_round_trip(seed, tmp_path):
    raw = _rm.random_model(seed)
    back = _round_trip(raw, tmp_path, "m%d.xml" % seed)
    a, b = raw.to_flat(), back.to_flat()
    assert a.shape == b.shape
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-15)


def test_named_models_round_trip(tmp_path):
    """Every model this repository names - the three the reference vendors (restated tables), the synthetic hand, the
    pen-in-hand, the four MJCF assets - through the writer and back."""
    from mjmpc_amd.models.half_cheetah import half_cheetah_raw
    from mjmpc_amd.models.hand24 import hand24_raw
    from mjmpc_amd.models.pen_hand import pen_hand_raw
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    from mjmpc_amd.models.swimmer import swimmer_raw
    from mjmpc_amd.models.synthetic import synthetic_raw
    raws = [reacher7dof_raw(), half_cheetah_raw(), swimmer_raw(), hand24_raw(), pen_hand_raw()] + \
           [synthetic_raw(n) for n in ("cartpole", "tray", "door", "fourbar")]
    for k, raw in enumerate(raws):
        back = _round_trip(raw, tmp_path, "h%d.xml" % k)
        np.testing.assert_allclose(back.to_flat(), raw.to_flat(), rtol=0, atol=1e-15)

"""CPU: the C-ABI library loads and exports every symbol include/mjmpc_amd.h declares
(no compute calls without a GPU), and the product path refuses to run without one."""
import os
import re

import pytest

from mjmpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mjmpc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mjmpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mjmpc_amd.build import build
    build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table covers the header exactly
    assert lib.mjmpc_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MJMPC_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "mjmpc_amd.h")).read()).group(1))


def test_no_cpu_fallback():
    lib = _lib.load()
    if lib.mjmpc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MjmpcError):
        _lib.require_gpu()
    from mjmpc_amd.envs.arm_engine import ArmRolloutEngine
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    with pytest.raises(_lib.MjmpcError):
        ArmRolloutEngine(reacher7dof_raw())


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "mjmpc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "reacher_ref" not in src or f in ("raw.py",), f


def test_argument_errors_are_codes_and_messages_not_crashes():
    """Bad arguments are rejected before anything touches the GPU: non-zero return code + mjmpc_last_error()."""
    import ctypes
    lib = _lib.load()
    bad = [
        lambda: lib.mjmpc_arm_rollout(None, _lib.F64, 8, 4, None, None, None, None, None, None, None),
        lambda: lib.mjmpc_arm_set_state(None, None, None, None, None),
        lambda: lib.mjmpc_arm_solver_failures(None, None),
        lambda: lib.mjmpc_analytic_rollout(0, None, 2, 1, None, _lib.F64, 8, 4, None, None, None, None, None, None, 0, None),
        lambda: lib.mjmpc_softmax_stats(_lib.F64, 8, 4, 2, None, None, None, None, None, 0, 0.1, 1, 0, 0, None, None, None),
        lambda: lib.mjmpc_td_lambda_returns(_lib.F64, 8, 4, 2, None, None, None, None, None, None, 0, 0.1, 1, 1.0, 1.0,
                                            None, None, None),
        lambda: lib.mjmpc_shift_mean(None, 4, 2, 0, None, None),
        lambda: lib.mjmpc_cholesky_lower(None, 3, None, None, None),
        lambda: lib.mjmpc_cov_add_diag(None, 3, None, 1.0, None),
        lambda: lib.mjmpc_sample_noise(_lib.F64, None, 8, 4, 2, None, None, 1, 0, 0, None, 1, None),
        lambda: lib.mjmpc_sample_noise_mt19937(_lib.F64, None, 64, 1.0, 1, None, None, None, None),
        lambda: lib.mjmpc_sample_noise_mt19937_jump(_lib.F64, None, 64, 1.0, 1, None, None, None, 19968, 2000, 4, 0, None,
                                                    None, None),
    ]
    for call in bad:
        rc = call()
        assert rc != 0
        assert len(lib.mjmpc_last_error()) > 0
    h = ctypes.c_void_p()
    assert lib.mjmpc_arm_create(None, 0, 0, ctypes.byref(h)) != 0          # no model blob
    assert lib.mjmpc_arm_destroy(None) == 0                                 # destroying nothing is a no-op
    assert lib.mjmpc_mt19937_stream_words(1000) > 4 * 500 / 0.79

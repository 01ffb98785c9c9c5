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
    assert lib.mjmpc_abi_version() == 1


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

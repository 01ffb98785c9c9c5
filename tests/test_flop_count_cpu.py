"""CPU: the FLOP-counting build of the oracle (oracle/flop_count.cpp: reacher_ref.c compiled as C++ with a counting
scalar - SURVEY 8d's "instrumented cpu_ref build" behind bench.py's roofline.valu) computes exactly what the oracle
computes, and its tally is the figure DESIGN.md quotes."""
import numpy as np

from mjmpc_amd.models.reacher7dof import reacher7dof_raw
from oracle.physics_ref import RefArm, count_flops


def test_counting_build_is_the_oracle_and_counts_30k_flop_per_particle_step():
    raw = reacher7dof_raw()
    flat = raw.to_flat()
    rs = np.random.RandomState(123)
    P, H = 64, 32
    noise = rs.standard_normal((P, H, 7))
    for t in range(2, H):
        noise[:, t] = 0.25 * noise[:, t] + 0.8 * noise[:, t - 1]
    mean, tgt = np.zeros((H, 7)), np.array([0.1, 0.1, 0.1])
    d = count_flops(flat, np.zeros(7), np.zeros(7), tgt, mean, noise)
    _, rew, _, _, _ = RefArm(flat).rollout(np.zeros(7), np.zeros(7), tgt, mean, noise, want_obs=False)
    # same source, same arithmetic - up to the polishing Newton step the counting build leaves out (it would be counted
    # as work of the algorithm; it moves results by less than the oracle's 1e-11 stopping tolerance)
    np.testing.assert_allclose(d["rew"], rew, rtol=1e-10, atol=1e-10)
    assert d["flops"] == d["add"] + d["mul"] + d["div"] + d["sqrt"] + d["trig"]
    assert 2.5e4 < d["flops"] < 3.5e4                          # 30 284 on this sample (bench.py reports the exact count)
    assert d["trig"] > 14 and d["sqrt"] > 2                    # 7 joints x 2 substeps x (sin, cos); the cost's norm

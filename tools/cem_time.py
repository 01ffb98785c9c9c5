"""Developer tool: event timings of the fused CEM step's launches (mjmpc_cem_select_moments, mjmpc_cem_finish with and
without the next step's draw) and of the separate launches they replace, on synthetic data.

    python tools/cem_time.py [P] [H] [A] [elite_frac]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mjmpc_amd import _lib
from mjmpc_amd.control._device import DeviceUpdater, _vp

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
A = int(sys.argv[3]) if len(sys.argv) > 3 else 7
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
k = int(P * frac)
dev = DeviceUpdater(H, A, np.ones(H))
lib = dev.lib
g = torch.Generator(device="cuda").manual_seed(0)
actions = torch.randn(P, H, A, device="cuda", dtype=torch.float64, generator=g)
q0 = torch.rand(P, device="cuda", dtype=torch.float64, generator=g) * 3 + 40
dev.set_mean(np.zeros((H, A)))
dev.set_cov(0.5 * np.eye(A))
ws = dev.workspace(P)
dev._q0_view(ws, P).copy_(q0)
step = torch.zeros(1, dtype=torch.int64, device="cuda")
act = torch.zeros(A, dtype=torch.float64, device="cuda")
noise = torch.empty(P, H, A, device="cuda", dtype=torch.float64)
costs = torch.zeros(P, H, device="cuda", dtype=torch.float64)
chol = dev.record("chol", A * A)
code = dev.code(actions)
s = dev.stream()


def select():
    _lib.check(lib.mjmpc_cem_select_moments(code, P, H, A, _vp(actions), None, P, 0, k, _vp(dev.mean), _vp(dev.cov), _vp(step), _vp(ws), s))


def finish(draw):
    _lib.check(lib.mjmpc_cem_finish(code, P, H, A, k, None, 1, float(k), 1, 0.8, 0, _vp(dev.mean), _vp(dev.cov), _vp(chol),
                                    _vp(dev.chol_status), None, 0.02, _vp(act), None, _vp(step), _vp(noise) if draw else None,
                                    7, 0, 0, _vp(ws), s))


def old_update():
    dev.cem_update(costs, actions, k, 0.8, True, q0=dev._q0_view(ws, P))


def old_noise():
    dev.sample_noise(P, None, [0.25, 0.8, 0.0], 7, 0, filtered=False)


def timeit(fn, n=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("P %d H %d A %d k %d" % (P, H, A, k))
print("select + moments        %7.1f us" % timeit(select))
print("finish, no draw         %7.1f us" % timeit(lambda: finish(False)))
print("finish + next samples   %7.1f us" % timeit(lambda: finish(True)))
print("separate update launches %6.1f us" % timeit(old_update))
print("separate cholesky + sampler %4.1f us" % timeit(old_noise))

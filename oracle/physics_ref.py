"""ctypes binding of oracle/libreacher_ref.so  --  TEST ORACLE, NOT PRODUCT CODE.

PARITY UNPINNED for the physics (see the header of reacher_ref.c).  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libreacher_ref.so")
    src = os.path.join(_HERE, "reacher_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libreacher_ref.so"], stdout=subprocess.DEVNULL)
    return so


def build_flops(force=False):
    """The FLOP-counting build of the same source (oracle/flop_count.cpp)."""
    so = os.path.join(_HERE, "libreacher_flops.so")
    srcs = [os.path.join(_HERE, f) for f in ("reacher_ref.c", "flop_count.cpp")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libreacher_flops.so"], stdout=subprocess.DEVNULL)
    return so


NATIVE_FLAGS = "-O3 -march=native -ffp-contract=fast -fPIC -fopenmp -std=gnu11"


def build_flags():
    """The flags of the library bench.py's cpu_baseline leg times (see build_native)."""
    return "gcc " + NATIVE_FLAGS


def build_native():
    """The same source built for THIS host (``-O3 -march=native``, FMA contraction allowed): what bench.py's
    ``cpu_baseline`` leg times, so that the CPU figure is not handicapped by the checker's portable ``-O2`` build.  The
    library is rebuilt whenever the host CPU differs from the one it was built on (it travels with gpurun snapshots
    from the build container to the GPU box).  Falls back to the portable build if this host cannot compile."""
    so = os.path.join(_HERE, "libreacher_ref_native.so")
    stamp = os.path.join(_HERE, "libreacher_ref_native.stamp")
    src = os.path.join(_HERE, "reacher_ref.c")
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = "".join(ln for ln in f if ln.startswith(("model name", "flags")))[:20000]
    except OSError:
        pass
    import hashlib
    want = hashlib.sha1((cpu + NATIVE_FLAGS).encode()).hexdigest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not os.path.exists(so) or have != want or os.path.getmtime(so) < os.path.getmtime(src):
        try:
            subprocess.check_call(["gcc"] + NATIVE_FLAGS.split() + ["-shared", "-o", so, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(stamp, "w") as f:
                f.write(want)
        except (OSError, subprocess.CalledProcessError):
            return build()
    return so


def count_flops(flat, qp0, qv0, target, mean, noise):
    """Floating-point operations the oracle executes per particle-step of ``rollout``: a dict with the
    tallies per kind and ``flops`` = add + mul + div + sqrt + trig (one each; compares are listed, not counted)."""
    L = ctypes.CDLL(build_flops())
    L.or_model_compile.restype = ctypes.c_void_p
    L.or_model_compile.argtypes = [_dp, ctypes.c_int]
    L.or_model_free.argtypes = [ctypes.c_void_p]
    L.or_rollout.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, ctypes.c_long, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.or_flops_get.argtypes = [ctypes.POINTER(ctypes.c_long)]
    flat, mean, noise = _c(flat), _c(mean), _c(noise)
    h = L.or_model_compile(_p(flat), flat.size)
    P, H, nu = noise.shape
    rew, act, done = np.zeros((P, H)), np.zeros((P, H, nu)), np.zeros((P, H))
    L.or_flops_reset()
    L.or_rollout(h, _p(_c(qp0)), _p(_c(qv0)), _p(_c(target)), P, H, _p(mean), _p(noise), None, _p(rew), _p(act), _p(done), None)
    out = (ctypes.c_long * 6)()
    L.or_flops_get(out)
    L.or_model_free(h)
    n = float(P * H)
    d = dict(zip(("add", "mul", "div", "sqrt", "trig", "cmp"), (v / n for v in out)))
    d["flops"] = d["add"] + d["mul"] + d["div"] + d["sqrt"] + d["trig"]
    d["rew"] = rew
    return d


_NATIVE = None


def _lib(native=False):
    global _LIB, _NATIVE
    if native:
        if _NATIVE is None:
            _NATIVE = _bind(ctypes.CDLL(build_native()))
        return _NATIVE
    if _LIB is None:
        _LIB = _bind(ctypes.CDLL(build()))
    return _LIB


def _bind(L):
    L.or_model_compile.restype = ctypes.c_void_p
    L.or_model_compile.argtypes = [_dp, ctypes.c_int]
    L.or_model_free.argtypes = [ctypes.c_void_p]
    for name in ("or_nv", "or_nq", "or_nbody", "or_dobs"):
        getattr(L, name).restype = ctypes.c_int
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.or_kinetic.restype = ctypes.c_double
    L.or_kinetic.argtypes = [ctypes.c_void_p, _dp, _dp]
    L.or_env_step.restype = ctypes.c_double
    L.or_env_step.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _dp, _dp]
    L.or_step.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _dp, _dp]
    L.or_step_mj.restype = ctypes.c_int
    L.or_step_mj.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _dp, _dp]
    L.or_get_resets.restype = ctypes.c_long
    L.or_get_resets.argtypes = [ctypes.c_void_p]
    L.or_rollout.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, ctypes.c_long, ctypes.c_int,
                             _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.or_rollout_cl.argtypes = L.or_rollout.argtypes
    L.or_threads.restype = ctypes.c_int
    L.or_threads.argtypes = [ctypes.c_int]
    return L


def threads(n=0, native=False):
    """OpenMP threads used by RefArm.rollout (n > 0 sets the count first)."""
    return _lib(native).or_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RefArm:
    """FP64 reference arm: compiled from ``RawModel.to_flat()`` by the C oracle itself."""

    def __init__(self, flat, native=False):
        """``native``: the -O3 -march=native build (bench.py's cpu_baseline leg); the checker uses the portable one."""
        flat = _c(flat)
        self._L = _lib(native)
        self._h = self._L.or_model_compile(_p(flat), flat.size)
        if not self._h:
            raise ValueError("oracle: bad model blob")
        self.nv = self._L.or_nv(self._h)
        self.nq = self._L.or_nq(self._h)
        self.nbody = self._L.or_nbody(self._h)
        self.qpos0 = np.zeros(self.nq)
        self._L.or_qpos0(ctypes.c_void_p(self._h), _p(self.qpos0))
        self.d_obs = self._L.or_dobs(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.or_model_free(self._h)
            self._h = None

    # -- run-time edits (dynamics randomization) ----------------------------
    def set_body_mass(self, body, mass):
        self._L.or_set_body_mass(ctypes.c_void_p(self._h), int(body), ctypes.c_double(mass))

    def set_body_inertia(self, body, tensor):
        self._L.or_set_body_inertia(ctypes.c_void_p(self._h), int(body), _p(_c(tensor).reshape(-1)))

    def set_dof_damping(self, dof, d):
        self._L.or_set_dof_damping(ctypes.c_void_p(self._h), int(dof), ctypes.c_double(d))

    def set_dof_frictionloss(self, dof, f):
        self._L.or_set_dof_frictionloss(ctypes.c_void_p(self._h), int(dof), ctypes.c_double(f))

    def set_sphere_radius(self, idx, r):
        self._L.or_set_sphere_radius(ctypes.c_void_p(self._h), int(idx), ctypes.c_double(r))

    def set_sphere_pos(self, idx, xyz):
        self._L.or_set_sphere_pos(ctypes.c_void_p(self._h), int(idx), _p(_c(xyz)))

    def set_sphere_mu(self, idx, mu):
        self._L.or_set_sphere_mu(ctypes.c_void_p(self._h), int(idx), ctypes.c_double(mu))

    # -- compiled constants -------------------------------------------------
    def inertial(self):
        mass = np.zeros(self.nbody)
        ipos = np.zeros((self.nbody, 3))
        inertia = np.zeros((self.nbody, 3, 3))
        self._L.or_get_inertial(ctypes.c_void_p(self._h), _p(mass), _p(ipos), _p(inertia))
        return mass, ipos, inertia

    def invweight0(self):
        dof = np.zeros(self.nv)
        body = np.zeros(self.nbody)
        self._L.or_get_invweight0(ctypes.c_void_p(self._h), _p(dof), _p(body))
        return dof, body

    def invweight0_rot(self):
        body = np.zeros(self.nbody)
        self._L.or_get_invweight0_rot(ctypes.c_void_p(self._h), _p(body))
        return body

    def newton_stats(self):
        out = (ctypes.c_long * 3)()
        self._L.or_get_newton_stats(ctypes.c_void_p(self._h), out)
        return dict(calls=out[0], iters=out[1], fails=out[2])

    def resets(self):
        """mj_resetData calls so far (mj_checkPos / mj_checkVel / mj_checkAcc found a NaN or an entry beyond 1e10)."""
        return int(self._L.or_get_resets(ctypes.c_void_p(self._h)))

    def step_mj(self, q, v, ctrl):
        """One mj_step with data.ctrl in-out: returns (q', v', site, ctrl after the step - zeroed by a reset -, resets)."""
        q, v, u = _c(q).copy(), _c(v).copy(), _c(ctrl).copy()
        site = np.zeros(6)
        n = self._L.or_step_mj(self._h, _p(q), _p(v), _p(u), _p(site), None)
        return q, v, site[:3].copy(), u, n

    def solve_rows(self, M, fs, J, aref, D, kind, floss):
        """The constraint solver alone (test hook): minimise 1/2 (a - M^-1 fs)' M (a - M^-1 fs) + sum_i s_i(J_i a - aref_i)
        with row kinds 0 unilateral, 1 equality, 2 friction loss; returns (a, row forces)."""
        M, fs, J = _c(M), _c(fs), _c(J)
        nv, nc = fs.size, J.shape[0]
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        a, force = np.zeros(nv), np.zeros(max(nc, 1))
        self._L.or_solve_rows(ctypes.c_void_p(self._h), nv, _p(M), _p(fs), nc, _p(J), _p(_c(aref)), _p(_c(D)),
                              kind.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(_c(floss)), _p(a), _p(force))
        return a, force[:nc]

    # -- building blocks ----------------------------------------------------
    def mass_matrix(self, q):
        M = np.zeros((self.nv, self.nv))
        self._L.or_mass_matrix(ctypes.c_void_p(self._h), _p(_c(q)), _p(M))
        return M

    def rne(self, q, v, a=None):
        tau = np.zeros(self.nv)
        self._L.or_rne(ctypes.c_void_p(self._h), _p(_c(q)), _p(_c(v)),
                       _p(None if a is None else _c(a)), _p(tau))
        return tau

    def site(self, q):
        x = np.zeros(3)
        self._L.or_site(ctypes.c_void_p(self._h), _p(_c(q)), _p(x))
        return x

    def kinetic(self, q, v):
        return self._L.or_kinetic(self._h, _p(_c(q)), _p(_c(v)))

    def step(self, q, v, ctrl):
        """One mj_step.  Returns (q', v', site position seen by that step, diag)."""
        q, v = _c(q).copy(), _c(v).copy()
        site = np.zeros(6)              # the site, then the object's axis (task 2) in the world
        diag = np.zeros(1 + self.nv)
        self._L.or_step(self._h, _p(q), _p(v), _p(_c(ctrl)), _p(site), _p(diag))
        self.last_axis = site[3:].copy()
        return q, v, site[:3].copy(), diag

    def env_step(self, q, v, u, target):
        q, v = _c(q).copy(), _c(v).copy()
        obs = np.zeros(self.d_obs)
        r = self._L.or_env_step(self._h, _p(q), _p(v), _p(_c(u)), _p(_c(target)), _p(obs))
        return q, v, r, obs

    def rollout(self, qp0, qv0, target, mean, noise, want_obs=True, mode="open_loop", horizon=None):
        """GymEnvWrapper.rollout -> (obs, rew, act, done, next_obs).  mode 'closed_loop_linear': ``mean``
        is the (d_obs+1, nu) weight matrix and ``horizon`` must be given when ``noise`` is None."""
        mean = _c(mean)
        closed = mode == "closed_loop_linear"
        H, nu = (horizon, mean.shape[1]) if closed else mean.shape
        if closed and noise is not None:
            H = noise.shape[1]
        if noise is not None:
            noise = _c(noise)
            P = noise.shape[0]
            assert noise.shape == (P, H, nu)
        else:
            P = 1
        rew = np.zeros((P, H))
        act = np.zeros((P, H, nu))
        done = np.zeros((P, H))
        obs = np.zeros((P, H, self.d_obs)) if want_obs else None
        nobs = np.zeros((P, H, self.d_obs)) if want_obs else None
        fn = self._L.or_rollout_cl if closed else self._L.or_rollout
        fn(self._h, _p(_c(qp0)), _p(_c(qv0)), _p(_c(target)), P, H,
           _p(mean), _p(noise), _p(obs), _p(rew), _p(act), _p(done), _p(nobs))
        return obs, rew, act, done, nobs

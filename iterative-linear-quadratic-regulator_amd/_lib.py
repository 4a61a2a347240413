"""ctypes binding of libilqr_hip.so (C-ABI declared in include/ilqr_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (or
``make -C iterative-linear-quadratic-regulator_amd/csrc``).  There is no CPU
fallback: if the library is missing, or no gfx950 device is visible, every
entry point raises.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libilqr_hip.so"
LIB_PATH = os.environ.get("ILQR_LIB") or os.path.join(HERE, LIB_NAME)   # ILQR_LIB: A/B builds (tools/)
CSRC = os.path.join(HERE, "csrc")

# ---- enums (include/ilqr_hip.h) --------------------------------------------------
OK, ERR_INVALID_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_STATE = range(6)
F32, F64 = 0, 1
SYS_PENDULUM, SYS_UA_DOUBLE_PENDULUM, SYS_DOUBLE_PENDULUM, SYS_LINEAR, SYS_CUSTOM = range(5)
INTEGRATORS = {"euler": 0, "midpoint": 1, "rk4": 2, "backward_euler": 3, "discrete": 4}
(X, U, K, UFF, X0, COST, STATUS, ITERS, ALPHA, TRIAL_COSTS, LIN, PLANT_X, PROBE) = range(13)
TRAJ_ACTIVE, TRAJ_CONVERGED, TRAJ_LINESEARCH_FAILED, TRAJ_MAXITER = range(4)
TRAJ_FLAG_NON_PD = 0x100
FLAG_KEEP_ITERATING = 1
FLAG_NO_FUSE = 2
FLAG_NO_PERSIST = 4
PHASES = ("linearize", "backward", "forward", "select", "other", "fused", "persist")
ABI_VERSION = 3

# every symbol include/ilqr_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "ilqr_abi_version", "ilqr_device_count", "ilqr_param_count", "ilqr_is_supported", "ilqr_last_error",
    "ilqr_create", "ilqr_create_custom", "ilqr_destroy", "ilqr_sync", "ilqr_set_problem", "ilqr_set", "ilqr_get",
    "ilqr_initial_rollout", "ilqr_linearize", "ilqr_backward", "ilqr_forward", "ilqr_select", "ilqr_iterate",
    "ilqr_flush", "ilqr_solve", "ilqr_backward_pass", "ilqr_backward_tensors", "ilqr_forward_pass", "ilqr_eval_points", "ilqr_mpc_reset",
    "ilqr_mpc_rearm", "ilqr_mpc_run", "ilqr_status_reduce", "ilqr_timing_enable", "ilqr_timing_reset", "ilqr_timing_get", "ilqr_algorithmic_bytes",
)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_x", C.c_int32), ("n_u", C.c_int32), ("horizon", C.c_int32), ("batch", C.c_int32),
        ("n_alpha", C.c_int32), ("n_trials", C.c_int32), ("dtype", C.c_int32), ("system", C.c_int32),
        ("integrator", C.c_int32), ("plant_integrator", C.c_int32), ("device", C.c_int32),
        ("maxiter", C.c_int32), ("flags", C.c_int32),
        ("dt", C.c_double), ("tol", C.c_double), ("alpha_factor", C.c_double), ("min_alpha", C.c_double),
        ("mu", C.c_double),
        ("params", C.POINTER(C.c_double)), ("n_params", C.c_int32), ("reserved", C.c_int32),
        ("stream", C.c_void_p),
    ]


class IlqrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libilqr_hip error {code}: {msg}")
        self.code = code


_lib = None


def build(verbose=False):
    """Compile libilqr_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libilqr_hip.so failed")
    return LIB_PATH


def load():
    """Load the in-tree shared library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP hot path has not been built and there is no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "iterative-linear-quadratic-regulator_amd/csrc`).")
    lib = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.ilqr_abi_version.restype = ci
    lib.ilqr_device_count.argtypes = [C.POINTER(ci)]
    lib.ilqr_param_count.argtypes = [ci, ci, ci]
    lib.ilqr_is_supported.argtypes = [ci, ci, ci, ci]
    lib.ilqr_last_error.argtypes = [vp]
    lib.ilqr_last_error.restype = C.c_char_p
    lib.ilqr_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    lib.ilqr_create_custom.argtypes = [C.POINTER(vp), C.POINTER(Config), C.c_char_p]
    lib.ilqr_destroy.argtypes = [vp]
    lib.ilqr_sync.argtypes = [vp]
    lib.ilqr_set_problem.argtypes = [vp, vp, vp]
    lib.ilqr_set.argtypes = [vp, ci, vp, C.c_size_t]
    lib.ilqr_get.argtypes = [vp, ci, vp, C.c_size_t]
    for name in ("ilqr_initial_rollout", "ilqr_linearize", "ilqr_backward", "ilqr_select", "ilqr_timing_reset", "ilqr_flush"):
        getattr(lib, name).argtypes = [vp]
    lib.ilqr_forward.argtypes = [vp, C.POINTER(cd), ci]
    lib.ilqr_iterate.argtypes = [vp, ci]
    lib.ilqr_solve.argtypes = [vp, vp, vp]
    lib.ilqr_backward_pass.argtypes = [vp, vp, vp, vp, vp]
    lib.ilqr_backward_tensors.argtypes = [vp, vp, vp, vp, vp]
    lib.ilqr_forward_pass.argtypes = [vp, vp, cd, vp, vp, vp, vp, vp, vp, vp]
    lib.ilqr_eval_points.argtypes = [vp, ci, ci] + [vp] * 14
    lib.ilqr_mpc_reset.argtypes = [vp, vp, vp]
    lib.ilqr_mpc_rearm.argtypes = [vp, vp, vp]
    lib.ilqr_mpc_run.argtypes = [vp, ci, vp, vp, vp]
    lib.ilqr_status_reduce.argtypes = [vp, vp]
    lib.ilqr_timing_enable.argtypes = [vp, ci]
    lib.ilqr_timing_get.argtypes = [vp, C.POINTER(cd), C.POINTER(C.c_int64)]
    lib.ilqr_algorithmic_bytes.argtypes = [vp, C.POINTER(cd)]
    if lib.ilqr_abi_version() != ABI_VERSION:
        raise RuntimeError("libilqr_hip.so ABI version mismatch: rebuild the library")
    _lib = lib
    return lib


def device_count():
    n = C.c_int(0)
    load().ilqr_device_count(C.byref(n))
    return n.value


def np_dtype(dtype):
    dt = np.dtype(dtype)
    if dt == np.float32:
        return dt, F32
    if dt == np.float64:
        return dt, F64
    raise ValueError(f"dtype must be float32 or float64, got {dt}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Handle:
    """Thin owner of one ``ilqr_handle`` (one device, one stream, one batch of trajectories)."""

    def __init__(self, *, system, n_x, n_u, horizon, batch, params, dt, integrator, dtype=np.float64,
                 n_alpha=10, n_trials=10, tol=1e-5, maxiter=100, alpha_factor=0.5, min_alpha=1e-8, mu=0.0,
                 plant_integrator=None, device=0, flags=0, stream=None, plugin=None):
        self.lib = load()
        self.np_dtype, dcode = np_dtype(dtype)
        if isinstance(integrator, str):
            if integrator not in INTEGRATORS:
                raise ValueError(f"Unknown integrator: '{integrator}'. Supported: 'rk4', 'midpoint', 'euler', "
                                 "'backward_euler'.")
            integrator = INTEGRATORS[integrator]
        if isinstance(plant_integrator, str):
            plant_integrator = INTEGRATORS[plant_integrator]
        self.n_x, self.n_u, self.N, self.B, self.A = int(n_x), int(n_u), int(horizon), int(batch), int(n_alpha)
        self.E = 2 * n_x * n_x + 2 * n_x * n_u + n_x + n_u + n_u * n_u
        p = np.ascontiguousarray(params, dtype=np.float64)
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.n_x, cfg.n_u, cfg.horizon, cfg.batch = self.n_x, self.n_u, self.N, self.B
        cfg.n_alpha, cfg.n_trials, cfg.dtype, cfg.system = self.A, int(n_trials), dcode, int(system)
        cfg.integrator = int(integrator)
        cfg.plant_integrator = -1 if plant_integrator is None else int(plant_integrator)
        cfg.device, cfg.maxiter, cfg.flags = int(device), int(maxiter), int(flags)
        cfg.dt, cfg.tol, cfg.alpha_factor, cfg.min_alpha, cfg.mu = float(dt), float(tol), float(alpha_factor), \
            float(min_alpha), float(mu)
        cfg.params = p.ctypes.data_as(C.POINTER(C.c_double))
        cfg.n_params = p.size
        cfg.stream = stream
        h = C.c_void_p()
        if plugin is not None:  # user-defined system: kernels live in the plugin (systems/custom_sys.py)
            rc = self.lib.ilqr_create_custom(C.byref(h), C.byref(cfg), os.fsencode(plugin))
        else:
            rc = self.lib.ilqr_create(C.byref(h), C.byref(cfg))
        if rc != OK:
            msg = self.lib.ilqr_last_error(None).decode()
            if rc == ERR_INVALID_ARG:
                raise ValueError(msg)
            raise IlqrError(rc, msg)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.ilqr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            msg = self.lib.ilqr_last_error(self.h).decode()
            if rc == ERR_INVALID_ARG:
                raise ValueError(msg)
            raise IlqrError(rc, msg)

    def _in(self, a, shape):
        a = np.ascontiguousarray(a, dtype=self.np_dtype)
        if a.shape != tuple(shape):
            raise ValueError(f"expected array of shape {tuple(shape)}, got {a.shape}")
        return a

    # ---- shapes ---------------------------------------------------------------------
    def shape(self, field):
        B, n, m, N = self.B, self.n_x, self.n_u, self.N
        return {X: (B, n, N + 1), U: (B, m, N), K: (B, N, m, n), UFF: (B, m, N), X0: (B, n), COST: (B,),
                STATUS: (B,), ITERS: (B,), ALPHA: (B,), TRIAL_COSTS: (B, self.A), LIN: (B, N, self.E),
                PLANT_X: (B, n), PROBE: (8,)}[field]

    def get(self, field):
        dt = np.int32 if field in (STATUS, ITERS) else (np.int64 if field == PROBE else self.np_dtype)
        out = np.empty(self.shape(field), dtype=dt)
        self._chk(self.lib.ilqr_get(self.h, field, _ptr(out), out.nbytes))
        return out

    def set(self, field, value):
        a = self._in(value, self.shape(field))
        self._chk(self.lib.ilqr_set(self.h, field, _ptr(a), a.nbytes))

    def set_problem(self, x0, U_init):
        x0 = self._in(x0, (self.B, self.n_x))
        U_init = self._in(U_init, (self.B, self.n_u, self.N))
        self._chk(self.lib.ilqr_set_problem(self.h, _ptr(x0), _ptr(U_init)))

    # ---- stages -----------------------------------------------------------------------
    def sync(self):
        self._chk(self.lib.ilqr_sync(self.h))

    def initial_rollout(self):
        self._chk(self.lib.ilqr_initial_rollout(self.h))

    def linearize(self):
        self._chk(self.lib.ilqr_linearize(self.h))

    def backward(self):
        self._chk(self.lib.ilqr_backward(self.h))

    def forward(self, alphas):
        a = np.ascontiguousarray(alphas, dtype=np.float64)
        self._chk(self.lib.ilqr_forward(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))

    def select(self):
        self._chk(self.lib.ilqr_select(self.h))

    def iterate(self, n=1):
        self._chk(self.lib.ilqr_iterate(self.h, int(n)))

    def flush(self):
        """Enqueue the acceptance step ilqr_iterate may have left pending (every state access does this by itself)."""
        self._chk(self.lib.ilqr_flush(self.h))

    def solve(self):
        iters = np.empty(self.B, dtype=np.int32)
        cost = np.empty(self.B, dtype=self.np_dtype)
        self._chk(self.lib.ilqr_solve(self.h, _ptr(iters), _ptr(cost)))
        return iters, cost

    # ---- pure functional calls ----------------------------------------------------------
    def backward_pass(self, X_, U_):
        X_ = self._in(X_, self.shape(X))
        U_ = self._in(U_, self.shape(U))
        uff = np.empty(self.shape(UFF), dtype=self.np_dtype)
        k = np.empty(self.shape(K), dtype=self.np_dtype)
        self._chk(self.lib.ilqr_backward_pass(self.h, _ptr(X_), _ptr(U_), _ptr(uff), _ptr(k)))
        return uff, k

    def backward_tensors(self, lin, term):
        """Riccati sweep on a caller-supplied expansion: lin (B, N, E), term (B, n + n*n) -> (U_ff, K)."""
        lin = self._in(lin, (self.B, self.N, self.E))
        term = self._in(term, (self.B, self.n_x + self.n_x * self.n_x))
        uff = np.empty(self.shape(UFF), dtype=self.np_dtype)
        k = np.empty(self.shape(K), dtype=self.np_dtype)
        self._chk(self.lib.ilqr_backward_tensors(self.h, _ptr(lin), _ptr(term), _ptr(uff), _ptr(k)))
        return uff, k

    def forward_pass(self, x0, alpha, X_old, U_old, U_ff, K_):
        x0 = self._in(x0, self.shape(X0))
        X_old = self._in(X_old, self.shape(X))
        U_old = self._in(U_old, self.shape(U))
        U_ff = self._in(U_ff, self.shape(UFF))
        K_ = self._in(K_, self.shape(K))
        Xn = np.empty(self.shape(X), dtype=self.np_dtype)
        Un = np.empty(self.shape(U), dtype=self.np_dtype)
        cost = np.empty(self.B, dtype=self.np_dtype)
        self._chk(self.lib.ilqr_forward_pass(self.h, _ptr(x0), float(alpha), _ptr(X_old), _ptr(U_old),
                                             _ptr(U_ff), _ptr(K_), _ptr(Xn), _ptr(Un), _ptr(cost)))
        return Xn, Un, cost

    EVAL_NAMES = ("f", "f_x", "f_u", "l", "l_x", "l_u", "l_xx", "l_ux", "l_uu", "l_f", "l_f_x", "l_f_xx")

    def eval_points(self, x, u=None, which=EVAL_NAMES, integrator=None):
        n, m = self.n_x, self.n_u
        x = np.ascontiguousarray(x, dtype=self.np_dtype).reshape(-1, n)
        npts = x.shape[0]
        if u is not None:
            u = np.ascontiguousarray(u, dtype=self.np_dtype).reshape(-1, m)
            if u.shape[0] != npts:
                raise ValueError("x and u must hold the same number of points")
        shapes = {"f": (n,), "f_x": (n, n), "f_u": (n, m), "l": (), "l_x": (n,), "l_u": (m,), "l_xx": (n, n),
                  "l_ux": (m, n), "l_uu": (m, m), "l_f": (), "l_f_x": (n,), "l_f_xx": (n, n)}
        outs = {k: np.empty((npts,) + shapes[k], dtype=self.np_dtype) for k in which}
        integ = -1 if integrator is None else (INTEGRATORS[integrator] if isinstance(integrator, str) else integrator)
        args = [_ptr(outs[k]) if k in outs else None for k in self.EVAL_NAMES]
        self._chk(self.lib.ilqr_eval_points(self.h, integ, npts, _ptr(x), _ptr(u), *args))
        return outs

    # ---- MPC ----------------------------------------------------------------------------------
    def mpc_reset(self, x0, U_init):
        x0 = self._in(x0, (self.B, self.n_x))
        U_init = self._in(U_init, (self.B, self.n_u, self.N))
        self._chk(self.lib.ilqr_mpc_reset(self.h, _ptr(x0), _ptr(U_init)))

    def mpc_rearm(self, x0, U_init):
        """Restart the controller but keep X, K, U_ff of the previous solve (run_iLQR_MPC.py:95 warm-up carry)."""
        x0 = self._in(x0, (self.B, self.n_x))
        U_init = self._in(U_init, (self.B, self.n_u, self.N))
        self._chk(self.lib.ilqr_mpc_rearm(self.h, _ptr(x0), _ptr(U_init)))

    def mpc_run(self, n_steps):
        u = np.empty((n_steps, self.B, self.n_u), dtype=self.np_dtype)
        x = np.empty((n_steps, self.B, self.n_x), dtype=self.np_dtype)
        c = np.empty((n_steps, self.B), dtype=self.np_dtype)
        self._chk(self.lib.ilqr_mpc_run(self.h, int(n_steps), _ptr(u), _ptr(x), _ptr(c)))
        return u, x, c

    def status_reduce(self, dev_ptr):
        """Write {min cost, max |dcost|, #active, #converged} (4 doubles) to DEVICE memory at dev_ptr."""
        self._chk(self.lib.ilqr_status_reduce(self.h, C.c_void_p(int(dev_ptr))))

    # ---- measurement ------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._chk(self.lib.ilqr_timing_enable(self.h, int(bool(on))))

    def timing_reset(self):
        self._chk(self.lib.ilqr_timing_reset(self.h))

    def timing_get(self):
        ms = (C.c_double * len(PHASES))()
        n = (C.c_int64 * len(PHASES))()
        self._chk(self.lib.ilqr_timing_get(self.h, ms, n))
        return {p: (ms[i], n[i]) for i, p in enumerate(PHASES)}

    def algorithmic_bytes(self):
        b = (C.c_double * len(PHASES))()
        self._chk(self.lib.ilqr_algorithmic_bytes(self.h, b))
        return {p: b[i] for i, p in enumerate(PHASES)}

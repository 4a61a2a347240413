/*
 * ilqr_hip.h -- C-ABI of libilqr_hip.so: batched iLQR hot path on MI355X (gfx950).
 *
 * The reference (MohamedAbou-Taleb/Iterative-Linear-Quadratic-Regulator) has no
 * FFI / plugin API: its boundary is the Python class surface used by
 * python/run_iLQR_open_loop.py and python/run_iLQR_MPC.py.  Each entry point
 * below therefore cites the reference *function* it replaces (paths relative to
 * the reference root).  The Python side binds these with ctypes
 * (iterative-linear-quadratic-regulator_amd/_lib.py); INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes only; no torch / C++ types.
 *  - every `void*` host buffer holds scalars of the handle's dtype
 *    (ILQR_F32 -> float, ILQR_F64 -> double), C-contiguous, with a LEADING
 *    batch axis in front of the reference's own layout:
 *        x0   [B][n_x]            X  [B][n_x][N+1]      U    [B][n_u][N]
 *        U_ff [B][n_u][N]         K  [B][N][n_u][n_x]   cost [B]
 *  - the handle owns every device buffer and its stream; the caller owns every
 *    host pointer; no host pointer is retained after a call returns.
 *  - one handle <-> one device <-> one stream; a handle is not thread-safe,
 *    distinct handles may be used from distinct threads / processes
 *    (multi-GPU = one process and one handle per GPU).
 *  - return value: ILQR_OK (0) or an ilqr_status error; ilqr_last_error() gives
 *    the message.  Numerical events (line-search failure, non-PD Q_uu) are NOT
 *    errors: they are per-trajectory status words (ILQR_GET_STATUS), mirroring
 *    the reference's printed warnings (iLQR_class.py:304-311).
 *  - there is NO CPU fallback: without a usable gfx950 device ilqr_create fails
 *    with ILQR_ERR_NO_DEVICE.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_ABI_VERSION 3

typedef struct ilqr_solver_s* ilqr_handle;

typedef enum ilqr_status {
    ILQR_OK = 0,
    ILQR_ERR_INVALID_ARG = 1, /* bad shape / enum / NULL: the shim raises ValueError (iLQR_class.py:52, system_base.py:198) */
    ILQR_ERR_HIP = 2,         /* a HIP runtime call failed */
    ILQR_ERR_UNSUPPORTED = 3, /* (system, n_x, n_u, dtype) combination not compiled in */
    ILQR_ERR_NO_DEVICE = 4,   /* no gfx950 device visible: the product path has no CPU fallback */
    ILQR_ERR_STATE = 5        /* call sequence error (e.g. iterate before set_problem) */
} ilqr_status;

typedef enum ilqr_dtype { ILQR_F32 = 0, ILQR_F64 = 1 } ilqr_dtype;

/* Built-in systems (reference: python/class_files/systems/). */
typedef enum ilqr_system {
    ILQR_SYS_PENDULUM = 0,           /* pendulum_sys.py:12-98            n_x=2 n_u=1 */
    ILQR_SYS_UA_DOUBLE_PENDULUM = 1, /* UA_double_pendulum_sys.py:9-208  n_x=4 n_u=1 */
    ILQR_SYS_DOUBLE_PENDULUM = 2,    /* double_pendulum_sys.py:9-206     n_x=4 n_u=2 */
    ILQR_SYS_LINEAR = 3,             /* x_dot = A x + B u (matlab/CLASSES/Linear_iLQR_CLASS.m:56-60) */
    ILQR_SYS_CUSTOM = 4              /* user-defined System subclass (system_base.py:255-275): dynamics compiled into a
                                        plugin, see ilqr_create_custom; no system parameters in the block */
} ilqr_system;

/* Integrators (system_base.py:50-140).  ILQR_INT_DISCRETE takes the system's
 * map as the discrete step itself (x+ = A x + B u for ILQR_SYS_LINEAR). */
typedef enum ilqr_integrator {
    ILQR_INT_EULER = 0,
    ILQR_INT_MIDPOINT = 1,
    ILQR_INT_RK4 = 2,
    ILQR_INT_BACKWARD_EULER = 3,
    ILQR_INT_DISCRETE = 4
} ilqr_integrator;

/* Per-trajectory status word (ILQR_GET_STATUS). Low byte = state, bit 8 = flag. */
enum {
    ILQR_TRAJ_ACTIVE = 0,            /* still iterating */
    ILQR_TRAJ_CONVERGED = 1,         /* |cost - cost_prev| <= tol (iLQR_class.py:267) */
    ILQR_TRAJ_LINESEARCH_FAILED = 2, /* no alpha accepted (iLQR_class.py:304-307) */
    ILQR_TRAJ_MAXITER = 3,           /* ran maxiter iterations (iLQR_class.py:309-311) */
    ILQR_TRAJ_FLAG_NON_PD = 0x100    /* some Q_uu was not positive definite: LU fallback used */
};

enum {
    ILQR_FLAG_KEEP_ITERATING = 1, /* throughput mode: trajectories never leave ACTIVE (no convergence /
                                     line-search break), so every iteration does the full batch's work */
    ILQR_FLAG_NO_FUSE = 2,        /* keep linearise, sweep and acceptance step as separate launches over a materialised
                                     expansion inside ilqr_iterate / ilqr_solve / ilqr_mpc_run (see ILQR_PHASE_FUSED) */
    ILQR_FLAG_NO_PERSIST = 4      /* one launch per phase of an iteration and the host's loop around them instead of the
                                     persistent kernel (see ILQR_PHASE_PERSIST) */
};

/*
 * Parameter block (doubles, converted to the handle's dtype on upload):
 *   [ system parameters | x_target (n_x) | Q (n_x*n_x) | R (n_u*n_u) | Q_f (n_x*n_x) ]   row-major
 * system parameters:
 *   PENDULUM            g, l, d                                    (pendulum_sys.py:27-29)
 *   (UA_)DOUBLE_PENDULUM g, m1, m2, l1, l2, d1, d2, theta1, theta2 (UA_double_pendulum_sys.py:27-35)
 *   LINEAR              A (n_x*n_x), B (n_x*n_u)                   row-major
 * ilqr_param_count() returns the expected total length.
 */
typedef struct ilqr_config {
    uint32_t struct_size; /* = sizeof(ilqr_config) */
    int32_t n_x, n_u;     /* must match the system (checked) */
    int32_t horizon;      /* N: number of control steps (iLQR_class.py:46-47) */
    int32_t batch;        /* B: independent trajectories on this device */
    int32_t n_alpha;      /* line-search alphas rolled out in parallel per pass (1..16) */
    int32_t n_trials;     /* backtracking trials per iteration; reference: 10 (iLQR_class.py:281) */
    int32_t dtype;        /* ilqr_dtype */
    int32_t system;       /* ilqr_system */
    int32_t integrator;   /* ilqr_integrator of the optimiser model */
    int32_t plant_integrator; /* ilqr_integrator of the MPC plant (run_iLQR_MPC.py:68-75), or -1 */
    int32_t device;       /* HIP device ordinal */
    int32_t maxiter;      /* iLQR_class.py:24 */
    int32_t flags;        /* ILQR_FLAG_* */
    double dt;
    double tol;           /* iLQR_class.py:23 */
    double alpha_factor;  /* iLQR_class.py:25 */
    double min_alpha;     /* iLQR_class.py:26 */
    double mu;            /* Levenberg regularisation of Q_uu (build extension; 0 = reference) */
    const double* params; /* parameter block, see above */
    int32_t n_params;
    int32_t reserved;
    void* stream;         /* hipStream_t to launch on, or NULL: the handle creates its own */
} ilqr_config;

/* Selector for ilqr_get / ilqr_set. */
typedef enum ilqr_field {
    ILQR_X = 0,        /* [B][n_x][N+1]     iLQR.X     (iLQR_class.py:55) */
    ILQR_U = 1,        /* [B][n_u][N]       iLQR.U     (:56) */
    ILQR_K = 2,        /* [B][N][n_u][n_x]  iLQR.K     (:59) */
    ILQR_UFF = 3,      /* [B][n_u][N]       iLQR.U_ff  (:61) */
    ILQR_X0 = 4,       /* [B][n_x]          iLQR.x_0   (:30) */
    ILQR_COST = 5,     /* [B]  current total cost, handle dtype */
    ILQR_STATUS = 6,   /* [B]  int32 status words (get only) */
    ILQR_ITERS = 7,    /* [B]  int32 backward passes executed in the current solve (get only) */
    ILQR_ALPHA = 8,    /* [B]  alpha accepted in the last iteration, 0 if none; handle dtype (get only) */
    ILQR_TRIAL_COSTS = 9, /* [B][n_alpha] costs of the last line-search pass; handle dtype (get only) */
    ILQR_LIN = 10,     /* [B][N][E] raw expansion of the last ilqr_linearize, E = 2n^2+2nm+n+m+m^2, per step:
                          f_x (n*n) f_u (n*m) l_x (n) l_u (m) l_xx (n*n) l_ux (m*n) l_uu (m*m), row-major (get only) */
    ILQR_PLANT_X = 11, /* [B][n_x] MPC plant state */
    ILQR_PROBE = 12    /* 8 x int64 diagnostic clock stamps {shader cycles, 100 MHz ticks} of workgroup 0:
                          [0,1] backward sweep, [2,3] forward rollout; filled only when the environment variable
                          ILQR_CLOCK_PROBE is set at ilqr_create (get only) */
} ilqr_field;

/* Phases timed by ilqr_timing_* (HIP events recorded on the handle's stream). */
enum {
    ILQR_PHASE_LINEARIZE = 0,
    ILQR_PHASE_BACKWARD = 1,
    ILQR_PHASE_FORWARD = 2,
    ILQR_PHASE_SELECT = 3,
    ILQR_PHASE_OTHER = 4,
    ILQR_PHASE_FUSED = 5, /* acceptance step + linearisation + sweep as one kernel (the default inside ilqr_iterate /
                             ilqr_solve / ilqr_mpc_run for the n_u = 1 DPP systems; ILQR_NO_FUSE=1 keeps the stages apart) */
    ILQR_PHASE_PERSIST = 6, /* the whole iteration loop of a workgroup's trajectories as one launch: ilqr_iterate(n) = one launch
                               of n iterations, ilqr_solve and ilqr_mpc_run one launch each (ILQR_FLAG_NO_PERSIST keeps one
                               fused launch + one rollout launch per iteration and the host's loop) */
    ILQR_N_PHASES = 7
};

/* ---- library-level ------------------------------------------------------ */
int ilqr_abi_version(void);
int ilqr_device_count(int* count);
/* expected n_params for (system, n_x, n_u), or -1 if the combination is unknown */
int ilqr_param_count(int system, int n_x, int n_u);
/* 1 if kernels for (system, n_x, n_u, dtype) are compiled into this build */
int ilqr_is_supported(int system, int n_x, int n_u, int dtype);
/* message of the last failure on this handle (or of the last failed ilqr_create when h == NULL) */
const char* ilqr_last_error(ilqr_handle h);

/* ---- lifetime: replaces iLQR.__init__ state allocation (iLQR_class.py:18-75)
 *      and System.__init__ (systems/system_base.py:25-251) ------------------ */
int ilqr_create(ilqr_handle* out, const ilqr_config* cfg);
/* Same, for a user-defined system (the reference's subclass contract: _f_cont_fcn, system_base.py:255-275) whose
 * continuous dynamics and Jacobians were generated and compiled into the plugin shared object at `plugin_path`
 * (iterative-linear-quadratic-regulator_amd/systems/custom_sys.py builds it from the subclass with hipcc against the
 * same kernel templates).  cfg->system must be ILQR_SYS_CUSTOM; the cost is the reference's quadratic form. */
int ilqr_create_custom(ilqr_handle* out, const ilqr_config* cfg, const char* plugin_path);
int ilqr_destroy(ilqr_handle h);
int ilqr_sync(ilqr_handle h);

/* ---- state: iLQR attributes read/written by the drivers
 *      (run_iLQR_open_loop.py:78-87, run_iLQR_MPC.py:118,121) ---------------- */
/* fresh solver as after the constructor: x_0, U = U_init, X = K = U_ff = 0 (iLQR_class.py:55-61) */
int ilqr_set_problem(ilqr_handle h, const void* x0, const void* U_init);
int ilqr_set(ilqr_handle h, int field, const void* src, size_t bytes);
int ilqr_get(ilqr_handle h, int field, void* dst, size_t bytes);

/* ---- the hot path, stage by stage (asynchronous on the handle's stream) ---- */
/* initial rollout with alpha = 0 through the carried K, X (iLQR_class.py:257-259);
 * also resets status/iteration counters: the head of optimize_trajectory */
int ilqr_initial_rollout(ilqr_handle h);
/* A_t, B_t, l_x, l_u, l_xx, l_ux, l_uu at every (b, t) and terminal l_f_x, l_f_xx
 * (iLQR_class.py:318-331 -> system_base.py:203-219) */
int ilqr_linearize(ilqr_handle h);
/* backward Riccati sweep over the expansion -> K, U_ff (iLQR_class.py:79-161) */
int ilqr_backward(ilqr_handle h);
/* candidate rollouts for alphas[0..n) in parallel, n <= n_alpha (iLQR_class.py:164-247) */
int ilqr_forward(ilqr_handle h, const double* alphas, int n);
/* backtracking acceptance "first alpha with cost_new <= cost" + convergence bookkeeping
 * (iLQR_class.py:267-271, 279-307) */
int ilqr_select(ilqr_handle h);
/* n_iters x (linearize, backward, forward over all trial alphas, select), no host sync.  The acceptance step of the
 * LAST iteration may still be pending when the call returns (the next iteration's kernel runs it for its own
 * trajectories); every entry point that reads or writes solver state completes it first, ilqr_flush does so explicitly. */
int ilqr_iterate(ilqr_handle h, int n_iters);
/* enqueue whatever bookkeeping ilqr_iterate deferred (iLQR_class.py:289-307 of its last iteration); asynchronous */
int ilqr_flush(ilqr_handle h);

/* ---- whole solve: iLQR.optimize_trajectory (iLQR_class.py:250-313), synchronous.
 *      iters_out [B] int32 and cost_out [B] (handle dtype) may be NULL. ------- */
int ilqr_solve(ilqr_handle h, int32_t* iters_out, void* cost_out);

/* ---- pure functional calls used by the drivers' warm-up and by parity tests;
 *      they do not touch the solver state ------------------------------------ */
/* iLQR.backward_pass(X, U) -> (U_ff, K)   (iLQR_class.py:68, 122-161) */
int ilqr_backward_pass(ilqr_handle h, const void* X, const void* U, void* U_ff_out, void* K_out);
/* The sweep of iLQR.backward_pass alone (iLQR_class.py:136-151) on an expansion the CALLER computed -- its own
 * autodiff, an identified model, time-varying LQ data -- instead of _get_all_derivatives_for_backward_pass
 * (iLQR_class.py:318-331).  lin [B][N][E]: per step f_x (n*n, row-major), f_u (n*m), l_x (n), l_u (m), l_xx (n*n),
 * l_ux (m*n), l_uu (m*m), E = 2n^2 + 2nm + n + m + m^2 (the ILQR_LIN record); term [B][n + n*n]: V_x, V_xx at
 * t = N (l_f_x, l_f_xx, iLQR_class.py:136-138).  Outputs as ilqr_backward_pass; cfg->mu applies.  The system the
 * handle was created for only fixes (n_x, n_u). */
int ilqr_backward_tensors(ilqr_handle h, const void* lin, const void* term, void* U_ff_out, void* K_out);
/* iLQR.forward_pass(x_0, alpha, X_old, U_old, U_ff, K) -> (X_new, U_new, cost)  (iLQR_class.py:75, 193-247) */
int ilqr_forward_pass(ilqr_handle h, const void* x0, double alpha, const void* X_old, const void* U_old,
                      const void* U_ff, const void* K, void* X_new, void* U_new, void* cost);
/* The 12 System callables at npts points (system_base.py:223-251); any output may be NULL.
 * x [npts][n_x], u [npts][n_u];  f [npts][n_x], f_x [npts][n_x][n_x], f_u [npts][n_x][n_u],
 * l [npts], l_x [npts][n_x], l_u [npts][n_u], l_xx [npts][n_x][n_x], l_ux [npts][n_u][n_x],
 * l_uu [npts][n_u][n_u], l_f [npts], l_f_x [npts][n_x], l_f_xx [npts][n_x][n_x].
 * integrator < 0 selects the handle's optimiser integrator. */
int ilqr_eval_points(ilqr_handle h, int integrator, int npts, const void* x, const void* u,
                     void* f, void* f_x, void* f_u, void* l, void* l_x, void* l_u, void* l_xx,
                     void* l_ux, void* l_uu, void* l_f, void* l_f_x, void* l_f_xx);

/* ---- MPC step (run_iLQR_MPC.py:116-143), device-resident ------------------- */
/* plant state <- x0, warm start <- U_init, fresh solver state */
int ilqr_mpc_reset(ilqr_handle h, const void* x0, const void* U_init);
/* Restart of the controller on a solver that has already solved: plant state and x_0 <- x0, warm start <- U_init,
 * while X, K, U_ff are KEPT.  This is the state run_iLQR_MPC.py enters its loop with: its "JIT warm-up" is one full
 * optimize_trajectory() on the same solver object (run_iLQR_MPC.py:95), so step 0's alpha = 0 rollout runs through
 * the warm-up's gains, u = U_init + K_warm (x - X_warm) (iLQR_class.py:257-259).  ilqr_mpc_reset is the cold start of
 * run_iLQR_UA_MPC.py, whose warm-up calls the pure functions only (:114-124). */
int ilqr_mpc_rearm(ilqr_handle h, const void* x0, const void* U_init);
/* n_steps x { x_0 <- plant state; U <- warm start; solve; u0 = U[:,0]; plant step with
 * plant_integrator; warm start <- shift(U) repeating the last column }.
 * u_out [n_steps][B][n_u], x_out [n_steps][B][n_x] (state after each step), cost_out [n_steps][B]; may be NULL */
int ilqr_mpc_run(ilqr_handle h, int n_steps, void* u_out, void* x_out, void* cost_out);

/* ---- multi-GPU hook -------------------------------------------
 * Writes 4 doubles to DEVICE memory `dev_out4` on the handle's stream:
 *   { min cost, max |cost - cost_prev|, #trajectories still active, #converged }
 * of this handle's shard.  The host side all-reduces them over RCCL (MIN / MAX / SUM / SUM);
 * it is the only inter-GPU exchange of the path -- trajectories never interact. */
int ilqr_status_reduce(ilqr_handle h, void* dev_out4);

/* ---- measurement ------------------------------------------------------------ */
int ilqr_timing_enable(ilqr_handle h, int on);
int ilqr_timing_reset(ilqr_handle h);
/* total milliseconds and launch counts per ILQR_PHASE_* since the last reset (synchronises) */
int ilqr_timing_get(ilqr_handle h, double ms[ILQR_N_PHASES], int64_t launches[ILQR_N_PHASES]);
/* algorithmic HBM bytes of one launch of each phase (SURVEY.md 8d formulas) */
int ilqr_algorithmic_bytes(ilqr_handle h, double bytes[ILQR_N_PHASES]);

#ifdef __cplusplus
}
#endif
#endif /* ILQR_HIP_H */

/*
 * ilqr_oracle.c -- plain-C CPU restatement of the reference iLQR hot path.
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY: linked by nothing in the product path.
 *
 * Follows /root/reference/python/class_files/iLQR_class.py (backward step :79-119,
 * sweep :122-161, rollout :164-247, outer loop + backtracking :250-313) and
 * systems/system_base.py (integrators :50-74, backward Euler :88-140, IFT
 * Jacobians :146-188), systems/pendulum_sys.py:60-98,
 * systems/UA_double_pendulum_sys.py:84-208, systems/double_pendulum_sys.py:84-206.
 * Written independently of the NumPy files in oracle/ (different decomposition: explicit loops,
 * Gaussian elimination) so the two restatements check each other
 * (tests/test_oracle_c.py).  The reference itself is Python/JAX and cannot be
 * compiled, so there is no oracle/_ref.
 *
 * Compiled twice: -DREAL=double (parity precision) and -DREAL=float (the
 * reference's JAX default precision).  One trajectory per call; the caller
 * (oracle/c_oracle.py, bench.py cpu_baseline) loops / forks over the batch.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#ifndef SUFFIX
#define SUFFIX _f64
#endif
#define FN(name) CAT(name, SUFFIX)

#define MAXN 16
#define MAXM 8

enum { SYS_PENDULUM = 0, SYS_UA_DP = 1, SYS_DP = 2, SYS_LINEAR = 3 };
enum { INT_EULER = 0, INT_MIDPOINT = 1, INT_RK4 = 2, INT_BE = 3, INT_DISCRETE = 4 };

typedef struct {
    int system, integrator, n, m;
    REAL dt;
    /* physical parameters (ABI order, include/ilqr_hip.h) */
    REAL sp[MAXN * MAXN + MAXN * MAXM];
    REAL xt[MAXN], Q[MAXN * MAXN], R[MAXM * MAXM], Qf[MAXN * MAXN];
} model_t;

static REAL rsin(REAL x) { return sizeof(REAL) == 4 ? (REAL)sinf((float)x) : (REAL)sin((double)x); }
static REAL rcos(REAL x) { return sizeof(REAL) == 4 ? (REAL)cosf((float)x) : (REAL)cos((double)x); }
static REAL rsqrt_(REAL x) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static REAL rabs(REAL x) { return x < 0 ? -x : x; }

/* solve A X = B (n x n, n x r) by Gaussian elimination with partial pivoting; A, B overwritten */
static void gesv(int n, int r, REAL* A, REAL* B) {
    for (int k = 0; k < n; ++k) {
        int p = k;
        REAL best = rabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (rabs(A[i * n + k]) > best) { best = rabs(A[i * n + k]); p = i; }
        if (p != k) {
            for (int j = 0; j < n; ++j) { REAL t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < r; ++j) { REAL t = B[k * r + j]; B[k * r + j] = B[p * r + j]; B[p * r + j] = t; }
        }
        for (int i = k + 1; i < n; ++i) {
            REAL l = A[i * n + k] / A[k * n + k];
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
            for (int j = 0; j < r; ++j) B[i * r + j] -= l * B[k * r + j];
        }
    }
    for (int k = n - 1; k >= 0; --k)
        for (int j = 0; j < r; ++j) {
            REAL acc = B[k * r + j];
            for (int i = k + 1; i < n; ++i) acc -= A[k * n + i] * B[i * r + j];
            B[k * r + j] = acc / A[k * n + k];
        }
}

/* continuous dynamics xd = f_c(x, u); if Jx != NULL also d f_c/dx (n x n) and d f_c/du (n x m) */
static void fcont(const model_t* M, const REAL* x, const REAL* u, REAL* xd, REAL* Jx, REAL* Ju) {
    const int n = M->n, m = M->m;
    if (M->system == SYS_PENDULUM) { /* pendulum_sys.py:60-75 */
        const REAL g = M->sp[0], l = M->sp[1], d = M->sp[2];
        xd[0] = x[1];
        xd[1] = u[0] - d * x[1] - (g / l) * rsin(x[0]);
        if (Jx) {
            Jx[0] = 0; Jx[1] = 1; Jx[2] = -(g / l) * rcos(x[0]); Jx[3] = -d;
            Ju[0] = 0; Ju[1] = 1;
        }
        return;
    }
    if (M->system == SYS_LINEAR) {
        const REAL* A = M->sp;
        const REAL* B = M->sp + n * n;
        for (int i = 0; i < n; ++i) {
            REAL acc = 0;
            for (int j = 0; j < n; ++j) acc += A[i * n + j] * x[j];
            for (int j = 0; j < m; ++j) acc += B[i * m + j] * u[j];
            xd[i] = acc;
        }
        if (Jx) { memcpy(Jx, A, sizeof(REAL) * n * n); memcpy(Ju, B, sizeof(REAL) * n * m); }
        return;
    }
    /* double pendulum: UA_double_pendulum_sys.py:84-208 */
    const REAL g = M->sp[0], m1 = M->sp[1], m2 = M->sp[2], l1 = M->sp[3], l2 = M->sp[4], d1 = M->sp[5],
               d2 = M->sp[6], th1 = M->sp[7], th2 = M->sp[8];
    const REAL q1 = x[0], q2 = x[1], q1d = x[2], q2d = x[3];
    const REAL s1 = rsin(q1), s2 = rsin(q2), s12 = rsin(q1 + q2), c2 = rcos(q2);
    const REAL m11 = (m1 * l1 * l1) / 4 + m2 * l1 * l1 + (m2 * l2 * l2) / 4 + m2 * l1 * l2 * c2 + th1 + th2;
    const REAL m12 = (m2 * l2 * l2) / 4 + (m2 * l1 * l2 * c2) / 2 + th2;
    const REAL m22 = (m2 * l2 * l2) / 4 + th2;
    REAL h[2];
    h[0] = (m2 * l1 * l2 * s2 * (2 * q1d * q2d + q2d * q2d)) / 2 - m2 * g * (l2 * s12 / 2 + l1 * s1) -
           (m1 * g * l1 * s1) / 2 - d1 * q1d + u[0];
    h[1] = -(m2 * l1 * l2 * s2 * (q1d * q1d)) / 2 - m2 * g * (l2 * s12) / 2 - d2 * q2d + (m == 2 ? u[1] : 0);
    REAL Mm[4] = {m11, m12, m12, m22}, qdd[2] = {h[0], h[1]};
    gesv(2, 1, Mm, qdd);
    xd[0] = q1d; xd[1] = q2d; xd[2] = qdd[0]; xd[3] = qdd[1];
    if (!Jx) return;
    const REAL c1 = rcos(q1), c12 = rcos(q1 + q2), a = m2 * l1 * l2;
    /* rhs of M dqdd = dh - dM qdd, columns: q1 q2 q1d q2d then the m controls */
    REAL rhs[2 * 6];
    const int r = 4 + m;
    rhs[0 * r + 0] = -m2 * g * (l2 * c12 / 2 + l1 * c1) - m1 * g * l1 * c1 / 2;
    rhs[1 * r + 0] = -m2 * g * l2 * c12 / 2;
    rhs[0 * r + 1] = a * c2 * (2 * q1d * q2d + q2d * q2d) / 2 - m2 * g * l2 * c12 / 2 -
                     (-a * s2 * qdd[0] - a * s2 / 2 * qdd[1]);
    rhs[1 * r + 1] = -a * c2 * q1d * q1d / 2 - m2 * g * l2 * c12 / 2 - (-a * s2 / 2 * qdd[0]);
    rhs[0 * r + 2] = a * s2 * q2d - d1;
    rhs[1 * r + 2] = -a * s2 * q1d;
    rhs[0 * r + 3] = a * s2 * (q1d + q2d);
    rhs[1 * r + 3] = -d2;
    rhs[0 * r + 4] = 1; rhs[1 * r + 4] = 0;
    if (m == 2) { rhs[0 * r + 5] = 0; rhs[1 * r + 5] = 1; }
    REAL M2[4] = {m11, m12, m12, m22};
    gesv(2, r, M2, rhs);
    memset(Jx, 0, sizeof(REAL) * 16);
    memset(Ju, 0, sizeof(REAL) * 4 * m);
    Jx[0 * 4 + 2] = 1; Jx[1 * 4 + 3] = 1;
    for (int j = 0; j < 4; ++j) { Jx[2 * 4 + j] = rhs[0 * r + j]; Jx[3 * 4 + j] = rhs[1 * r + j]; }
    for (int j = 0; j < m; ++j) { Ju[2 * m + j] = rhs[0 * r + 4 + j]; Ju[3 * m + j] = rhs[1 * r + 4 + j]; }
}

static void backward_euler(const model_t* M, const REAL* x, const REAL* u, REAL* xn) {
    const int n = M->n;
    const REAL dt = M->dt;
    REAL k[MAXN], Jx[MAXN * MAXN], Ju[MAXN * MAXM], J[MAXN * MAXN], F[MAXN];
    fcont(M, x, u, k, 0, 0);
    for (int i = 0; i < n; ++i) xn[i] = x[i] + dt * k[i];
    fcont(M, xn, u, k, Jx, Ju);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) J[i * n + j] = (i == j) - dt * Jx[i * n + j];
    REAL nrm = 0;
    for (int i = 0; i < n; ++i) { F[i] = xn[i] - x[i] - dt * k[i]; nrm += F[i] * F[i]; }
    nrm = rsqrt_(nrm);
    int it = 0;
    while (nrm > (REAL)1e-5 && it < 20) {
        REAL Jc[MAXN * MAXN], d[MAXN];
        memcpy(Jc, J, sizeof(REAL) * n * n);
        for (int i = 0; i < n; ++i) d[i] = -F[i];
        gesv(n, 1, Jc, d);
        for (int i = 0; i < n; ++i) xn[i] += d[i];
        fcont(M, xn, u, k, 0, 0);
        nrm = 0;
        for (int i = 0; i < n; ++i) { F[i] = xn[i] - x[i] - dt * k[i]; nrm += F[i] * F[i]; }
        nrm = rsqrt_(nrm);
        ++it;
    }
}

/* discrete step; with fx != NULL also the exact Jacobians of the discrete map */
static void step(const model_t* M, const REAL* x, const REAL* u, REAL* xn, REAL* fx, REAL* fu) {
    const int n = M->n, m = M->m;
    const REAL dt = M->dt;
    REAL k[4][MAXN], Kx[4][MAXN * MAXN], Ku[4][MAXN * MAXM];
    if (M->integrator == INT_DISCRETE) { fcont(M, x, u, xn, fx, fu); return; }
    if (M->integrator == INT_BE) {
        backward_euler(M, x, u, xn);
        if (!fx) return;
        REAL kk[MAXN], Jx[MAXN * MAXN], Ju[MAXN * MAXM], J[MAXN * MAXN], rhs[MAXN * (MAXN + MAXM)];
        fcont(M, xn, u, kk, Jx, Ju);
        const int r = n + m;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) { J[i * n + j] = (i == j) - dt * Jx[i * n + j]; rhs[i * r + j] = (i == j); }
            for (int j = 0; j < m; ++j) rhs[i * r + n + j] = dt * Ju[i * m + j];
        }
        gesv(n, r, J, rhs);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) fx[i * n + j] = rhs[i * r + j];
            for (int j = 0; j < m; ++j) fu[i * m + j] = rhs[i * r + n + j];
        }
        return;
    }
    const int ns = (M->integrator == INT_EULER) ? 1 : (M->integrator == INT_MIDPOINT ? 2 : 4);
    const REAL c[4] = {0, dt / 2, dt / 2, dt};
    for (int s = 0; s < ns; ++s) {
        REAL xs[MAXN];
        for (int i = 0; i < n; ++i) xs[i] = x[i] + (s ? c[s] * k[s - 1][i] : 0);
        if (!fx) { fcont(M, xs, u, k[s], 0, 0); continue; }
        REAL Jx[MAXN * MAXN], Ju[MAXN * MAXM];
        fcont(M, xs, u, k[s], Jx, Ju);
        if (s == 0) {
            memcpy(Kx[0], Jx, sizeof(REAL) * n * n);
            memcpy(Ku[0], Ju, sizeof(REAL) * n * m);
        } else {
            /* Kx = Jx (I + c Kx_prev), Ku = Jx (c Ku_prev) + Ju */
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < n; ++j) {
                    REAL acc = 0;
                    for (int q = 0; q < n; ++q) acc += Jx[i * n + q] * ((q == j) + c[s] * Kx[s - 1][q * n + j]);
                    Kx[s][i * n + j] = acc;
                }
                for (int j = 0; j < m; ++j) {
                    REAL acc = 0;
                    for (int q = 0; q < n; ++q) acc += Jx[i * n + q] * (c[s] * Ku[s - 1][q * m + j]);
                    Ku[s][i * m + j] = acc + Ju[i * m + j];
                }
            }
        }
    }
    const REAL w4[4] = {1, 2, 2, 1};
    for (int i = 0; i < n; ++i) {
        if (ns == 1) xn[i] = x[i] + k[0][i] * dt;
        else if (ns == 2) xn[i] = x[i] + dt * k[1][i];
        else xn[i] = x[i] + (dt / 6) * (k[0][i] + 2 * k[1][i] + 2 * k[2][i] + k[3][i]);
    }
    if (!fx) return;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            REAL v;
            if (ns == 1) v = dt * Kx[0][i * n + j];
            else if (ns == 2) v = dt * Kx[1][i * n + j];
            else {
                v = 0;
                for (int s = 0; s < 4; ++s) v += w4[s] * Kx[s][i * n + j];
                v *= dt / 6;
            }
            fx[i * n + j] = (i == j) + v;
        }
        for (int j = 0; j < m; ++j) {
            REAL v;
            if (ns == 1) v = dt * Ku[0][i * m + j];
            else if (ns == 2) v = dt * Ku[1][i * m + j];
            else {
                v = 0;
                for (int s = 0; s < 4; ++s) v += w4[s] * Ku[s][i * m + j];
                v *= dt / 6;
            }
            fu[i * m + j] = v;
        }
    }
}

static REAL stage_cost(const model_t* M, const REAL* x, const REAL* u) {
    const int n = M->n, m = M->m;
    REAL cx = 0, cu = 0;
    for (int i = 0; i < n; ++i) {
        REAL r = 0;
        for (int j = 0; j < n; ++j) r += M->Q[i * n + j] * (x[j] - M->xt[j]);
        cx += (x[i] - M->xt[i]) * r;
    }
    for (int i = 0; i < m; ++i) {
        REAL r = 0;
        for (int j = 0; j < m; ++j) r += M->R[i * m + j] * u[j];
        cu += u[i] * r;
    }
    return ((REAL)0.5 * cx + (REAL)0.5 * cu) * M->dt;
}

static REAL terminal_cost(const model_t* M, const REAL* x) {
    const int n = M->n;
    REAL c = 0;
    for (int i = 0; i < n; ++i) {
        REAL r = 0;
        for (int j = 0; j < n; ++j) r += M->Qf[i * n + j] * (x[j] - M->xt[j]);
        c += (x[i] - M->xt[i]) * r;
    }
    return (REAL)0.5 * c;
}

/* ---- exported ------------------------------------------------------------------------- */

/* params: ABI block [sys | x_target | Q | R | Qf] as doubles */
void* FN(oracle_model_create)(int system, int integrator, int n, int m, double dt, const double* params, int nsys) {
    model_t* M = (model_t*)calloc(1, sizeof(model_t));
    M->system = system; M->integrator = integrator; M->n = n; M->m = m; M->dt = (REAL)dt;
    const double* p = params;
    for (int i = 0; i < nsys; ++i) M->sp[i] = (REAL)p[i];
    p += nsys;
    for (int i = 0; i < n; ++i) M->xt[i] = (REAL)p[i];
    p += n;
    for (int i = 0; i < n * n; ++i) M->Q[i] = (REAL)p[i];
    p += n * n;
    for (int i = 0; i < m * m; ++i) M->R[i] = (REAL)p[i];
    p += m * m;
    for (int i = 0; i < n * n; ++i) M->Qf[i] = (REAL)p[i];
    return M;
}
void FN(oracle_model_destroy)(void* M) { free(M); }

/* X (n, N+1), U (m, N) -> U_ff (m, N), K (N, m, n); all (dim, time) like the reference */
void FN(oracle_backward)(const void* Mv, int N, const REAL* X, const REAL* U, REAL* Uff, REAL* K) {
    const model_t* M = (const model_t*)Mv;
    const int n = M->n, m = M->m;
    REAL Vx[MAXN], Vxx[MAXN * MAXN], x[MAXN], u[MAXM];
    for (int i = 0; i < n; ++i) x[i] = X[i * (N + 1) + N];
    for (int i = 0; i < n; ++i) {
        REAL r = 0;
        for (int j = 0; j < n; ++j) {
            const REAL qs = (REAL)0.5 * (M->Qf[i * n + j] + M->Qf[j * n + i]);
            r += qs * (x[j] - M->xt[j]);
            Vxx[i * n + j] = qs;
        }
        Vx[i] = r;
    }
    for (int t = N - 1; t >= 0; --t) {
        REAL xn[MAXN], fx[MAXN * MAXN], fu[MAXN * MAXM];
        for (int i = 0; i < n; ++i) x[i] = X[i * (N + 1) + t];
        for (int i = 0; i < m; ++i) u[i] = U[i * N + t];
        step(M, x, u, xn, fx, fu);
        REAL Qx[MAXN], Qu[MAXM], Qxx[MAXN * MAXN], Qux[MAXM * MAXN], Quu[MAXM * MAXM], P[MAXN * MAXN], Pu[MAXM * MAXN];
        for (int j = 0; j < n; ++j) {
            REAL lx = 0, acc = 0;
            for (int q = 0; q < n; ++q) {
                lx += (REAL)0.5 * (M->Q[j * n + q] + M->Q[q * n + j]) * (x[q] - M->xt[q]);
                acc += fx[q * n + j] * Vx[q];
            }
            Qx[j] = lx * M->dt + acc;
        }
        for (int j = 0; j < m; ++j) {
            REAL lu = 0, acc = 0;
            for (int q = 0; q < m; ++q) lu += (REAL)0.5 * (M->R[j * m + q] + M->R[q * m + j]) * u[q];
            for (int q = 0; q < n; ++q) acc += fu[q * m + j] * Vx[q];
            Qu[j] = lu * M->dt + acc;
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                REAL acc = 0;
                for (int q = 0; q < n; ++q) acc += fx[q * n + i] * Vxx[q * n + j];
                P[i * n + j] = acc;
            }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) {
                REAL acc = 0;
                for (int q = 0; q < n; ++q) acc += fu[q * m + i] * Vxx[q * n + j];
                Pu[i * n + j] = acc;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                REAL acc = 0;
                for (int q = 0; q < n; ++q) acc += P[i * n + q] * fx[q * n + j];
                Qxx[i * n + j] = (REAL)0.5 * (M->Q[i * n + j] + M->Q[j * n + i]) * M->dt + acc;
            }
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) {
                REAL acc = 0;
                for (int q = 0; q < n; ++q) acc += Pu[i * n + q] * fx[q * n + j];
                Qux[i * n + j] = acc;
            }
            for (int j = 0; j < m; ++j) {
                REAL acc = 0;
                for (int q = 0; q < n; ++q) acc += Pu[i * n + q] * fu[q * m + j];
                Quu[i * m + j] = (REAL)0.5 * (M->R[i * m + j] + M->R[j * m + i]) * M->dt + acc;
            }
        }
        /* [K | k] = -Quu^-1 [Qux | Qu]  (iLQR_class.py:109-110) */
        REAL A[MAXM * MAXM], rhs[MAXM * (MAXN + 1)];
        memcpy(A, Quu, sizeof(REAL) * m * m);
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) rhs[i * (n + 1) + j] = Qux[i * n + j];
            rhs[i * (n + 1) + n] = Qu[i];
        }
        gesv(m, n + 1, A, rhs);
        REAL Kt[MAXM * MAXN], kt[MAXM];
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) Kt[i * n + j] = -rhs[i * (n + 1) + j];
            kt[i] = -rhs[i * (n + 1) + n];
        }
        /* V_x = Q_x + K'Q_u ; V_xx = Q_xx + Q_ux'K  (:113-114) */
        for (int i = 0; i < n; ++i) {
            REAL acc = 0;
            for (int s = 0; s < m; ++s) acc += Kt[s * n + i] * Qu[s];
            Vx[i] = Qx[i] + acc;
            for (int j = 0; j < n; ++j) {
                REAL a2 = 0;
                for (int s = 0; s < m; ++s) a2 += Qux[s * n + i] * Kt[s * n + j];
                Vxx[i * n + j] = Qxx[i * n + j] + a2;
            }
        }
        for (int i = 0; i < m; ++i) {
            Uff[i * N + t] = kt[i];
            for (int j = 0; j < n; ++j) K[(t * m + i) * n + j] = Kt[i * n + j];
        }
    }
}

/* rollout (iLQR_class.py:193-247); returns the total cost */
REAL FN(oracle_forward)(const void* Mv, int N, const REAL* x0, REAL alpha, const REAL* Xo, const REAL* Uo,
                        const REAL* Uff, const REAL* K, REAL* Xn, REAL* Un) {
    const model_t* M = (const model_t*)Mv;
    const int n = M->n, m = M->m;
    REAL x[MAXN], u[MAXM], xn[MAXN], cost = 0;
    for (int i = 0; i < n; ++i) x[i] = x0[i];
    for (int t = 0; t < N; ++t) {
        for (int j = 0; j < m; ++j) {
            REAL fb = 0;
            for (int i = 0; i < n; ++i) fb += K[(t * m + j) * n + i] * (x[i] - Xo[i * (N + 1) + t]);
            u[j] = Uo[j * N + t] + alpha * Uff[j * N + t] + fb;
        }
        for (int i = 0; i < n; ++i) Xn[i * (N + 1) + t] = x[i];
        for (int j = 0; j < m; ++j) Un[j * N + t] = u[j];
        cost += stage_cost(M, x, u);
        step(M, x, u, xn, 0, 0);
        for (int i = 0; i < n; ++i) x[i] = xn[i];
    }
    for (int i = 0; i < n; ++i) Xn[i * (N + 1) + N] = x[i];
    return cost + terminal_cost(M, x);
}

/*
 * optimize_trajectory (iLQR_class.py:250-313).  State X, U, Uff, K is carried in and out (quirk Q1).
 * fixed_iters > 0: throughput mode -- run exactly that many iterations, never break (bench.py cpu_baseline).
 * Returns the number of backward passes executed; *status: 1 converged, 2 line-search failed, 3 maxiter.
 * alpha_hist / cost_hist (may be NULL): [maxiter] accepted alpha (0 = none) and cost after each iteration executed.
 */
int FN(oracle_solve_hist)(const void* Mv, int N, const REAL* x0, REAL* X, REAL* U, REAL* Uff, REAL* K, double tol,
                          int maxiter, double alpha_factor, double min_alpha, int n_trials, int fixed_iters,
                          REAL* cost_out, int* status, double* alpha_hist, REAL* cost_hist) {
    const model_t* M = (const model_t*)Mv;
    const int n = M->n, m = M->m;
    REAL* Xn = (REAL*)malloc(sizeof(REAL) * n * (N + 1));
    REAL* Un = (REAL*)malloc(sizeof(REAL) * m * N);
    REAL cost = FN(oracle_forward)(M, N, x0, 0, X, U, Uff, K, Xn, Un);
    memcpy(X, Xn, sizeof(REAL) * n * (N + 1));
    memcpy(U, Un, sizeof(REAL) * m * N);
    REAL cost_prev = cost;
    int iters = 0, st = 3;
    const int total = fixed_iters > 0 ? fixed_iters : maxiter;
    for (int i = 0; i < total; ++i) {
        if (fixed_iters <= 0 && i > 0 && rabs(cost - cost_prev) <= (REAL)tol) { st = 1; break; }
        cost_prev = cost;
        FN(oracle_backward)(M, N, X, U, Uff, K);
        ++iters;
        double alpha = 1.0;
        int accepted = 0;
        for (int j = 0; j < n_trials; ++j) {
            REAL c = FN(oracle_forward)(M, N, x0, (REAL)alpha, X, U, Uff, K, Xn, Un);
            if (c <= cost) {
                memcpy(X, Xn, sizeof(REAL) * n * (N + 1));
                memcpy(U, Un, sizeof(REAL) * m * N);
                cost = c;
                accepted = 1;
                break;
            }
            alpha *= alpha_factor;
            if (alpha < min_alpha) break;
        }
        /* per-iteration trace for the parity tests: accepted alpha (0 = none) and the cost after the iteration */
        if (alpha_hist) alpha_hist[i] = accepted ? alpha : 0.0;
        if (cost_hist) cost_hist[i] = cost;
        if (!accepted && fixed_iters <= 0) { st = 2; break; }
    }
    free(Xn);
    free(Un);
    *cost_out = cost;
    *status = st;
    return iters;
}

int FN(oracle_solve)(const void* Mv, int N, const REAL* x0, REAL* X, REAL* U, REAL* Uff, REAL* K, double tol,
                     int maxiter, double alpha_factor, double min_alpha, int n_trials, int fixed_iters,
                     REAL* cost_out, int* status) {
    return FN(oracle_solve_hist)(Mv, N, x0, X, U, Uff, K, tol, maxiter, alpha_factor, min_alpha, n_trials, fixed_iters,
                                 cost_out, status, 0, 0);
}

/* single point: f, f_x, f_u (tests) */
void FN(oracle_step)(const void* Mv, const REAL* x, const REAL* u, REAL* xn, REAL* fx, REAL* fu) {
    step((const model_t*)Mv, x, u, xn, fx, fu);
}

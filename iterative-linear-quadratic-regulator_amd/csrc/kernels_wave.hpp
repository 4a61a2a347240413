// kernels_wave.hpp -- wave-cooperative kernels for n_x > 4 (BASELINE config c5: synthetic
// linear-quadratic system, n_x = 16, n_u = 8, N = 500).
//
// With n = 16 a single lane cannot hold the step's operands (V_xx, A_t, P, Q_xx are 256 scalars
// each), so here ONE WAVE owns one trajectory and the (n, m)-sized tiles of a timestep are staged
// in LDS: the 64 lanes split every small dense product of iLQR_class.py:100-114 by output element,
// read their operands from LDS, and meet at a workgroup barrier (the workgroup IS the wave) between
// the dependent phases.  Q_uu is factored by Cholesky (LU with partial pivoting if it is not positive
// definite, as the reference's solve would do), and the m+... right-hand sides [Q_ux | Q_u] are solved
// one column per lane.
//
// Expansion layout for these systems: lin[N][B][E] (one contiguous E-scalar record per (t, b), so a
// wave streams its trajectory's tile with fully coalesced loads), E = 2n^2 + 2nm + n + m + m^2 in
// ILQR_LIN order.  The next step's record is prefetched into registers while the current one computes.
#pragma once
#include "kernels.hpp"

namespace ilqr {

// ---------------------------------------------------------------------------
// linearize for linear systems: one wave per (b, t); every lane writes E/64 scalars of the record.
// f_x = A (discrete) or I + dt*A (euler), f_u = B or dt*B; cost terms as in Cost<>.
// ---------------------------------------------------------------------------
template <typename T, int NX, int NU>
__global__ void __launch_bounds__(64) linearize_wave_kernel(KArgs<T> a) {
    using Dyn = Linear<T, NX, NU>;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    constexpr int oFU = NX * NX, oLX = oFU + NX * NU, oLU = oLX + NX, oLXX = oLU + NU, oLUX = oLXX + NX * NX,
                  oLUU = oLUX + NU * NX;
    const size_t B = a.B;
    const size_t wid = blockIdx.x;
    const int t = (int)(wid / B);
    const int b = (int)(wid % B);
    if (t > a.N || !traj_active(a.status[b])) return;
    const int lane = threadIdx.x;
    const int slot = a.cur_slot[b];
    const T* __restrict__ p = a.params;
    __shared__ T dx[NX], uu[NU];
    if (lane < NX) dx[lane] = a.X[(((size_t)slot * (a.N + 1) + t) * NX + lane) * B + b] - p[PL::XT + lane];
    if (lane < NU && t < a.N) uu[lane] = a.U[(((size_t)slot * a.N + t) * NU + lane) * B + b];
    __syncthreads();
    if (t == a.N) {
        for (int e = lane; e < NX + NX * NX; e += 64) {
            T v;
            if (e < NX) {
                v = T(0);
                for (int j = 0; j < NX; ++j) v += p[PL::QFS + e * NX + j] * dx[j];
            } else {
                v = p[PL::QFS + (e - NX)];
            }
            a.term[(size_t)e * B + b] = v;
        }
        return;
    }
    const bool euler = a.integ != ILQR_INT_DISCRETE;
    T* out = a.lin + ((size_t)t * B + b) * E;
    for (int e = lane; e < E; e += 64) {
        T v;
        if (e < oFU) {
            const int i = e / NX, j = e % NX;
            v = euler ? T(i == j) + a.dt * p[e] : p[e];
        } else if (e < oLX) {
            v = euler ? a.dt * p[e] : p[e];
        } else if (e < oLU) {
            const int i = e - oLX;
            T acc = T(0);
            for (int j = 0; j < NX; ++j) acc += p[PL::QS + i * NX + j] * dx[j];
            v = acc * a.dt;
        } else if (e < oLXX) {
            const int i = e - oLU;
            T acc = T(0);
            for (int j = 0; j < NU; ++j) acc += p[PL::RS + i * NU + j] * uu[j];
            v = acc * a.dt;
        } else if (e < oLUX) {
            v = p[PL::QS + (e - oLXX)] * a.dt;
        } else if (e < oLUU) {
            v = T(0);
        } else {
            v = p[PL::RS + (e - oLUU)] * a.dt;
        }
        out[e] = v;
    }
}

// ---------------------------------------------------------------------------
// backward sweep: one wave per trajectory, tiles in LDS.
// ---------------------------------------------------------------------------
template <typename T, int NX, int NU>
__global__ void __launch_bounds__(64) backward_wave_kernel(KArgs<T> a) {
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    constexpr int oFU = NX * NX, oLX = oFU + NX * NU, oLU = oLX + NX, oLXX = oLU + NU, oLUX = oLXX + NX * NX,
                  oLUU = oLUX + NU * NX;
    constexpr int R = gain_record(NX, NU);
    constexpr int PER = (E + 63) / 64;
    constexpr int NRHS = NX + 1;
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int st = a.status[b];
    if (!traj_active(st)) return;
    const size_t B = a.B;
    const int N = a.N;

    __shared__ T tile[E];               // f_x f_u l_x l_u l_xx l_ux l_uu of the current step
    __shared__ T V[NX * NX], Vx[NX];    // carried value function
    __shared__ T P[NX * NX], Pu[NU * NX];
    __shared__ T Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU], Qx[NX], Qu[NU];
    __shared__ T Lc[NU * NU], Z[NU * NRHS];  // Cholesky factor; solutions [K | k] (negated on use)
    __shared__ int s_pd;

    for (int e = lane; e < NX * NX; e += 64) V[e] = a.term[(size_t)(NX + e) * B + b];
    if (lane < NX) Vx[lane] = a.term[(size_t)lane * B + b];
    const T* __restrict__ lin = a.lin + (size_t)b * E;
    const size_t tstride = B * E;
    T pre[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int e = lane + 64 * r;
        pre[r] = e < E ? lin[(size_t)(N - 1) * tstride + e] : T(0);
    }
    bool all_pd = true;
    for (int t = N - 1; t >= 0; --t) {
        // ---- phase 0: publish the prefetched record, request the next one -------------------
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = lane + 64 * r;
            if (e < E) tile[e] = pre[r];
        }
        const int tn = t > 0 ? t - 1 : 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = lane + 64 * r;
            pre[r] = e < E ? lin[(size_t)tn * tstride + e] : T(0);
        }
        __syncthreads();
        // ---- phase 1: P = f_x' V ; Pu = f_u' V ; Q_x ; Q_u ------------------------------------
        for (int e = lane; e < NX * NX; e += 64) {
            const int i = e / NX, j = e % NX;
            T acc = T(0);
#pragma unroll 4
            for (int s = 0; s < NX; ++s) acc += tile[s * NX + i] * V[s * NX + j];
            P[e] = acc;
        }
        for (int e = lane; e < NU * NX; e += 64) {
            const int i = e / NX, j = e % NX;
            T acc = T(0);
#pragma unroll 4
            for (int s = 0; s < NX; ++s) acc += tile[oFU + s * NU + i] * V[s * NX + j];
            Pu[e] = acc;
        }
        if (lane < NX) {
            T acc = T(0);
            for (int s = 0; s < NX; ++s) acc += tile[s * NX + lane] * Vx[s];
            Qx[lane] = tile[oLX + lane] + acc;
        } else if (lane >= 32 && lane < 32 + NU) {
            const int j = lane - 32;
            T acc = T(0);
            for (int s = 0; s < NX; ++s) acc += tile[oFU + s * NU + j] * Vx[s];
            Qu[j] = tile[oLU + j] + acc;
        }
        __syncthreads();
        // ---- phase 2: Q_xx = l_xx + P f_x ; Q_ux = l_ux + Pu f_x ; Q_uu = l_uu + Pu f_u ---------
        for (int e = lane; e < NX * NX; e += 64) {
            const int i = e / NX, j = e % NX;
            T acc = T(0);
#pragma unroll 4
            for (int s = 0; s < NX; ++s) acc += P[i * NX + s] * tile[s * NX + j];
            Qxx[e] = tile[oLXX + e] + acc;
        }
        for (int e = lane; e < NU * NX; e += 64) {
            const int i = e / NX, j = e % NX;
            T acc = T(0);
#pragma unroll 4
            for (int s = 0; s < NX; ++s) acc += Pu[i * NX + s] * tile[s * NX + j];
            Qux[e] = tile[oLUX + e] + acc;
        }
        for (int e = lane; e < NU * NU; e += 64) {
            const int i = e / NU, j = e % NU;
            T acc = T(0);
#pragma unroll 4
            for (int s = 0; s < NX; ++s) acc += Pu[i * NX + s] * tile[oFU + s * NU + j];
            Quu[e] = tile[oLUU + e] + acc;
        }
        __syncthreads();
        // ---- phase 3: Cholesky of Q_uu + mu I (column by column, rows in parallel) ---------------
        if (lane == 0) s_pd = 1;
        for (int c = 0; c < NU; ++c) {
            __syncthreads();
            if (lane == 0) {
                T d = Quu[c * NU + c] + a.mu;
                for (int s = 0; s < c; ++s) d -= Lc[c * NU + s] * Lc[c * NU + s];
                if (!(d > T(0))) s_pd = 0;
                Lc[c * NU + c] = M<T>::sqrt(d);
            }
            __syncthreads();
            if (lane > c && lane < NU) {
                T v = Quu[lane * NU + c];
                for (int s = 0; s < c; ++s) v -= Lc[lane * NU + s] * Lc[c * NU + s];
                Lc[lane * NU + c] = v / Lc[c * NU + c];
            }
        }
        __syncthreads();
        const bool pd = s_pd != 0;
        all_pd = all_pd && pd;
        // ---- phase 4: solve for [K | k]: one right-hand side per lane ----------------------------
        if (pd) {
            if (lane < NRHS) {
                T y[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    T v = lane < NX ? Qux[i * NX + lane] : Qu[i];
#pragma unroll
                    for (int s = 0; s < i; ++s) v -= Lc[i * NU + s] * y[s];
                    y[i] = v / Lc[i * NU + i];
                }
#pragma unroll
                for (int i = NU - 1; i >= 0; --i) {
                    T v = y[i];
#pragma unroll
                    for (int s = i + 1; s < NU; ++s) v -= Lc[s * NU + i] * y[s];
                    y[i] = v / Lc[i * NU + i];
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) Z[i * NRHS + lane] = -y[i];
            }
        } else if (lane == 0) {
            // not positive definite: Gaussian elimination with partial pivoting, what the reference's
            // jnp.linalg.solve always does (iLQR_class.py:109-110); serial, rare
            for (int i = 0; i < NU; ++i) {
                for (int j = 0; j < NU; ++j) Lc[i * NU + j] = Quu[i * NU + j] + (i == j ? a.mu : T(0));
                for (int j = 0; j < NX; ++j) Z[i * NRHS + j] = Qux[i * NX + j];
                Z[i * NRHS + NX] = Qu[i];
            }
            for (int k = 0; k < NU; ++k) {
                int piv = k;
                T best = M<T>::abs(Lc[k * NU + k]);
                for (int i = k + 1; i < NU; ++i)
                    if (M<T>::abs(Lc[i * NU + k]) > best) { best = M<T>::abs(Lc[i * NU + k]); piv = i; }
                if (piv != k) {
                    for (int j = 0; j < NU; ++j) { T w = Lc[k * NU + j]; Lc[k * NU + j] = Lc[piv * NU + j]; Lc[piv * NU + j] = w; }
                    for (int j = 0; j < NRHS; ++j) { T w = Z[k * NRHS + j]; Z[k * NRHS + j] = Z[piv * NRHS + j]; Z[piv * NRHS + j] = w; }
                }
                for (int i = k + 1; i < NU; ++i) {
                    const T l = Lc[i * NU + k] / Lc[k * NU + k];
                    for (int j = k + 1; j < NU; ++j) Lc[i * NU + j] -= l * Lc[k * NU + j];
                    for (int j = 0; j < NRHS; ++j) Z[i * NRHS + j] -= l * Z[k * NRHS + j];
                }
            }
            for (int k = NU - 1; k >= 0; --k)
                for (int j = 0; j < NRHS; ++j) {
                    T acc = Z[k * NRHS + j];
                    for (int i = k + 1; i < NU; ++i) acc -= Lc[k * NU + i] * Z[i * NRHS + j];
                    Z[k * NRHS + j] = acc / Lc[k * NU + k];
                }
            for (int i = 0; i < NU * NRHS; ++i) Z[i] = -Z[i];
        }
        __syncthreads();
        // ---- phase 5: gains out, value update (iLQR_class.py:113-114; full form when mu > 0) ---------
        T* rec = a.gains + ((size_t)t * B + b) * R;
        for (int e = lane; e < NU * NX; e += 64) rec[e] = Z[(e / NX) * NRHS + (e % NX)];
        if (lane < NU) rec[NU * NX + lane] = Z[lane * NRHS + NX];
        constexpr int VPER = (NX * NX + 63) / 64;
        T vnew[VPER];
#pragma unroll
        for (int r = 0; r < VPER; ++r) {
            const int e = lane + 64 * r;
            if (e >= NX * NX) break;
            const int i = e / NX, j = e % NX;
            T acc = T(0);
            if (a.mu == T(0)) {
                for (int s = 0; s < NU; ++s) acc += Qux[s * NX + i] * Z[s * NRHS + j];
            } else {
                for (int s = 0; s < NU; ++s) {
                    T qk = T(0);  // (Q_uu K)[s][j]
                    for (int q = 0; q < NU; ++q) qk += Quu[s * NU + q] * Z[q * NRHS + j];
                    acc += Z[s * NRHS + i] * (qk + Qux[s * NX + j]) + Qux[s * NX + i] * Z[s * NRHS + j];
                }
            }
            vnew[r] = Qxx[e] + acc;
        }
        T vxnew = T(0);
        if (lane < NX) {
            T acc = T(0);
            if (a.mu == T(0)) {
                for (int s = 0; s < NU; ++s) acc += Z[s * NRHS + lane] * Qu[s];
            } else {
                for (int s = 0; s < NU; ++s) {
                    T qk = T(0);  // (Q_uu k)[s]
                    for (int q = 0; q < NU; ++q) qk += Quu[s * NU + q] * Z[q * NRHS + NX];
                    acc += Z[s * NRHS + lane] * (qk + Qu[s]) + Qux[s * NX + lane] * Z[s * NRHS + NX];
                }
            }
            vxnew = Qx[lane] + acc;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < VPER; ++r)
            if (lane + 64 * r < NX * NX) V[lane + 64 * r] = vnew[r];
        if (lane < NX) Vx[lane] = vxnew;
        __syncthreads();
    }
    if (lane == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

}  // namespace ilqr

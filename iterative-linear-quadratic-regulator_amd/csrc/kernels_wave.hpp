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
    const int t = a.t_first + (int)(wid / B);     // (t_first > 0: the sparse form's launch over t = N-1, N only)
    const int b = (int)(wid % B);
    if (t > a.N || !traj_active(a.status[b])) return;
    const int lane = threadIdx.x;
    const int slot = a.cur_slot[b];
    const T* __restrict__ p = a.params;
    __shared__ T dx[NX], uu[NU];
    if (lane < NX) dx[lane] = a.X[vec_at(B, a.N + 1, NX, slot, t, b) + lane] - p[PL::XT + lane];
    if (lane < NU && t < a.N) uu[lane] = a.U[vec_at(B, a.N, NU, slot, t, b) + lane];
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
    // The matrices of a Linear system's expansion are the same at every (t, b): when the sweep that follows is the
    // constant-matrix form (backward_mfma16_kernel<T, true>) it reads them from the record of t = N-1 only, and writing
    // them 500 times over was 97 % of this kernel's stores (KArgs::lin_sparse; ilqr_get(ILQR_LIN) re-runs the full form).
    const bool gradients_only = a.lin_sparse && t != a.N - 1;
    for (int e = lane; e < E; e += 64) {
        if (gradients_only && (e < oLX || e >= oLXX)) continue;
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
// The sparse form's gradients (KArgs::lin_sparse): l_x, l_u of every (t, b) as one DENSE side tensor [N][B][n_x + n_u]
// at the front of the expansion buffer (which the sparse form otherwise leaves unused: the constant-matrix sweep takes
// its matrices from the records of t = N-1 at the buffer's end).  One lane per point.  Round 2 wrote them into their
// slots of the 3424-byte records with one 64-thread workgroup per point: 513 k workgroups (240 us at B = 1024), and the
// sweep that followed took 1070 us instead of 630 -- reading 96 fresh bytes out of every 3424-byte record is what was
// slow (tools/c5_anomaly.py: the sweep is only slow right behind that linearisation, not behind the rollout).  Same
// sums in the same order as linearize_wave_kernel: bit-identical values.
// ---------------------------------------------------------------------------
// Round 3: the 320 entries of Qs and Rs are staged in LDS once per 256-thread workgroup and read back as broadcast 16-byte
// reads.  As kernel-argument-indexed constants they had been scalar loads in twenty dependent chunks (the SGPR file holds
// ~100): one wave took 11 us for ~450 instructions whatever the batch.
template <typename T, int NX, int NU>
__global__ void __launch_bounds__(256) linearize_grad_dense_kernel(KArgs<T> a) {
    using Dyn = Linear<T, NX, NU>;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    __shared__ __attribute__((aligned(16))) T sq[NX * NX + NU * NU];
    const T* __restrict__ p = a.params;
    for (int k = threadIdx.x; k < NX * NX + NU * NU; k += 256) sq[k] = k < NX * NX ? p[PL::QS + k] : p[PL::RS + k - NX * NX];
    __syncthreads();
    const size_t B = a.B;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int t = (int)(idx / B);
    const int b = (int)(idx % B);
    if (t >= a.N || !traj_active(a.status[b])) return;
    const int slot = a.cur_slot[b];
    T x[NX], u[NU];
    vec_load<T, NX>(a.X + vec_at(B, a.N + 1, NX, slot, t, b), x);
    vec_load<T, NU>(a.U + vec_at(B, a.N, NU, slot, t, b), u);
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] -= p[PL::XT + j];
    T g[NX + NU];
    const T dt = a.dt;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < NX; ++j) acc += sq[i * NX + j] * x[j];
        g[i] = acc * dt;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < NU; ++j) acc += sq[NX * NX + i * NU + j] * u[j];
        g[NX + i] = acc * dt;
    }
    vec_store<T, NX + NU>(a.lin + ((size_t)t * B + b) * (NX + NU), g);
}

// ---------------------------------------------------------------------------
// forward rollout for linear systems, n_x > 4: ONE WAVE per (trajectory, alpha) candidate.
// The system is time-invariant, so every lane keeps its share of A (or I + dt A), B (or dt B), Q, R, Q_f
// in registers for the whole rollout -- the lane-per-rollout kernel re-read the 704-entry parameter block
// at every step (11.8 us/step at n = 16).  Lane (i, p) = (row i of A, part p of its columns); the state and
// the control of the current step live in LDS and every lane reads the pieces it needs; dot products are
// finished by xor-shuffles over the lanes of a row.  Replaces iLQR._forward_pass_scan
// (iLQR_class.py:193-247) exactly like forward_kernel.
// ---------------------------------------------------------------------------
template <typename T> ILQR_DEV T shfl_xor_t(T v, int mask) { return __shfl_xor(v, mask, 64); }

template <typename T, int NX, int NU>
__global__ void __launch_bounds__(64) forward_wave_kernel(KArgs<T> a) {
    using Dyn = Linear<T, NX, NU>;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    constexpr int R = gain_record(NX, NU);
    constexpr int PR = 64 / NX;                 // lanes per row of A
    constexpr int CA = NX / PR;                 // A / Q columns per lane
    constexpr int CB = (NU + PR - 1) / PR;      // B columns per lane (masked beyond NU)
    constexpr int PK = 64 / NU;                 // lanes per row of K
    constexpr int CK = (NX + PK - 1) / PK;      // K columns per lane (masked beyond NX)
    static_assert(64 % NX == 0 && 64 % NU == 0 && NX % PR == 0, "lane mapping needs NX, NU dividing 64");
    const int b = blockIdx.x, ai = blockIdx.y;
    const int lane = threadIdx.x;
    if (a.init_mode && b == 0 && ai == 0 && lane == 0) a.counters[a.counter_idx] = 0;   // the select that follows counts into it
    if (!a.init_mode && (!traj_active(a.status[b]) || a.accepted[b])) return;
    const size_t B = a.B;
    const int N = a.N;
    const int slot = a.cur_slot[b];
    const int cslot = (slot + 1 + ai) % a.n_slots;
    const T alpha = a.alphas[ai];
    const T* __restrict__ p = a.params;
    const bool discrete = a.integ == ILQR_INT_DISCRETE;
    const int i = lane / PR, pa = lane % PR;    // A mapping
    const int j = lane / PK, pk = lane % PK;    // K mapping
    T Aco[CA], Qco[CA], Qfco[CA], xtq[CA], Bco[CB];
#pragma unroll
    for (int q = 0; q < CA; ++q) {
        const int c = pa * CA + q;
        const T av = p[i * NX + c];
        Aco[q] = discrete ? av : T(i == c) + a.dt * av;
        Qco[q] = p[PL::Q + i * NX + c];
        Qfco[q] = p[PL::QF + i * NX + c];
        xtq[q] = p[PL::XT + c];
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
        const int c = pa * CB + q;
        const T bv = c < NU ? p[NX * NX + i * NU + c] : T(0);
        Bco[q] = discrete ? bv : a.dt * bv;
    }
    const T xti = p[PL::XT + i];
    const int jr = lane / NU, cr = lane % NU;   // R mapping (lanes < NU*NU)
    const T Rco = lane < NU * NU ? p[PL::R + jr * NU + cr] : T(0);

    __shared__ T sx[NX], su[NU];
    if (lane < NX) sx[lane] = a.x0[(size_t)lane * B + b];
    __syncthreads();
    const T* Xo = a.X + vec_at(B, N + 1, NX, slot, 0, b);
    const T* Uo = a.U + vec_at(B, N, NU, slot, 0, b);
    const T* G = a.gains + (size_t)b * R;
    T* Xc = a.X + vec_at(B, N + 1, NX, cslot, 0, b);
    T* Uc = a.U + vec_at(B, N, NU, cslot, 0, b);
    const size_t sX = B * NX, sU = B * NU;   // one time step further
    T cx = T(0), cu = T(0);
    for (int t = 0; t < N; ++t) {
        // ---- u = u_old + alpha k + K (x - x_old)   (iLQR_class.py:181-182) -------------------------
        T part = T(0);
#pragma unroll
        for (int e = 0; e < CK; ++e) {
            const int c = pk * CK + e;
            if (c < NX) part += G[(size_t)t * B * R + j * NX + c] * (sx[c] - Xo[t * sX + c]);
        }
#pragma unroll
        for (int m = 1; m < PK; m <<= 1) part += shfl_xor_t(part, m);
        const T uj = Uo[t * sU + j] + alpha * G[(size_t)t * B * R + NU * NX + j] + part;
        if (pk == 0) {
            su[j] = uj;
            Uc[t * sU + j] = uj;
        }
        __syncthreads();
        // ---- x+ = A x + B u ; stage cost partials ----------------------------------------------------
        T acc = T(0), qpart = T(0);
#pragma unroll
        for (int q = 0; q < CA; ++q) {
            const T xv = sx[pa * CA + q];
            acc += Aco[q] * xv;
            qpart += Qco[q] * (xv - xtq[q]);
        }
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            const int c = pa * CB + q;
            if (c < NU) acc += Bco[q] * su[c];
        }
        const T xi = sx[i];
        cx += (xi - xti) * qpart;
        if (lane < NU * NU) cu += Rco * su[jr] * su[cr];
#pragma unroll
        for (int m = 1; m < PR; m <<= 1) acc += shfl_xor_t(acc, m);
        if (pa == 0) Xc[t * sX + i] = xi;   // the state USED at this step (:188)
        __syncthreads();
        if (pa == 0) sx[i] = acc;
        __syncthreads();
    }
    // terminal state and cost l_f = 0.5 dx' Q_f dx
    T qf = T(0);
#pragma unroll
    for (int q = 0; q < CA; ++q) qf += Qfco[q] * (sx[pa * CA + q] - xtq[q]);
    const T xi = sx[i];
    if (pa == 0) Xc[N * sX + i] = xi;
    T total = (T(0.5) * cx + T(0.5) * cu) * a.dt + T(0.5) * (xi - xti) * qf;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) total += shfl_xor_t(total, m);
    if (lane == 0) a.costs[(size_t)ai * B + b] = total;
}

// ---------------------------------------------------------------------------
// backward sweep: one wave per trajectory, tiles in LDS.
// ---------------------------------------------------------------------------
// out[r][c0 .. c0+W) (+)= sum_s X(r, s) * Y[s][c0 .. c0+W), operands in LDS, one strip of W consecutive
// outputs per lane (W = R*C/64, or 1 with the upper lanes idle when R*C < 64).  XT: the left operand is
// stored transposed (X(r, s) = Xm[s*R + r]), which is how f_x' V, f_u' V and Q_ux' K read it.
template <typename T, int R, int K, int C, bool XT, typename Init>
ILQR_DEV void lds_matmul(const T* Xm, const T* Ym, T* out, int lane, Init init) {
    constexpr int TOT = R * C;
    constexpr int W = TOT >= 64 ? TOT / 64 : 1;
    static_assert(C % W == 0, "a strip must not cross a row");
    const int o = lane * W;
    if (o < TOT) {
        const int r = o / C, c0 = o % C;
        T acc[W];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w] = init(r, c0 + w);
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const T xv = XT ? Xm[s * R + r] : Xm[r * K + s];
#pragma unroll
            for (int w = 0; w < W; ++w) acc[w] += xv * Ym[s * C + c0 + w];
        }
#pragma unroll
        for (int w = 0; w < W; ++w) out[o + w] = acc[w];
    }
}

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

    __shared__ __attribute__((aligned(16))) T tile[E + 4];  // f_x f_u l_x l_u l_xx l_ux l_uu of the current step
    __shared__ __attribute__((aligned(16))) T V[NX * NX], Vx[NX];
    __shared__ __attribute__((aligned(16))) T P[NX * NX], Pu[NU * NX];
    __shared__ __attribute__((aligned(16))) T Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU], Qx[NX], Qu[NU];
    __shared__ __attribute__((aligned(16))) T Lc[NU * NU], Z[NU * NRHS];  // LU fallback scratch; [K | k]

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
    const T* fx = tile;
    const T* fu = tile + oFU;
    for (int t = N - 1; t >= 0; --t) {
        // ---- phase 0: publish the prefetched record, request the next one ---------------------------
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
        // ---- phase 1: P = f_x' V ; Pu = f_u' V ; Q_x = l_x + f_x' V_x ; Q_u = l_u + f_u' V_x --------------
        lds_matmul<T, NX, NX, NX, true>(fx, V, P, lane, [](int, int) { return T(0); });
        lds_matmul<T, NU, NX, NX, true>(fu, V, Pu, lane, [](int, int) { return T(0); });
        if (lane < NX) {
            T acc = tile[oLX + lane];
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += fx[s * NX + lane] * Vx[s];
            Qx[lane] = acc;
        } else if (lane >= 32 && lane < 32 + NU) {
            const int j = lane - 32;
            T acc = tile[oLU + j];
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += fu[s * NU + j] * Vx[s];
            Qu[j] = acc;
        }
        __syncthreads();
        // ---- phase 2: Q_xx = l_xx + P f_x ; Q_ux = l_ux + Pu f_x ; Q_uu = l_uu + Pu f_u -------------------
        lds_matmul<T, NX, NX, NX, false>(P, fx, Qxx, lane, [&](int r, int c) { return tile[oLXX + r * NX + c]; });
        lds_matmul<T, NU, NX, NX, false>(Pu, fx, Qux, lane, [&](int r, int c) { return tile[oLUX + r * NX + c]; });
        lds_matmul<T, NU, NX, NU, false>(Pu, fu, Quu, lane, [&](int r, int c) { return tile[oLUU + r * NU + c]; });
        __syncthreads();
        // ---- phase 3: Cholesky of Q_uu + mu I, redundantly in every lane's registers (no barriers) -----------
        T Lr[NU][NU], Li[NU];   // Li[c] = 1 / L[c][c]: the substitutions multiply by it (one division per column
                                // instead of 2 * NU per right-hand side on the step's serial chain)
        bool pd = true;
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            T d = Quu[c * NU + c] + a.mu;
#pragma unroll
            for (int s2 = 0; s2 < c; ++s2) d -= Lr[c][s2] * Lr[c][s2];
            pd = pd && (d > T(0));
            const T inv = fast_rsqrt(pd ? d : T(1));   // (not positive definite: the LU branch below takes over)
            const T lcc = d * inv;
            Lr[c][c] = lcc;
            Li[c] = inv;
#pragma unroll
            for (int i2 = c + 1; i2 < NU; ++i2) {
                T v = Quu[i2 * NU + c];
#pragma unroll
                for (int s2 = 0; s2 < c; ++s2) v -= Lr[i2][s2] * Lr[c][s2];
                Lr[i2][c] = v * inv;
            }
        }
        all_pd = all_pd && pd;
        // ---- phase 4: [K | k] = -(Q_uu + mu I)^-1 [Q_ux | Q_u], one right-hand side per lane ------------------
        if (pd) {
            if (lane < NRHS) {
                T y[NU];
#pragma unroll
                for (int i2 = 0; i2 < NU; ++i2) {
                    T v = lane < NX ? Qux[i2 * NX + lane] : Qu[i2];
#pragma unroll
                    for (int s2 = 0; s2 < i2; ++s2) v -= Lr[i2][s2] * y[s2];
                    y[i2] = v * Li[i2];
                }
#pragma unroll
                for (int i2 = NU - 1; i2 >= 0; --i2) {
                    T v = y[i2];
#pragma unroll
                    for (int s2 = i2 + 1; s2 < NU; ++s2) v -= Lr[s2][i2] * y[s2];
                    y[i2] = v * Li[i2];
                }
#pragma unroll
                for (int i2 = 0; i2 < NU; ++i2) Z[i2 * NRHS + lane] = -y[i2];
            }
        } else if (lane == 0) {
            // not positive definite: Gaussian elimination with partial pivoting, what the reference's
            // jnp.linalg.solve always does (iLQR_class.py:109-110); serial, rare
            for (int i2 = 0; i2 < NU; ++i2) {
                for (int j2 = 0; j2 < NU; ++j2) Lc[i2 * NU + j2] = Quu[i2 * NU + j2] + (i2 == j2 ? a.mu : T(0));
                for (int j2 = 0; j2 < NX; ++j2) Z[i2 * NRHS + j2] = Qux[i2 * NX + j2];
                Z[i2 * NRHS + NX] = Qu[i2];
            }
            for (int k = 0; k < NU; ++k) {
                int piv = k;
                T best = M<T>::abs(Lc[k * NU + k]);
                for (int i2 = k + 1; i2 < NU; ++i2)
                    if (M<T>::abs(Lc[i2 * NU + k]) > best) { best = M<T>::abs(Lc[i2 * NU + k]); piv = i2; }
                if (piv != k) {
                    for (int j2 = 0; j2 < NU; ++j2) { T w = Lc[k * NU + j2]; Lc[k * NU + j2] = Lc[piv * NU + j2]; Lc[piv * NU + j2] = w; }
                    for (int j2 = 0; j2 < NRHS; ++j2) { T w = Z[k * NRHS + j2]; Z[k * NRHS + j2] = Z[piv * NRHS + j2]; Z[piv * NRHS + j2] = w; }
                }
                for (int i2 = k + 1; i2 < NU; ++i2) {
                    const T l = Lc[i2 * NU + k] / Lc[k * NU + k];
                    for (int j2 = k + 1; j2 < NU; ++j2) Lc[i2 * NU + j2] -= l * Lc[k * NU + j2];
                    for (int j2 = 0; j2 < NRHS; ++j2) Z[i2 * NRHS + j2] -= l * Z[k * NRHS + j2];
                }
            }
            for (int k = NU - 1; k >= 0; --k)
                for (int j2 = 0; j2 < NRHS; ++j2) {
                    T acc = Z[k * NRHS + j2];
                    for (int i2 = k + 1; i2 < NU; ++i2) acc -= Lc[k * NU + i2] * Z[i2 * NRHS + j2];
                    Z[k * NRHS + j2] = acc / Lc[k * NU + k];
                }
            for (int i2 = 0; i2 < NU * NRHS; ++i2) Z[i2] = -Z[i2];
        }
        __syncthreads();
        // ---- phase 5: gains out, value update (iLQR_class.py:113-114; full form when mu > 0) -------------------
        T* rec = a.gains + ((size_t)t * B + b) * R;
        for (int e = lane; e < NU * NX; e += 64) rec[e] = Z[(e / NX) * NRHS + (e % NX)];
        if (lane < NU) rec[NU * NX + lane] = Z[lane * NRHS + NX];
        constexpr int VW = (NX * NX) / 64 > 0 ? (NX * NX) / 64 : 1;   // strip of V per lane
        T vnew[VW];
        const int vo = lane * VW;
        if (vo < NX * NX) {
            const int i2 = vo / NX, j0 = vo % NX;
#pragma unroll
            for (int w = 0; w < VW; ++w) {
                const int j2 = j0 + w;
                T acc = T(0);
                if (a.mu == T(0)) {
#pragma unroll
                    for (int s2 = 0; s2 < NU; ++s2) acc += Qux[s2 * NX + i2] * Z[s2 * NRHS + j2];
                } else {
                    for (int s2 = 0; s2 < NU; ++s2) {
                        T qk = T(0);  // (Q_uu K)[s][j]
                        for (int q = 0; q < NU; ++q) qk += Quu[s2 * NU + q] * Z[q * NRHS + j2];
                        acc += Z[s2 * NRHS + i2] * (qk + Qux[s2 * NX + j2]) + Qux[s2 * NX + i2] * Z[s2 * NRHS + j2];
                    }
                }
                vnew[w] = Qxx[vo + w] + acc;
            }
        }
        T vxnew = T(0);
        if (lane < NX) {
            T acc = T(0);
            if (a.mu == T(0)) {
#pragma unroll
                for (int s2 = 0; s2 < NU; ++s2) acc += Z[s2 * NRHS + lane] * Qu[s2];
            } else {
                for (int s2 = 0; s2 < NU; ++s2) {
                    T qk = T(0);  // (Q_uu k)[s]
                    for (int q = 0; q < NU; ++q) qk += Quu[s2 * NU + q] * Z[q * NRHS + NX];
                    acc += Z[s2 * NRHS + lane] * (qk + Qu[s2]) + Qux[s2 * NX + lane] * Z[s2 * NRHS + NX];
                }
            }
            vxnew = Qx[lane] + acc;
        }
        __syncthreads();
        if (vo < NX * NX) {
#pragma unroll
            for (int w = 0; w < VW; ++w) V[vo + w] = vnew[w];
        }
        if (lane < NX) Vx[lane] = vxnew;
        __syncthreads();
    }
    if (lane == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

}  // namespace ilqr

// forward_mfma16.hpp -- line-searched rollout for n_x = 16, n_u = 8 (BASELINE config c5) on the matrix cores.
//
// The candidates of one line-search pass share everything but alpha: the same nominal trajectory, the same gains, the
// same (time-invariant, linear) dynamics.  Put candidate c in COLUMN c of a 16 x 16 matrix and the rollout of all of
// them (iLQR_class.py:164-247, one lax.scan per trial alpha in the reference) is a matrix recursion,
//     U_t     = u_old 1' + k alpha' + K_t (X_t - x_old 1')            (:181-182)
//     X_{t+1} = A X_t + B U_t                                          (system_base.py:50-74, euler / discrete)
//     cost_c += dt/2 (dx_c' Q dx_c + u_c' R u_c)                       (:333-341)
// whose products are exactly v_mfma_*_16x16x4 shaped.  One wave owns one trajectory and all its candidates (up to 16;
// the reference's 10 fill 10 columns); every matrix lives in the instruction's C/D layout (backward_mfma16.hpp: lane
// 16 g + c holds rows ROW(g, 0..3) of column c), so the left factors M enter as C-layout copies of M' held in registers
// for the whole rollout (A, B, Q, R, Q_f) or loaded per step straight from the gain record (K: rows >= 8 read as 0
// through the descriptor's range check), and nothing is ever moved across lanes until the final cost reduction.
// forward_wave_kernel -- one wave per (trajectory, alpha), state in LDS, three barriers per step -- took 515 us at the
// c5 shard (B = 128, N = 500); see DESIGN.md section 4 for this kernel's figure.
#pragma once
#include "backward_mfma16.hpp"

namespace ilqr {

// M' in C-layout for a row-major M (rows x cols, zero beyond): register r of lane (g, c) = M[c][ROW(g, r)]
template <typename T> ILQR_DEV void load_left_factor(const T* __restrict__ M, int rows, int cols, int g, int c, T scale, T* out) {
    using MF = Mfma16<T>;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = MF::row(g, r);
        out[r] = (c < rows && k < cols) ? scale * M[c * cols + k] : T(0);
    }
}

// fp32 inputs of one step in the register ring: x_old, u_old, k (row-indexed) and K' (C-layout), 16 bytes per lane each
struct Fm16In {
    static constexpr int NLD = 4;
    f32x4n xo, uo, Kt, kk;
#define ILQR_FM16_LOADS(OUT)                                                                         \
    asm volatile("s_nop 4\n\t"                                                                       \
                 "buffer_load_dwordx4 %0, %4, %8, %11 offen\n\t"                                     \
                 "buffer_load_dwordx4 %1, %5, %9, %12 offen\n\t"                                     \
                 "buffer_load_dwordx4 %2, %6, %10, %13 offen\n\t"                                    \
                 "buffer_load_dwordx4 %3, %7, %10, %13 offen"                                         \
                 : OUT(xo), OUT(uo), OUT(Kt), OUT(kk)                                                 \
                 : "v"(vxo), "v"(vuo), "v"(vK), "v"(vk), "s"(sX), "s"(sU), "s"(sG), "s"(ox), "s"(ou), "s"(og) \
                 : "memory")
    // FIRST: nothing to tie to; a refill ties every destination to the consumed slot (see RawTile, backward_tile16.hpp)
    template <bool FIRST>
    ILQR_DEV void issue(const i32x4& sX, const i32x4& sU, const i32x4& sG, int vxo, int vuo, int vK, int vk, int ox, int ou,
                        int og) {
        if constexpr (FIRST) ILQR_FM16_LOADS(ILQR_OUT_FIRST);
        else ILQR_FM16_LOADS(ILQR_OUT_REFILL);
    }
#undef ILQR_FM16_LOADS
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xo), "+v"(uo), "+v"(Kt), "+v"(kk) : "i"(N) : "memory");
    }
};

template <typename T>
__global__ void __launch_bounds__(64) forward_mfma16_kernel(KArgs<T> a) {
    using MF = Mfma16<T>;
    using acc = typename MF::acc;
    constexpr int NX = 16, NU = 8, S = (int)sizeof(T);
    using Dyn = Linear<T, NX, NU>;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    constexpr int R = gain_record(NX, NU);
    const int b = blockIdx.x;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    if (a.init_mode && b == 0 && lane == 0) a.counters[a.counter_idx] = 0;   // the select that follows counts into it
    if (!a.init_mode && (!traj_active(a.status[b]) || a.accepted[b])) return;
    const size_t B = a.B;
    const int N = a.N;
    const int slot = a.cur_slot[b];
    const bool col_live = c < a.n_pass;                     // column c carries candidate c
    const int cslot = (slot + 1 + c) % a.n_slots;
    T alpha = T(0);
#pragma unroll
    for (int k = 0; k < kMaxAlpha; ++k) alpha = (c == k && k < a.n_pass) ? a.alphas[k] : alpha;
    const T* __restrict__ p = a.params;
    const bool discrete = a.integ == ILQR_INT_DISCRETE;
    const acc zero = {T(0), T(0), T(0), T(0)};

    // ---- constant left factors, C-layout copies of M': (I + dt A)' or A', (dt B)' or B', Q', R', Q_f' ------------------
    T At[4], Bt[4], Qt[4], Rt[4], Qft[4], xt[4];
    load_left_factor<T>(p, NX, NX, g, c, discrete ? T(1) : a.dt, At);
    load_left_factor<T>(p + NX * NX, NX, NU, g, c, discrete ? T(1) : a.dt, Bt);
    load_left_factor<T>(p + PL::Q, NX, NX, g, c, T(1), Qt);
    load_left_factor<T>(p + PL::R, NU, NU, g, c, T(1), Rt);
    load_left_factor<T>(p + PL::QF, NX, NX, g, c, T(1), Qft);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = MF::row(g, r);
        if (!discrete) At[r] += T(row == c);               // euler: x+ = (I + dt A) x + dt B u
        xt[r] = p[PL::XT + row];
    }
    // Is the quadratic cost diagonal (every driver of the reference builds Q, R with diag())?  Then dx' Q dx needs no
    // product: four multiply-adds per lane instead of four MFMAs on the matrix pipe.  Wave-uniform.
    bool qdiag = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) qdiag = qdiag && (MF::row(g, r) == c || (Qt[r] == T(0) && Rt[r] == T(0)));
    qdiag = __ballot(!qdiag) == 0ull;
    T qd[4], rd[4];                                         // diagonal entries, row-indexed
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = MF::row(g, r);
        qd[r] = p[PL::Q + row * NX + row];
        rd[r] = row < NU ? p[PL::R + row * NU + row] : T(0);
    }

    // ---- per-lane byte offsets ---------------------------------------------------------------------------------
    constexpr int kBeyond = 0x7ffffff0;
    const int row0 = MF::row(g, 0);
    constexpr int RS = MF::row(0, 1) - MF::row(0, 0);       // row stride between registers: 1 (f32) / 4 (f64)
    const unsigned bytesX = (unsigned)((size_t)a.n_slots * (N + 1) * NX * B * S);
    const unsigned bytesU = (unsigned)((size_t)a.n_slots * N * NU * B * S);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, bytesX), rU = make_rsrc(a.U, bytesU);
    const __amdgpu_buffer_rsrc_t rG = make_rsrc(a.gains, (unsigned)((size_t)N * B * R * S));
    const int stepX = (int)(B * NX * S), stepU = (int)(B * NU * S), stepG = (int)(B * R * S);
    // nominal trajectory, row-indexed (the same in every column): x_old[ROW(g, r)], u_old / k [ROW(g, r)] (rows < 8)
    const int vxo = (int)(vec_at(B, N + 1, NX, slot, 0, b) * S) + S * row0;
    const int vuo = row0 < NU ? (int)(vec_at(B, N, NU, slot, 0, b) * S) + S * row0 : kBeyond;
    // K' in C-layout: K[c][ROW(g, r)], rows c < 8;  k[ROW(g, r)]
    const int vK = c < NU ? (int)((size_t)b * R * S) + S * (NX * c + row0) : kBeyond;
    const int vk = row0 < NU ? (int)((size_t)b * R * S) + S * (NU * NX + row0) : kBeyond;
    // candidate column c: X[cslot][t][b][ROW(g, r)], U[cslot][t][b][ROW(g, r)] (rows < 8)
    const int vXc = col_live ? (int)(vec_at(B, N + 1, NX, cslot, 0, b) * S) + S * row0 : kBeyond;
    const int vUc = (col_live && row0 < NU) ? (int)(vec_at(B, N, NU, cslot, 0, b) * S) + S * row0 : kBeyond;

    // rows ROW(g, 0..3) of a vector of `nrows` entries.  f32: a lane group holds four consecutive rows, and a group
    // beyond the vector carries an out-of-range offset (reads 0, stores dropped); f64: register q holds row g + 4 q, so
    // the registers beyond the vector are skipped.
    auto load4 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff, T* o, int nrows) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = MF::row(0, q) < nrows ? buf_load1(r, voff + S * RS * q, soff, T(0)) : T(0);
    };
    auto store4 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff, const T* v, int nrows) {
        if constexpr (RS == 1) {
            buf_store_vec<T, 4>(r, voff, soff, v);           // f32: rows 4g .. 4g+3 are 16 contiguous bytes
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (MF::row(0, q) < nrows) buf_store1(r, voff + S * RS * q, soff, v[q]);
        }
    };

    T X[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = a.x0[(size_t)MF::row(g, r) * B + b];
    T cx = T(0), cu = T(0);

    // one rollout step for all columns: the inputs of step t are x_old, u_old (row-indexed), K' (C-layout), k
    auto do_step = [&](const T* xo, const T* uo, const T* Kt, const T* kk, int t) {
        // A X on its own accumulator: it does not wait for the control, and the matrix pipe takes it while the vector
        // ALU forms dX; K dX -> U -> B U is the step's dependent chain (an MFMA issues every 32 cycles, dependent or not)
        T DX[4], U[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) DX[r] = X[r] - xo[r];
        acc kd = zero, ax = zero, bu = zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) kd = MF::mma(Kt[r], DX[r], kd);
#pragma unroll
        for (int r = 0; r < 4; ++r) ax = MF::mma(At[r], X[r], ax);
        // U = u_old 1' + k alpha' + K (X - x_old 1')
#pragma unroll
        for (int r = 0; r < 4; ++r) U[r] = uo[r] + alpha * kk[r] + kd[r];     // (rows >= 8: 0 + 0 + 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) bu = MF::mma(Bt[r], U[r], bu);
        // the state USED at this step and its control (:188): exactly two store instructions in fp32 (they are counted)
        store4(rX, vXc, uniform(t * stepX), X, NX);
        store4(rU, vUc, uniform(t * stepU), U, NU);
        // stage cost, per lane over its rows (reduced over the lane groups once, at the end)
        T DT[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) DT[r] = X[r] - xt[r];
        if (qdiag) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cx += DT[r] * (qd[r] * DT[r]);
                cu += U[r] * (rd[r] * U[r]);
            }
        } else {
            acc qx = zero, ru = zero;
#pragma unroll
            for (int r = 0; r < 4; ++r) { qx = MF::mma(Qt[r], DT[r], qx); ru = MF::mma(Rt[r], U[r], ru); }
#pragma unroll
            for (int r = 0; r < 4; ++r) { cx += DT[r] * qx[r]; cu += U[r] * ru[r]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) X[r] = ax[r] + bu[r];   // X+ = A X + B U
    };

    if constexpr (RS == 1) {
        // fp32: the four inputs of a step are four 16-byte loads per lane (rows 4g .. 4g+3 are contiguous), kept PF
        // steps ahead in a register ring by inline-asm buffer loads with self-counted vmcnt -- the technique, and the
        // reasons, of RawTile in backward_tile16.hpp; csrc/verify_ring_isa.py checks this kernel's assembly too.  A
        // step issues NLD = 4 loads and NST = 2 stores.
        using In = Fm16In;
        constexpr int NLD = In::NLD, NST = 2, PF = 8;
        static_assert((PF - 1) * (NLD + NST) <= 63, "vmcnt field");
        const i32x4 sX = make_srd(a.X, bytesX), sU = make_srd(a.U, bytesU), sG = make_srd(a.gains, (unsigned)((size_t)N * B * R * S));
        auto issue = [&](In& in, int t, auto first) {
            in.template issue<decltype(first)::value>(sX, sU, sG, vxo, vuo, vK, vk, uniform(t * stepX), uniform(t * stepU),
                                                      uniform(t * stepG));
        };
        auto step = [&](const In& in, int t) {
            const T xo[4] = {in.xo.x, in.xo.y, in.xo.z, in.xo.w}, uo[4] = {in.uo.x, in.uo.y, in.uo.z, in.uo.w};
            const T Kt[4] = {in.Kt.x, in.Kt.y, in.Kt.z, in.Kt.w}, kk[4] = {in.kk.x, in.kk.y, in.kk.z, in.kk.w};
            do_step(xo, uo, Kt, kk, t);
        };
        constexpr std::true_type kFirst{};
        constexpr std::false_type kRefill{};
        int t = 0;
        // leading remainder: one slot, fully waited (whole rings only in the pipelined loop)
        for (int r = N % PF; r > 0; --r, ++t) {
            In in;
            issue(in, t, kFirst);
            in.template wait<0>();
            step(in, t);
        }
        if (t < N) {
            In ring[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) issue(ring[q], t + q, kFirst);
#pragma unroll
            for (int q = 0; q < PF; ++q) {   // first pass: the prologue loads may be the only operations in flight
                ring[q].template wait<(PF - 1) * NLD>();
                step(ring[q], t + q);
                issue(ring[q], (t + q + PF < N) ? t + q + PF : N - 1, kRefill);
            }
            for (t += PF; t < N; t += PF) {
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    ring[q].template wait<(PF - 1) * (NLD + NST)>();
                    step(ring[q], t + q);
                    issue(ring[q], (t + q + PF < N) ? t + q + PF : N - 1, kRefill);   // clamped: branch-free refill
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        // fp64: register q of a lane holds row g + 4 q, so every input is four 8-byte loads; one step of compiler-
        // counted lookahead
        T xo[4], uo[4], Kt[4], kk[4];
        load4(rX, vxo, 0, xo, NX); load4(rU, vuo, 0, uo, NU); load4(rG, vK, 0, Kt, NX); load4(rG, vk, 0, kk, NU);
        {
            // every load of the prologue is consumed before the loop, so that hipcc's s_waitcnt bookkeeping enters it
            // with nothing pending (otherwise the loop header merges the two states into "wait for the previous step's
            // stores": backward_mfma16.hpp)
            T touch = T(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) touch += xo[r] + uo[r] + Kt[r] + kk[r] + X[r];
            asm volatile("" ::"v"(touch));
        }
        for (int t = 0; t < N; ++t) {
            // the next step's inputs do not depend on the carried state: request them before this step's chain
            T xo_n[4], uo_n[4], Kt_n[4], kk_n[4];
            const int tn = t + 1 < N ? t + 1 : t;
            load4(rX, vxo, uniform(tn * stepX), xo_n, NX); load4(rU, vuo, uniform(tn * stepU), uo_n, NU);
            load4(rG, vK, uniform(tn * stepG), Kt_n, NX); load4(rG, vk, uniform(tn * stepG), kk_n, NU);
            do_step(xo, uo, Kt, kk, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) { xo[r] = xo_n[r]; uo[r] = uo_n[r]; Kt[r] = Kt_n[r]; kk[r] = kk_n[r]; }
        }
    }
    // terminal state and cost l_f = 0.5 dx' Q_f dx (not scaled by dt)
    store4(rX, vXc, uniform(N * stepX), X, NX);
    T DT[4], cf = T(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) DT[r] = X[r] - xt[r];
    acc qf = zero;
#pragma unroll
    for (int r = 0; r < 4; ++r) qf = MF::mma(Qft[r], DT[r], qf);
#pragma unroll
    for (int r = 0; r < 4; ++r) cf += DT[r] * qf[r];
    T total = (T(0.5) * cx + T(0.5) * cu) * a.dt + T(0.5) * cf;
    const int a16 = 4 * (lane ^ 16), a32 = 4 * (lane ^ 32);
    total = MF::group_sum(total, a16, a32);
    if (g == 0 && col_live) a.costs[(size_t)c * B + b] = total;
}

}  // namespace ilqr

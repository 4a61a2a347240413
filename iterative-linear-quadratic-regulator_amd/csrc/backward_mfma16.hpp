// backward_mfma16.hpp -- backward Riccati sweep for n_x = 16, n_u = 8 (BASELINE config c5) on the matrix cores.
//
// One wave owns one trajectory, as in backward_wave_kernel, but the value function never leaves the wave's
// registers and the dense products of iLQR_class.py:100-114 run as v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64
// (exact f32 / f64 FMA chains, so the rounding is that of an ordinary dot product): this is the one place on the
// path where n is large enough for MFMA -- a 16 x 16 x 16 product is exactly four instructions.
//
// Everything is kept in the instruction's C/D layout ("C-layout"): lane l = 16 g + c holds, in register r, element
// (row ROW(g, r), column c) of a 16 x 16 matrix, ROW(g, r) = 4 g + r for f32 and g + 4 r for f64.  An MFMA step r fed
// with register r of two C-layout matrices X, Y as its A and B operands contracts over k = ROW(g, r): the four steps
// together compute X' Y, whatever the order in which they visit k.  Every product of a Riccati step has that form,
// with both factors already in C-layout, so no value ever has to be transposed or moved across lanes:
//     Pt  = V' A                       (= (A'V)' : the reference's left-associated f_x.T @ V_xx)
//     Qxx = l_xx + Pt' A               (= l_xx + (A'V) A)
//     Put = V' B ,  Qux = l_ux + Put' A ,  Quu = l_uu + Put' B
//     V+  = Qxx + Qux' K                           (short form, iLQR_class.py:114)
// The matrix-vector parts (Q_x = l_x + A'V_x, Q_u = l_u + B'V_x, V_x+ = Q_x + K'Q_u) stay on the vector ALU: an
// f32-input MFMA runs at only twice a lone wave's vector rate, and a 16-column product for one useful column would
// put four more 32-cycle instructions on the step's dependent chain.
// The gain solve [K | k] = -Quu^-1 [Qux | Qu] (:109-110) runs on the vector ALU: the 36 entries of Quu's lower
// triangle are read to scalars (v_readlane), every lane factors Quu redundantly (Cholesky; rsqrt + Newton), and
// each lane substitutes one right-hand side -- column c of Qux in lanes 0..31, Q_u in lanes 32..63 -- gathered with
// ds_bpermute (LDS crossbar, no LDS memory).  If Quu is not positive definite the lanes fall back to LU with partial
// pivoting, what the reference's solve does unconditionally.
//
// mu > 0 (the build's Levenberg extension) and n_x = 8 stay on backward_wave_kernel.
#pragma once
#include "kernels.hpp"

namespace ilqr {

typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef double f64x4a __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    using acc = f32x4a;
    static ILQR_DEV constexpr int row(int g, int r) { return 4 * g + r; }
    static ILQR_DEV constexpr int grp_of(int i) { return i / 4; }   // lane group and register that hold row i
    static ILQR_DEV constexpr int reg_of(int i) { return i % 4; }
    static ILQR_DEV acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static ILQR_DEV float readlane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
    static ILQR_DEV float bperm(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }
    // v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- / 32-lane halves on the vector ALU: no LDS-crossbar round
    // trip (~120 cycles each for ds_bpermute, which a lone wave cannot hide).  swap16(v): {lower row's value, upper
    // row's value} of every pair of 16-lane rows, in both rows; swap32(v): the same for the two 32-lane halves.
    static ILQR_DEV void swap16(float v, float& lo, float& hi) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
    }
    static ILQR_DEV void swap32(float v, float& lo, float& hi) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
    }
    // sum over the four lane groups (g = 0..3), result in every lane
    static ILQR_DEV float group_sum(float v, int, int) {
        float a, b;
        swap16(v, a, b);
        v = a + b;
        swap32(v, a, b);
        return a + b;
    }
};
template <> struct Mfma16<double> {
    using acc = f64x4a;
    static ILQR_DEV constexpr int row(int g, int r) { return g + 4 * r; }
    static ILQR_DEV constexpr int grp_of(int i) { return i % 4; }
    static ILQR_DEV constexpr int reg_of(int i) { return i / 4; }
    static ILQR_DEV acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static ILQR_DEV double readlane(double v, int lane) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
    }
    static ILQR_DEV double bperm(int byte_addr, double v) {
        return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)),
                                __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
    }
    static ILQR_DEV double group_sum(double v, int a16, int a32) {   // (f64 keeps the LDS-crossbar form)
        v += bperm(a16, v);
        v += bperm(a32, v);
        return v;
    }
};

// init + X' Y for two C-layout matrices (four MFMA steps, one dependent chain)
template <typename T>
ILQR_DEV typename Mfma16<T>::acc mm_tn(const T* X, const T* Y, typename Mfma16<T>::acc init) {
    typename Mfma16<T>::acc d = init;
#pragma unroll
    for (int r = 0; r < 4; ++r) d = Mfma16<T>::mma(X[r], Y[r], d);
    return d;
}

// one timestep's expansion: the matrices in C-layout (zero where the (16, 8) blocks are padded to 16 x 16), the two
// gradient vectors "column-indexed" (lane (g, c) holds entry c)
template <typename T> struct Tile16x8 {
    T A[4], Bm[4], lxx[4], lux[4], luu[4], lx, lu;
};

// Per-lane byte offsets into one expansion record [f_x | f_u | l_x | l_u | l_xx | l_ux | l_uu] and into one gain record
// [K (8 x 16) | k (8)], fixed for the whole sweep.  ROW(g, r) = ROW(g, 0) + RS r, so register r of a matrix is the
// lane's offset plus an immediate; a lane whose element lies in the zero padding (or that has nothing to store)
// carries an offset beyond the record, where a raw buffer load returns 0 and a raw buffer store is dropped
// (tools/micro/range_probe.hip measured the rule): no selects, no branches, a fixed number of memory operations.
struct Lane16x8 {
    int vA, vB, vLux, vLuu, vLx, vLu, vK, vk;
};
ILQR_DEV float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
ILQR_DEV double buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)a.y, (int)a.x);
}
// 16 bytes: four floats / two doubles
ILQR_DEV void buf_store16(__amdgpu_buffer_rsrc_t r, int voff, const float* v) {
    const u32x4 w = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
}
ILQR_DEV void buf_store16(__amdgpu_buffer_rsrc_t r, int voff, const double* v) {
    const u32x4 w = {(unsigned)__double2loint(v[0]), (unsigned)__double2hiint(v[0]), (unsigned)__double2loint(v[1]),
                     (unsigned)__double2hiint(v[1])};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
}

template <typename T> ILQR_DEV __amdgpu_buffer_rsrc_t tile16x8_rsrc_of(const T* rec) {
    constexpr int NX = 16, NU = 8;
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    return make_rsrc(rec, E * (int)sizeof(T));
}

template <typename T>
ILQR_DEV void tile16x8_load(Tile16x8<T>& t, const T* rec, const Lane16x8& o) {
    using MF = Mfma16<T>;
    constexpr int NX = 16, NU = 8, S = (int)sizeof(T);
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    constexpr int oFU = NX * NX, oLX = oFU + NX * NU, oLU = oLX + NX, oLXX = oLU + NU, oLUX = oLXX + NX * NX,
                  oLUU = oLUX + NU * NX;
    constexpr int RS = MF::row(0, 1) - MF::row(0, 0);            // row stride between registers: 1 (f32) / 4 (f64)
    // the descriptor covers exactly this record; `rec` is wave-uniform (block index and loop counter), which make_rsrc
    // states through readfirstlane -- otherwise hipcc wraps every load in a waterfall loop (cdna_hip_programming.md T20)
    const __amdgpu_buffer_rsrc_t r = make_rsrc(rec, E * S);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool reg_ok = MF::row(0, q) < NU;                    // f64: registers 2, 3 hold rows >= 8
        t.A[q] = buf_load1(r, o.vA + S * NX * RS * q, 0, T(0));
        t.lxx[q] = buf_load1(r, o.vA + S * NX * RS * q, S * oLXX, T(0));
        t.Bm[q] = buf_load1(r, o.vB + S * NU * RS * q, S * oFU, T(0));
        t.lux[q] = reg_ok ? buf_load1(r, o.vLux + S * NX * RS * q, S * oLUX, T(0)) : T(0);
        t.luu[q] = reg_ok ? buf_load1(r, o.vLuu + S * NU * RS * q, S * oLUU, T(0)) : T(0);
    }
    t.lx = buf_load1(r, o.vLx, S * oLX, T(0));
    t.lu = buf_load1(r, o.vLu, S * oLU, T(0));
}

// Diagnostic build only (-DILQR_MFMA16_STAMPS, tools/c5_stamps.py): s_memtime at the phase boundaries of a step, summed
// over the sweep by workgroup 0 and written to the probe buffer (values nothing else reads).  ILQR_STAMP(k, v) issues a
// v_mov of `v` first, so the stamp is taken once `v` is available (a lone wave issues in order).
#ifdef ILQR_MFMA16_STAMPS
#define ILQR_STAMP(k, v)                                                                  \
    do {                                                                                  \
        float sink_;                                                                      \
        asm volatile("v_mov_b32 %0, %1" : "=v"(sink_) : "v"((float)(v)));                 \
        const long long now_ = __builtin_readcyclecounter();                              \
        stamp_acc[k] += now_ - stamp_last;                                                \
        stamp_last = now_;                                                                \
    } while (0)
#else
#define ILQR_STAMP(k, v) do {} while (0)
#endif

// CONST: the matrices of the expansion (f_x, f_u, l_xx, l_ux, l_uu) do not depend on (t, b) -- a Linear system with the
// parameter-block quadratic cost, which is what the library's own linearisation of n = 16 always is -- so they are
// loaded once and a step fetches l_x and l_u only (2 loads instead of 22).  Caller-supplied tensors (ilqr_backward_tensors)
// take the general form.
template <typename T, bool CONST>
__global__ void __launch_bounds__(64) backward_mfma16_kernel(KArgs<T> a) {
    using MF = Mfma16<T>;
    using acc = typename MF::acc;
    constexpr int NX = 16, NU = 8, S = (int)sizeof(T);
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    constexpr int R = gain_record(NX, NU);
    constexpr int RS = MF::row(0, 1) - MF::row(0, 0);
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int g = lane >> 4, c = lane & 15;
    const int st = a.status[b];
    if (!traj_active(st)) return;
    const size_t B = a.B;
    const int N = a.N;

    // terminal condition (iLQR_class.py:136-138): V in C-layout; V_x "row-indexed" (register r of lane (g, .) holds
    // entry ROW(g, r): the B operand of a product with a matrix of 16 equal columns)
    T V[4], Vx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = MF::row(g, r);
        V[r] = a.term[(size_t)(NX + row * NX + c) * B + b];
        Vx[r] = a.term[(size_t)row * B + b];
    }
    const T* __restrict__ lin = a.lin + (size_t)b * E;
    const size_t tstride = B * E;
    Lane16x8 off;
    constexpr int kBeyond = 0x7ffffff0;
    const int row0 = MF::row(g, 0);
    const bool lane_ok = row0 < NU;                                // f32: lane groups 2, 3 only hold rows >= 8
    off.vA = S * (NX * row0 + c);
    off.vB = c < NU ? S * (NU * row0 + c) : kBeyond;
    off.vLux = lane_ok ? S * (NX * row0 + c) : kBeyond;
    off.vLuu = (lane_ok && c < NU) ? S * (NU * row0 + c) : kBeyond;
    off.vLx = S * c;
    off.vLu = c < NU ? S * c : kBeyond;
    off.vK = lane_ok ? S * (NX * row0 + c) : kBeyond;            // K[ROW(g, r)][c] (rows < 8)
    off.vk = lane == 32 ? S * NU * NX : kBeyond;                 // k: one lane of the half that solved for it
    // cross-lane addresses (ds_bpermute: byte address = 4 * source lane), fixed for the whole sweep
    const int a16 = 4 * (lane ^ 16), a32 = 4 * (lane ^ 32);
    int aRow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) aRow[r] = 4 * MF::row(g, r);       // lane (0, ROW(g, r)): column-indexed -> row-indexed

    // the constant-matrix form's gradients: behind f_x, f_u in the record, or the dense [N][B][24] tensor at the front
    // of the buffer when the linearisation wrote its sparse form (KArgs::lin_sparse)
    const bool dense_grad = CONST && a.lin_sparse;
    const T* __restrict__ gvec = dense_grad ? a.lin + (size_t)b * (NX + NU) : lin + (NX * NX + NX * NU);
    const size_t gstride = dense_grad ? B * (NX + NU) : tstride;
    Tile16x8<T> cur, nxt;
    tile16x8_load(cur, lin + (size_t)(N - 1) * tstride, off);
    if constexpr (CONST) {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(gvec + (size_t)(N - 1) * gstride, (NX + NU) * S);
        cur.lx = buf_load1(r0, off.vLx, 0, T(0));
        cur.lu = buf_load1(r0, off.vLu, S * NX, T(0));
    }
    {
        // Every load of the prologue (first tile, terminal value function) is consumed here, before the loop: hipcc's s_waitcnt bookkeeping then enters the
        // loop with nothing pending.  Otherwise the loop header merges "22 loads pending" (from here) with "22 loads and
        // the 6 gain stores behind them pending" (from the back edge) into the tighter of the two counts, and every step
        // waits for the previous step's STORES to complete before it may touch its tile (measured: 17 % of the sweep).
        T touch = cur.lx + cur.lu;
#pragma unroll
        for (int r = 0; r < 4; ++r) touch += cur.A[r] + cur.Bm[r] + cur.lxx[r] + cur.lux[r] + cur.luu[r] + V[r] + Vx[r];
        asm volatile("" ::"v"(touch));
    }
    bool all_pd = true;
#ifdef ILQR_MFMA16_STAMPS
    long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_last = __builtin_readcyclecounter();
#endif
    const acc zero = {T(0), T(0), T(0), T(0)};
    auto arr = [](const acc& v, T* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; };
    auto vec = [](const T* v) { acc o = {v[0], v[1], v[2], v[3]}; return o; };
    // sum over the four lane groups: every lane (., c) ends with the total of column c
    auto group_sum = [&](T v) { return MF::group_sum(v, a16, a32); };

    for (int t = N - 1; t >= 0; --t) {
        // the next step's expansion does not depend on the carried value function: request it now
        if constexpr (CONST) {
            // l_x, l_u: 24 adjacent scalars, in the record (full form) or in the dense side tensor (sparse form)
            const __amdgpu_buffer_rsrc_t rn = make_rsrc(gvec + (size_t)(t > 0 ? t - 1 : 0) * gstride, (NX + NU) * S);
            nxt.lx = buf_load1(rn, off.vLx, 0, T(0));
            nxt.lu = buf_load1(rn, off.vLu, S * NX, T(0));
        } else {
            tile16x8_load(nxt, lin + (size_t)(t > 0 ? t - 1 : 0) * tstride, off);
        }
        ILQR_STAMP(8, off.vA);
        ILQR_STAMP(9, cur.Bm[0] + cur.A[3] + cur.lu + V[0]);

        // ---- Q-function (iLQR_class.py:100-104) ---------------------------------------------------------------
        // An f32 16x16x16 product is four 32-cycle MFMAs (the f32-input matrix rate is only twice a lone wave's
        // vector rate), so the matrix pipe is kept for the six matrix-matrix products and everything vector-shaped
        // stays on the vector ALU, which would otherwise idle behind the dependent MFMA chains.
        T Put[4], Pt[4], Quu[4], Qux[4], Qxx[4];
        // MFMA issue order is pinned (ILQR_MFMA_NEXT = a scheduling barrier behind every instruction): left alone, hipcc
        // issues each product's four dependent instructions back to back (40 cycles of latency each, measured: ~1100 of a
        // step's ~3500 cycles in front of the factorisation); round-robin over independent chains the pipe takes one
        // every 32 cycles.
        // Round 3: the order follows what the factorisation waits for.  The vector ALU cannot start on Q_uu before
        // (B'V)' and Q_uu itself are through the pipe, and nothing else is: (B'V)' goes first with (A'V)' filling the
        // pipe behind it while its result drains, then Q_uu's four instructions; Q_ux (needed by the substitution) and
        // Q_xx (needed by the value update) are issued last and execute UNDER the factorisation's ~1000 cycles of vector
        // work instead of in front of it (the matrix pipe runs beside the vector ALU; a lone wave only has to issue them).
#define ILQR_MFMA_NEXT() __builtin_amdgcn_sched_barrier(0)
        acc qx_acc, qxx_acc;
        {
            acc p0 = zero, p1 = zero, t0 = zero, t1 = zero;
            p0 = MF::mma(V[0], cur.Bm[0], p0); ILQR_MFMA_NEXT();
            p1 = MF::mma(V[1], cur.Bm[1], p1); ILQR_MFMA_NEXT();
            p0 = MF::mma(V[2], cur.Bm[2], p0); ILQR_MFMA_NEXT();
            p1 = MF::mma(V[3], cur.Bm[3], p1); ILQR_MFMA_NEXT();
            t0 = MF::mma(V[0], cur.A[0], t0); ILQR_MFMA_NEXT();
            t1 = MF::mma(V[1], cur.A[1], t1); ILQR_MFMA_NEXT();
            t0 = MF::mma(V[2], cur.A[2], t0); ILQR_MFMA_NEXT();
            t1 = MF::mma(V[3], cur.A[3], t1); ILQR_MFMA_NEXT();
            arr(p0 + p1, Put);
            ILQR_MFMA_NEXT();
            // Q_uu (the factorisation waits for it: two accumulators)
            acc u0 = vec(cur.luu), u1 = zero;
            u0 = MF::mma(Put[0], cur.Bm[0], u0); ILQR_MFMA_NEXT();
            u1 = MF::mma(Put[1], cur.Bm[1], u1); ILQR_MFMA_NEXT();
            u0 = MF::mma(Put[2], cur.Bm[2], u0); ILQR_MFMA_NEXT();
            u1 = MF::mma(Put[3], cur.Bm[3], u1); ILQR_MFMA_NEXT();
            arr(t0 + t1, Pt);
            ILQR_MFMA_NEXT();
            // Q_ux, Q_xx: in the shadow of the factorisation
            acc x = vec(cur.lux), y2 = vec(cur.lxx);
            x = MF::mma(Put[0], cur.A[0], x); ILQR_MFMA_NEXT();
            y2 = MF::mma(Pt[0], cur.A[0], y2); ILQR_MFMA_NEXT();
            x = MF::mma(Put[1], cur.A[1], x); ILQR_MFMA_NEXT();
            y2 = MF::mma(Pt[1], cur.A[1], y2); ILQR_MFMA_NEXT();
            x = MF::mma(Put[2], cur.A[2], x); ILQR_MFMA_NEXT();
            y2 = MF::mma(Pt[2], cur.A[2], y2); ILQR_MFMA_NEXT();
            x = MF::mma(Put[3], cur.A[3], x); ILQR_MFMA_NEXT();
            y2 = MF::mma(Pt[3], cur.A[3], y2); ILQR_MFMA_NEXT();
            arr(u0 + u1, Quu);
            qx_acc = x;
            qxx_acc = y2;
        }
        ILQR_STAMP(0, Put[0]);
        // Q_u = l_u + B'V_x and Q_x = l_x + A'V_x, column-indexed: 4 products per lane, summed over the lane groups
        T qu_c = T(0), qx_c = T(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qu_c += cur.Bm[r] * Vx[r];
            qx_c += cur.A[r] * Vx[r];
        }
        qu_c = cur.lu + group_sum(qu_c);
        qx_c = cur.lx + group_sum(qx_c);
#undef ILQR_MFMA_NEXT
        ILQR_STAMP(1, Quu[0]);
        ILQR_STAMP(2, qu_c + qx_c);

        // ---- gain solve (:109-110) ---------------------------------------------------------------------------------
        // Quu's lower triangle and Q_u to scalars: row i of Quu lives in lane group grp_of(i), register reg_of(i)
        T q[NU][NU], qu[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) q[i][j] = MF::readlane(Quu[MF::reg_of(i)], 16 * MF::grp_of(i) + j);
            qu[i] = MF::readlane(qu_c, i);
        }
        // this lane's right-hand side: column c of Q_ux (lanes 0..31, both halves solve the same 16 columns) or Q_u
        arr(qx_acc, Qux);          // (first reader of Q_ux: the right-hand sides)
        T rhs[NU];
        if constexpr (sizeof(T) == 4) {
            // f32: rows 0-3 of column c sit in lane (0, c), rows 4-7 in lane (1, c): one 16-lane swap per register
            // gives both lanes the whole column
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T lo, hi;
                MF::swap16(Qux[r], lo, hi);
                rhs[r] = g < 2 ? lo : qu[r];
                rhs[4 + r] = g < 2 ? hi : qu[4 + r];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const T col = MF::bperm(4 * (16 * MF::grp_of(i) + c), Qux[MF::reg_of(i)]);
                rhs[i] = g < 2 ? col : qu[i];
            }
        }
        ILQR_STAMP(3, rhs[0] + rhs[7] + q[7][7]);
        // Cholesky Quu = L L', redundantly in every lane (all operands are wave-uniform); Li[c] = 1 / L[c][c]
        T L[NU][NU], Li[NU];
        bool pd = true;
        T y[NU];
        if constexpr (sizeof(T) == 4) {
            // fp32 (round 3): the same factorisation and forward substitution on ROW PAIRS in packed FP32 -- rows 2p, 2p + 1
            // of a column share one v_pk_fma_f32 (the multiplier L[cc][s] / y[s] is one half of a pair, broadcast by
            // op_sel).  Every element still receives the same products in the same order (ascending s), the diagonal is
            // the pair element (cc, cc) of the same recurrence: bit for bit the scalar results, in ~75 fewer vector
            // instructions per step of a wave that is bound by its instruction count (tools/c5_issue_model.py).
            typedef float f2 __attribute__((ext_vector_type(2)));
            constexpr int NP = NU / 2;
            f2 L2[NP][NU];                       // L2[p][s] = (L[2p][s], L[2p + 1][s])
#pragma unroll
            for (int cc = 0; cc < NU; ++cc) {
                f2 v2[NP];
#pragma unroll
                for (int pp = cc / 2; pp < NP; ++pp) {
                    v2[pp].x = q[2 * pp >= cc ? 2 * pp : cc][cc];          // (row 2p < cc only when 2p + 1 = cc: unused half)
                    v2[pp].y = q[2 * pp + 1][cc];
#pragma unroll
                    for (int s2 = 0; s2 < cc; ++s2) {
                        const float m = (cc & 1) ? L2[cc / 2][s2].y : L2[cc / 2][s2].x;    // L[cc][s]
                        v2[pp] = __builtin_elementwise_fma(-L2[pp][s2], f2{m, m}, v2[pp]);
                    }
                }
                const float d = (cc & 1) ? v2[cc / 2].y : v2[cc / 2].x;
                pd = pd && (d > 0.0f);
                const float inv = fast_rsqrt(pd ? d : 1.0f);
                Li[cc] = inv;
#pragma unroll
                for (int pp = cc / 2; pp < NP; ++pp) L2[pp][cc] = v2[pp] * f2{inv, inv};
            }
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) L[i][j] = (i & 1) ? L2[i / 2][j].y : L2[i / 2][j].x;
            all_pd = all_pd && pd;
            ILQR_STAMP(4, Li[7]);
            if (pd) {
                f2 w2[NP];
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) w2[pp] = f2{rhs[2 * pp], rhs[2 * pp + 1]};
#pragma unroll
                for (int s2 = 0; s2 < NU; ++s2) {          // L y = rhs, one column of L at a time (each entry: ascending s)
                    y[s2] = ((s2 & 1) ? w2[s2 / 2].y : w2[s2 / 2].x) * Li[s2];
#pragma unroll
                    for (int pp = (s2 + 1) / 2; pp < NP; ++pp)
                        w2[pp] = __builtin_elementwise_fma(-L2[pp][s2], f2{y[s2], y[s2]}, w2[pp]);
                }
            }
        } else {
#pragma unroll
        for (int cc = 0; cc < NU; ++cc) {
            T d = q[cc][cc];
#pragma unroll
            for (int s = 0; s < cc; ++s) d -= L[cc][s] * L[cc][s];
            pd = pd && (d > T(0));
            const T inv = fast_rsqrt(pd ? d : T(1));
            L[cc][cc] = d * inv;
            Li[cc] = inv;
#pragma unroll
            for (int i = cc + 1; i < NU; ++i) {
                T v = q[i][cc];
#pragma unroll
                for (int s = 0; s < cc; ++s) v -= L[i][s] * L[cc][s];
                L[i][cc] = v * inv;
            }
        }
        all_pd = all_pd && pd;
        ILQR_STAMP(4, Li[7]);
        }
        if (pd) {
            if constexpr (sizeof(T) != 4) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {             // L y = rhs
                T v = rhs[i];
#pragma unroll
                for (int s = 0; s < i; ++s) v -= L[i][s] * y[s];
                y[i] = v * Li[i];
            }
            }
#pragma unroll
            for (int i = NU - 1; i >= 0; --i) {        // L' z = y
                T v = y[i];
#pragma unroll
                for (int s = i + 1; s < NU; ++s) v -= L[s][i] * y[s];
                y[i] = v * Li[i];
            }
        } else {
            // not positive definite: Gaussian elimination with partial pivoting on the full matrix, what the reference's
            // jnp.linalg.solve always does (iLQR_class.py:109-110); rare, so the upper triangle is only fetched here
            T Ac[NU][NU], rr[NU][1];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
#pragma unroll
                for (int j = 0; j < NU; ++j)
                    Ac[i][j] = j <= i ? q[i][j] : MF::readlane(Quu[MF::reg_of(i)], 16 * MF::grp_of(i) + j);
                rr[i][0] = rhs[i];
            }
            lu_solve_inplace<T, NU, 1>(Ac, rr);
#pragma unroll
            for (int i = 0; i < NU; ++i) y[i] = rr[i][0];
        }
        // y = Quu^-1 rhs: lanes 0..31 hold column c of -K, lanes 32..63 hold -k
        ILQR_STAMP(5, y[0]);

        // V_x+ = Q_x + K'Q_u (:113), column-indexed in the lanes that solved a column of K, then row-indexed for the
        // next step: register r of lane (g, .) <- entry ROW(g, r), held by lane (0, ROW(g, r))
        T vxn = qx_c;
#pragma unroll
        for (int i = 0; i < NU; ++i) vxn -= y[i] * qu[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) Vx[r] = MF::bperm(aRow[r], vxn);

        // K in C-layout (rows >= 8 of the padded matrix are zero): lane (g, c) register r <- K[ROW(g, r)][c]
        T K[4];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) K[r] = g == 0 ? -y[r] : (g == 1 ? -y[4 + r] : T(0));
        } else {
            // f64: every lane group holds rows g, g + 4 (registers 0, 1); lane groups 2, 3 solved Q_u, so they fetch
            // their two entries of column c from a lane that solved it
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                T pick = T(0);
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const T cand = MF::bperm(4 * (16 * (gg & 1) + c), -y[MF::row(gg, r)]);
                    pick = g == gg ? cand : pick;
                }
                K[r] = pick;
            }
            K[2] = T(0);
            K[3] = T(0);
        }

        // ---- gains out: K_t row-major [8][16], then k_t [8]; lanes with nothing to store carry a dropped offset ------
        {
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gains + ((size_t)t * B + b) * R, R * S);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (MF::row(0, r) < NU) {
                    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(K[r]), rg, off.vK + S * NX * RS * r, 0, 0);
                    else buf_store1(rg, off.vK + S * NX * RS * r, 0, K[r]);
                }
            T kk[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) kk[i] = -y[i];
#pragma unroll
            for (int i = 0; i < NU; i += 16 / S) buf_store16(rg, off.vk + S * i, kk + i);
        }

        // ---- V+ = Q_xx + Q_ux' K (:114); two accumulators: the next step's first product waits for it -------------------
        {
            arr(qxx_acc, Qxx);
            acc d0 = vec(Qxx), d1 = zero;
            d0 = MF::mma(Qux[0], K[0], d0);
            d1 = MF::mma(Qux[1], K[1], d1);
            if constexpr (sizeof(T) == 4) {        // (f64: registers 2, 3 of K are rows >= 8: nothing to add)
                d0 = MF::mma(Qux[2], K[2], d0);
                d1 = MF::mma(Qux[3], K[3], d1);
            }
            arr(d0 + d1, V);
        }
        ILQR_STAMP(6, V[0] + Vx[0]);
        if constexpr (CONST) { cur.lx = nxt.lx; cur.lu = nxt.lu; }
        else cur = nxt;
        ILQR_STAMP(7, cur.A[0]);
    }
#ifdef ILQR_MFMA16_STAMPS
    if (a.probe && blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) a.probe[8 + k] = stamp_acc[k];
    }
#endif
    if (lane == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

}  // namespace ilqr

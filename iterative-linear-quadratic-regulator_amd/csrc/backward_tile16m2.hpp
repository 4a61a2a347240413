// backward_tile16m2.hpp -- the DPP backward sweep for n_x = 4, n_u = 2 (the reference's fully actuated double
// pendulum, double_pendulum_sys.py; iLQR_class.py:79-161).  Same mapping as backward_tile16.hpp -- lane (i, j) of a
// 16-lane DPP row owns V_xx[i][j], 4 trajectories per wave -- with a 2 x 2 Q_uu solved in closed form.
//
// Tile, 64 scalars per (t, b):
//   [ 0..15]  SK[c][d] = f_x[(c + d) % 4][c]            (as the n_u = 1 tile)
//   [16..31]  l_xx[i][j]
//   [32..63]  for j = 0..3, 8 scalars: f_u[j][0], f_u[j][1], l_x[j], l_ux[0][j], l_ux[1][j], e0, e1, e2
//             e of j = 0: l_u[0], l_u[1], l_uu[0][0];   e of j = 1: l_uu[0][1], l_uu[1][1], 0;   else 0
// = the 58 algorithmic scalars + 6 pad.  Gain record (gain_record(4, 2) = 12): K[0][0..3], K[1][0..3], k[0], k[1].
//
// Q_ux is computed twice, in column form (lane holds Q_ux[c][j]) for the gains and in row form (Q_ux[c][i]) for the
// value update, the latter through V_xx's symmetry (see tile16_step_f32): no transposition through the LDS crossbar.
// The instruction order is left to hipcc; fp32 keeps a ring of 8 tiles in flight (asm loads, self-counted vmcnt),
// fp64 leaves the loads to hipcc too (one tile of lookahead).
#pragma once
#include "backward_tile16.hpp"

namespace ilqr {

constexpr int kTile16M2 = 64;

template <typename T> struct Tile16M2 {
    T ski[4], skj[4];
    T gj[8];   // group j: f_u[j][0], f_u[j][1], l_x[j], l_ux[0][j], l_ux[1][j], e0, e1, e2
    T gi[8];   // group i (row form): f_u[i][0], f_u[i][1], -, l_ux[0][i], l_ux[1][i], ...
    T lxx;
};

struct TileOffsetsM2 { int vi, vj, vl, gi, gj; };

template <typename T>
ILQR_DEV void tile16m2_load(Tile16M2<T>& tl, __amdgpu_buffer_rsrc_t r, const TileOffsetsM2& o, int soff) {
    BufLoad<0, T>::v4(r, o.vi, soff, tl.ski);
    BufLoad<0, T>::v4(r, o.vj, soff, tl.skj);
    BufLoad<0, T>::v4(r, o.gj, soff, tl.gj);
    BufLoad<4, T>::v4(r, o.gj, soff, tl.gj + 4);
    BufLoad<0, T>::v4(r, o.gi, soff, tl.gi);
    BufLoad<4, T>::v4(r, o.gi, soff, tl.gi + 4);
    tl.lxx = BufLoad<0, T>::v1(r, o.vl, soff);
}

// ---- the tile ring (same technique and same rules as RawTile in backward_tile16.hpp: all loads of a slot in ONE asm
// statement opening with s_nop 4, early-clobber outputs, waits counted by hand) -------------------------------------
template <typename T> struct RawTileM2;
template <> struct RawTileM2<float> {
    static constexpr int NLOAD = 7;
    f32x4n ski, skj, gj0, gj1;
    typedef float f32x2n __attribute__((ext_vector_type(2)));
    f32x2n gia, gib;   // f_u[i][0..1] ; l_ux[0..1][i]
    float lxx;
#define ILQR_RAWTILE_M2(OUT)                                                                 \
    asm volatile(                                                                            \
        "s_nop 4\n\t"                                                                        \
        "buffer_load_dwordx4 %0, %7, %12, %13 offen\n\t"                                     \
        "buffer_load_dwordx4 %1, %8, %12, %13 offen\n\t"                                     \
        "buffer_load_dwordx4 %2, %9, %12, %13 offen\n\t"                                     \
        "buffer_load_dwordx4 %3, %9, %12, %13 offen offset:16\n\t"                           \
        "buffer_load_dwordx2 %4, %10, %12, %13 offen\n\t"                                    \
        "buffer_load_dwordx2 %5, %10, %12, %13 offen offset:12\n\t"                          \
        "buffer_load_dword %6, %11, %12, %13 offen"                                          \
        : OUT(ski), OUT(skj), OUT(gj0), OUT(gj1), OUT(gia), OUT(gib), OUT(lxx)               \
        : "v"(o.vi), "v"(o.vj), "v"(o.gj), "v"(o.gi), "v"(o.vl), "s"(srd), "s"(soff)         \
        : "memory")
    // FIRST: prologue (early-clobber outputs); refills tie the destinations to the consumed tile (see RawTile)
    template <bool FIRST> ILQR_DEV void issue(const i32x4& srd, const TileOffsetsM2& o, int soff) {
        if constexpr (FIRST) ILQR_RAWTILE_M2(ILQR_OUT_FIRST);
        else ILQR_RAWTILE_M2(ILQR_OUT_REFILL);
    }
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%7)"
                     : "+v"(ski), "+v"(skj), "+v"(gj0), "+v"(gj1), "+v"(gia), "+v"(gib), "+v"(lxx) : "i"(N) : "memory");
    }
    ILQR_DEV void unpack(Tile16M2<float>& t) const {
        t.ski[0] = ski.x; t.ski[1] = ski.y; t.ski[2] = ski.z; t.ski[3] = ski.w;
        t.skj[0] = skj.x; t.skj[1] = skj.y; t.skj[2] = skj.z; t.skj[3] = skj.w;
        t.gj[0] = gj0.x; t.gj[1] = gj0.y; t.gj[2] = gj0.z; t.gj[3] = gj0.w;
        t.gj[4] = gj1.x; t.gj[5] = gj1.y; t.gj[6] = gj1.z; t.gj[7] = gj1.w;
        t.gi[0] = gia.x; t.gi[1] = gia.y; t.gi[3] = gib.x; t.gi[4] = gib.y;
        t.lxx = lxx;
    }
};

// sum down the 4 rows of a column, result in every row
template <typename T> ILQR_DEV T col_sum(T v) {
    v += dpp<kDown2>(v);
    v += dpp<kDown1>(v);
    return v;
}

// One Riccati step.  V = V_xx[i][j]; vx = V_x[j] (column form).  Outputs the lane's gains: K0 = K[0][j], K1 = K[1][j]
// (column form), k0, k1 (replicated); pd = (Q_uu + mu I) positive definite.
template <typename T, bool REG>
ILQR_DEV void tile16m2_step(const Tile16M2<T>& c, T m0, T m1, T mu, T& V, T& vx, T& K0, T& K1, T& k0, T& k1, bool& pd) {
    const T fu0j = c.gj[0], fu1j = c.gj[1];
    // P = f_x' V_xx
    const T P = contract_col(c.ski, V);
    // pu_c[j] = sum_i f_u[i][c] V[i][j]  (column form)
    const T pu0 = col_sum(c.gi[0] * V), pu1 = col_sum(c.gi[1] * V);
    // Q_ux[c][j] = l_ux[c][j] + (pu_c f_x)[j] ; Q_x[j] = l_x[j] + (f_x' V_x)[j]
    const T Qux0 = contract_row(c.skj, pu0, c.gj[3]);
    const T Qux1 = contract_row(c.skj, pu1, c.gj[4]);
    const T Qx = contract_row(c.skj, vx, c.gj[2]);
    // Q_uu = l_uu + pu f_u ; Q_u = l_u + f_u' V_x   (the l_u / l_uu scalars ride lanes j = 0, 1 into the quad sums)
    const T q00 = quad_sum(pu0 * fu0j + m0 * c.gj[7]);
    const T q01 = quad_sum(pu0 * fu1j + m1 * c.gj[5]);
    const T q11 = quad_sum(pu1 * fu1j + m1 * c.gj[6]);
    const T qu0 = quad_sum(fu0j * vx + m0 * c.gj[5]);
    const T qu1 = quad_sum(fu1j * vx + m0 * c.gj[6]);
    // [K | k] = -(Q_uu + mu I)^-1 [Q_ux | Q_u]: closed form of the 2 x 2 solve (iLQR_class.py:109-110)
    const T a = REG ? q00 + mu : q00, d = REG ? q11 + mu : q11, b = q01;
    const T det = a * d - b * b;
    pd = (a > T(0)) && (det > T(0));
    const T inv = fast_rcp(det);
    const T ia = d * inv, ib = -(b * inv), id = a * inv;       // inverse = [[ia, ib], [ib, id]]
    K0 = -(ia * Qux0 + ib * Qux1);
    K1 = -(ib * Qux0 + id * Qux1);
    k0 = -(ia * qu0 + ib * qu1);
    k1 = -(ib * qu0 + id * qu1);
    // row form of Q_ux: pr_c[i] = sum_k f_u[k][c] V[i][k] (V_xx symmetric up to rounding), then down the column
    const T pr0 = quad_sum(fu0j * V), pr1 = quad_sum(fu1j * V);
    const T Quxi0 = contract_col(c.ski, pr0) + c.gi[3];
    const T Quxi1 = contract_col(c.ski, pr1) + c.gi[4];
    const T Qxx = contract_row(c.skj, P, c.lxx);
    if constexpr (!REG) {
        // short form (:113-114): V_xx = Q_xx + Q_ux' K ; V_x = Q_x + K' Q_u
        V = Qxx + Quxi0 * K0 + Quxi1 * K1;
        vx = Qx + K0 * qu0 + K1 * qu1;
    } else {
        // full update for a regularised gain: V_xx = Q_xx + K' Q_uu K + K' Q_ux + Q_ux' K (Q_uu without mu)
        const T Ki0 = -(ia * Quxi0 + ib * Quxi1), Ki1 = -(ib * Quxi0 + id * Quxi1);   // K[c][i]
        const T QK0 = q00 * K0 + q01 * K1, QK1 = q01 * K0 + q11 * K1;                 // (Q_uu K)[c][j]
        V = Qxx + Ki0 * (QK0 + Qux0) + Ki1 * (QK1 + Qux1) + Quxi0 * K0 + Quxi1 * K1;
        const T Qk0 = q00 * k0 + q01 * k1, Qk1 = q01 * k0 + q11 * k1;                 // Q_uu k
        vx = Qx + K0 * (Qk0 + qu0) + K1 * (Qk1 + qu1) + Qux0 * k0 + Qux1 * k1;
    }
}

// ---- fp32 without regularisation: the step as a scheduled instruction stream (gen_tile16m2_step.py) ---------------------------
// Its view of the tile: SK[j][0..3], the eight scalars of group j, and four single scalars whose QUAD holds what the lane
// needs of row i -- a = SK[i][j] (quad: SK[i][0..3]), c = group i's scalar j (quad: f_u[i][0], f_u[i][1], ...) -- which the step
// reads through DPP quad broadcasts instead of loading SK[i][.] and group i a second time (60 instead of 84 bytes per lane
// and step); Q_ux reaches its row form (lane (i, j): Q_ux[c][i]) by four masked quad broadcasts of the column form.
struct TileQ2 {
    float sj[4];
    float g[8];
    float a, lxx, c;
};
// which scalar of the step a lane stores into the gain record, as wave-wide lane masks (loop invariants in SGPRs)
struct GainSel {
    unsigned long long i1, jn0, r23;     // lanes of row 1; lanes with j != 0; lanes of rows 2, 3
    ILQR_DEV static GainSel of(int i, int j) {
        GainSel s;
        s.i1 = __ballot(i == 1);
        s.jn0 = __ballot(j != 0);
        s.r23 = __ballot(i >= 2);
        return s;
    }
};
#include "tile16m2_step_gen.inc"

struct TileOffsetsQ2 { int vj, gj, vl, vc; };
ILQR_DEV void tileq2_load_buf(TileQ2& t, __amdgpu_buffer_rsrc_t r, const TileOffsetsQ2& o, int soff) {
    BufLoad<0, float>::v4(r, o.vj, soff, t.sj);
    BufLoad<0, float>::v4(r, o.gj, soff, t.g);
    BufLoad<4, float>::v4(r, o.gj, soff, t.g + 4);
    t.a = BufLoad<0, float>::v1(r, o.vl, soff);
    t.lxx = BufLoad<16, float>::v1(r, o.vl, soff);
    t.c = BufLoad<0, float>::v1(r, o.vc, soff);
}
ILQR_DEV void tileq2_load_lds(TileQ2& t, const float* tp, int i, int j, int l16) {
    const float4 s = *reinterpret_cast<const float4*>(tp + 4 * j);
    const float4 g0 = *reinterpret_cast<const float4*>(tp + 32 + 8 * j);
    const float4 g1 = *reinterpret_cast<const float4*>(tp + 36 + 8 * j);
    t.sj[0] = s.x; t.sj[1] = s.y; t.sj[2] = s.z; t.sj[3] = s.w;
    t.g[0] = g0.x; t.g[1] = g0.y; t.g[2] = g0.z; t.g[3] = g0.w;
    t.g[4] = g1.x; t.g[5] = g1.y; t.g[6] = g1.z; t.g[7] = g1.w;
    t.a = tp[l16];
    t.lxx = tp[16 + l16];
    t.c = tp[32 + 8 * i + j];
}
// the ring slot of this view (rules as RawTileM2)
struct RawTileQ2 {
    static constexpr int NLOAD = 6;
    f32x4n sj, g0, g1;
    float a, lxx, c;
#define ILQR_RAWTILE_Q2(OUT)                                                                 \
    asm volatile(                                                                            \
        "s_nop 4\n\t"                                                                        \
        "buffer_load_dwordx4 %0, %6, %10, %11 offen\n\t"                                     \
        "buffer_load_dwordx4 %1, %7, %10, %11 offen\n\t"                                     \
        "buffer_load_dwordx4 %2, %7, %10, %11 offen offset:16\n\t"                           \
        "buffer_load_dword %3, %8, %10, %11 offen\n\t"                                       \
        "buffer_load_dword %4, %8, %10, %11 offen offset:64\n\t"                             \
        "buffer_load_dword %5, %9, %10, %11 offen"                                            \
        : OUT(sj), OUT(g0), OUT(g1), OUT(a), OUT(lxx), OUT(c)                                \
        : "v"(o.vj), "v"(o.gj), "v"(o.vl), "v"(o.vc), "s"(srd), "s"(soff)                    \
        : "memory")
    template <bool FIRST> ILQR_DEV void issue(const i32x4& srd, const TileOffsetsQ2& o, int soff) {
        if constexpr (FIRST) ILQR_RAWTILE_Q2(ILQR_OUT_FIRST);
        else ILQR_RAWTILE_Q2(ILQR_OUT_REFILL);
    }
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(sj), "+v"(g0), "+v"(g1), "+v"(a), "+v"(lxx), "+v"(c) : "i"(N) : "memory");
    }
    ILQR_DEV void unpack(TileQ2& t) const {
        t.sj[0] = sj.x; t.sj[1] = sj.y; t.sj[2] = sj.z; t.sj[3] = sj.w;
        t.g[0] = g0.x; t.g[1] = g0.y; t.g[2] = g0.z; t.g[3] = g0.w;
        t.g[4] = g1.x; t.g[5] = g1.y; t.g[6] = g1.z; t.g[7] = g1.w;
        t.a = a; t.lxx = lxx; t.c = c;
    }
};

template <typename T, bool REG>
__global__ void __launch_bounds__(256) backward_tile16m2_kernel(KArgs<T> a) {
    constexpr int R = gain_record(4, 2);   // 12
    const int lane = threadIdx.x & 63;
    const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3;
    const int gidx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool valid = gidx < a.B;
    const int b = valid ? gidx : a.B - 1;  // out-of-range groups shadow the last trajectory, never store
    const int st = a.status[b];
    const bool act = valid && traj_active(st);
    if (__ballot(act) == 0ull) return;
    if (a.reset_slots && act && l16 == 0) a.cur_slot[b] = 0;   // see linearize_kernel
    const size_t B = a.B;
    const int N = a.N;
    T V = a.term[(size_t)(4 + l16) * B + b];
    T vx = a.term[(size_t)j * B + b];
    const unsigned lin_bytes = (unsigned)((size_t)N * B * kTile16M2 * sizeof(T));
    const unsigned gain_bytes = (unsigned)((size_t)N * B * R * sizeof(T));
    const __amdgpu_buffer_rsrc_t rlin = make_rsrc(a.lin, lin_bytes);
    const __amdgpu_buffer_rsrc_t rgain = make_rsrc(a.gains, gain_bytes);
    const int tstride = (int)(B * kTile16M2 * sizeof(T));
    const int rstride = (int)(B * R * sizeof(T));
    TileOffsetsM2 off;
    off.vi = (int)((b * kTile16M2 + 4 * i) * sizeof(T));
    off.vj = (int)((b * kTile16M2 + 4 * j) * sizeof(T));
    off.vl = (int)((b * kTile16M2 + 16 + l16) * sizeof(T));
    off.gi = (int)((b * kTile16M2 + 32 + 8 * i) * sizeof(T));
    off.gj = (int)((b * kTile16M2 + 32 + 8 * j) * sizeof(T));
    // lanes (0, j) store K[0][j], lanes (1, j) K[1][j], lanes (2, 0) and (2, 1) k[0], k[1]
    const bool storer = act && (i < 2 || (i == 2 && j < 2));
    const int rec_off = (int)((b * R + (i < 2 ? 4 * i + j : 8 + j)) * sizeof(T));
    const T m0 = T(j == 0), m1 = T(j == 1);
    bool all_pd = true;
    auto do_step = [&](const Tile16M2<T>& c, int t) {
        T K0, K1, k0, k1;
        bool pd;
        tile16m2_step<T, REG>(c, m0, m1, a.mu, V, vx, K0, K1, k0, k1, pd);
        all_pd = all_pd && pd;
        const T out = (i == 0) ? K0 : ((i == 1) ? K1 : ((j == 0) ? k0 : k1));
        if (storer) buf_store1(rgain, rec_off, uniform(t * rstride), out);
    };
    if constexpr (sizeof(T) == 4 && !REG) {
        // fp32, mu = 0: the scheduled step on its own view of the tile; ring as below
        TileOffsetsQ2 oq;
        oq.vj = off.vj;
        oq.gj = off.gj;
        oq.vl = (int)((b * kTile16M2 + l16) * sizeof(T));
        oq.vc = (int)((b * kTile16M2 + 32 + 8 * i + j) * sizeof(T));
        LaneConst<float> lc;
        lc.m0 = m0;
        lc.m1 = m1;
        lc.tr_byte = 0;
        const GainSel sel = GainSel::of(i, j);
        auto do_step_q = [&](const TileQ2& c, int t) {
            float out;
            bool pd;
            tile16m2_step_f32(c, lc, sel, V, vx, out, pd);
            all_pd = all_pd && pd;
            if (storer) buf_store1(rgain, rec_off, uniform(t * rstride), out);
        };
        constexpr int D = 8, NL = RawTileQ2::NLOAD;
        static_assert((D - 1) * (NL + 1) <= 63, "vmcnt field");
        int t = N - 1;
        for (int r = N % D; r > 0; --r, --t) {
            TileQ2 c;
            tileq2_load_buf(c, rlin, oq, uniform(t * tstride));
            do_step_q(c, t);
        }
        if (t >= 0) {
            const i32x4 srd = make_srd(a.lin, lin_bytes);
            RawTileQ2 ring[D];
#pragma unroll
            for (int u = 0; u < D; ++u) ring[u].template issue<true>(srd, oq, uniform((t - u) * tstride));
#pragma unroll
            for (int u = 0; u < D; ++u) {
                ring[u].template wait<(D - 1) * NL>();
                TileQ2 c;
                ring[u].unpack(c);
                do_step_q(c, t - u);
                const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                ring[u].template issue<false>(srd, oq, uniform(tn * tstride));
            }
            for (t -= D; t >= 0; t -= D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    ring[u].template wait<(D - 1) * (NL + 1)>();
                    TileQ2 c;
                    ring[u].unpack(c);
                    do_step_q(c, t - u);
                    const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                    ring[u].template issue<false>(srd, oq, uniform(tn * tstride));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if constexpr (sizeof(T) == 4) {
        // fp32: D tiles per lane in flight, self-counted vmcnt (see backward_tile16_kernel); every step issues exactly
        // NL loads and one store, so slot u has landed when at most (D-1)*(NL+1) younger operations are outstanding
        constexpr int D = 8, NL = RawTileM2<T>::NLOAD;
        static_assert((D - 1) * (NL + 1) <= 63, "vmcnt field");
        int t = N - 1;
        for (int r = N % D; r > 0; --r, --t) {   // remainder steps first, so that the ring runs whole passes only
            Tile16M2<T> c;
            tile16m2_load(c, rlin, off, uniform(t * tstride));
            do_step(c, t);
        }
        if (t >= 0) {
            const i32x4 srd = make_srd(a.lin, lin_bytes);
            RawTileM2<T> ring[D];
#pragma unroll
            for (int u = 0; u < D; ++u) ring[u].template issue<true>(srd, off, uniform((t - u) * tstride));
#pragma unroll
            for (int u = 0; u < D; ++u) {   // first pass: the prologue's loads may be the only operations in flight
                ring[u].template wait<(D - 1) * NL>();
                Tile16M2<T> c;
                ring[u].unpack(c);
                do_step(c, t - u);
                const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                ring[u].template issue<false>(srd, off, uniform(tn * tstride));
            }
            for (t -= D; t >= 0; t -= D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    ring[u].template wait<(D - 1) * (NL + 1)>();
                    Tile16M2<T> c;
                    ring[u].unpack(c);
                    do_step(c, t - u);
                    const int tn = (t - u - D) > 0 ? (t - u - D) : 0;   // clamped: branch-free refill, drained below
                    ring[u].template issue<false>(srd, off, uniform(tn * tstride));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        // fp64: the loads are left to hipcc (one tile of lookahead)
        Tile16M2<T> cur, nxt;
        tile16m2_load(cur, rlin, off, uniform((N - 1) * tstride));
        for (int t = N - 1; t >= 0; --t) {
            const int tn = t > 0 ? t - 1 : 0;
            tile16m2_load(nxt, rlin, off, uniform(tn * tstride));
            do_step(cur, t);
            cur = nxt;
        }
    }
    if (act && l16 == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

// position in the 64-scalar tile of entry e of the dense ILQR_LIN record of a (4, 2) system:
// record = [f_x 16 | f_u 8 ([i][c]) | l_x 4 | l_u 2 | l_xx 16 | l_ux 8 ([c][j]) | l_uu 4 ([c][d])]
ILQR_DEV int tile16m2_index_of(int e) {
    if (e < 16) { const int i = e >> 2, j = e & 3; return 4 * j + ((i - j + 4) & 3); }   // f_x[i][j]
    e -= 16;
    if (e < 8) return 32 + 8 * (e >> 1) + (e & 1);         // f_u[i][c]
    e -= 8;
    if (e < 4) return 32 + 8 * e + 2;                      // l_x[i]
    e -= 4;
    if (e < 2) return 32 + 5 + e;                          // l_u[c] -> e0, e1 of group 0
    e -= 2;
    if (e < 16) return 16 + e;                             // l_xx
    e -= 16;
    if (e < 8) return 32 + 8 * (e & 3) + 3 + (e >> 2);     // l_ux[c][j]
    e -= 8;
    // l_uu[c][d]: [0][0] -> group 0 e2; [0][1] and [1][0] -> group 1 e0 (symmetric part, see scatter); [1][1] -> group 1 e1
    if (e == 0) return 32 + 7;
    if (e == 3) return 32 + 8 + 6;
    return 32 + 8 + 5;
}

template <typename T>
__global__ void tile16m2_gather_dense_kernel(T* dense, const T* lin, int B, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * 58) return;
    const int e = (int)(idx % 58);
    const int t = (int)((idx / 58) % N);
    const int b = (int)(idx / ((size_t)58 * N));
    dense[idx] = lin[((size_t)t * B + b) * kTile16M2 + tile16m2_index_of(e)];
}

// dense records -> tiles (pads zeroed by the caller's memset).  l_uu[0][1] and l_uu[1][0] share one slot: the sweep
// uses the symmetric Q_uu of the reference's quadratic forms, so the mean of the two is stored.
template <typename T>
__global__ void tile16m2_scatter_dense_kernel(const T* dense, T* lin, int B, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * 58) return;
    const int e = (int)(idx % 58);
    const int t = (int)((idx / 58) % N);
    const int b = (int)(idx / ((size_t)58 * N));
    if (e == 56) return;                                   // l_uu[1][0]: folded into [0][1] below
    T v = dense[idx];
    if (e == 55) v = T(0.5) * (v + dense[idx + 1]);
    lin[((size_t)t * B + b) * kTile16M2 + tile16m2_index_of(e)] = v;
}

}  // namespace ilqr

// backward_tile16.hpp -- backward Riccati sweep for n_x = 4, n_u = 1 (the north-star
// shape: under-actuated double pendulum), one 16-lane DPP row per trajectory.
//
// Why: the sweep is a 200-step dependent chain per trajectory; with one lane per
// trajectory a batch of 4096 is 64 waves on a 1024-SIMD chip and each step costs ~300
// serial instructions.  Here lane (i, j) = 4*i + j of a 16-lane DPP row owns element
// (i, j) of the 4x4 value Hessian V_xx, a wave carries 4 trajectories, and a batch of
// 4096 is 1024 waves = one per SIMD.  The small dense products of
// iLQR_class.py:100-104 become 4-term contractions whose cross-lane operands arrive
// through DPP row rotations (down a column) and quad permutations (along a row), fused
// into the multiply-add itself (v_fmac_f32_dpp) -- no LDS, no MFMA (n, m are far too
// small for either to pay).  A step is ~50 vector instructions.
//
// Tile layout (48 scalars per (t, b), written by linearize_kernel<..., TILE16=true>):
//   [ 0..15]  SK[c][d] = f_x[(c+d)%4][c]        column c of A_t, rotated so that entry d is the
//                                                coefficient that meets the d-th rotation of the
//                                                moving operand -- every lane reads "its" column
//                                                with ONE 16-byte load and a static register order
//   [16..31]  l_xx[i][j]
//   [32..47]  for j = 0..3: { f_u[j], l_x[j], l_ux[j], e_j },  e_0 = l_u, e_1 = l_uu, e_2 = e_3 = 0
// = the 46 algorithmic scalars + 2 pad, each stored once: no byte inflation over the dense form.
//
// Because tile loads never depend on the carried value function, a ring of D tiles per lane is
// kept in flight in registers (D*5 loads per lane outstanding) so HBM latency hides under compute.
#pragma once
#include "dynamics.hpp"

namespace ilqr {

// ---- DPP plumbing ------------------------------------------------------------------------------
constexpr int dpp_quad(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kRowRor = 0x120;                     // row_ror:n  dst[l] = src[(l - n) mod 16]
constexpr int kDown1 = kRowRor + 12;               // lane l <- lane l + 4   (next matrix row)
constexpr int kDown2 = kRowRor + 8;                // lane l <- lane l + 8
constexpr int kDown3 = kRowRor + 4;                // lane l <- lane l + 12
constexpr int kRight1 = dpp_quad(1, 2, 3, 0);      // lane (i, j) <- lane (i, j + 1)
constexpr int kRight2 = dpp_quad(2, 3, 0, 1);
constexpr int kRight3 = dpp_quad(3, 0, 1, 2);
constexpr int kSwap1 = dpp_quad(1, 0, 3, 2);       // butterfly partners inside a quad

template <int CTRL, int BANK = 0xf, bool BOUND = true> ILQR_DEV int dpp_bits(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, BOUND);
}
template <int CTRL> ILQR_DEV float dpp(float v) {
    return __int_as_float(dpp_bits<CTRL>(0, __float_as_int(v)));
}
template <int CTRL> ILQR_DEV double dpp(double v) {
    // DPP moves 32 bits: a double crosses lanes as two halves
    const int lo = dpp_bits<CTRL>(0, __double2loint(v));
    const int hi = dpp_bits<CTRL>(0, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// write only the lanes of matrix row BANK (bank_mask = 1 << row); the others keep `old`
template <int CTRL, int BANK> ILQR_DEV float dpp_row(float old, float v) {
    return __int_as_float(dpp_bits<CTRL, BANK, false>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int BANK> ILQR_DEV double dpp_row(double old, double v) {
    const int lo = dpp_bits<CTRL, BANK, false>(__double2loint(old), __double2loint(v));
    const int hi = dpp_bits<CTRL, BANK, false>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

constexpr int kTile16 = 48;

template <typename T> struct Tile16 {
    T ski[4];  // SK[i][0..3]: column i of A, for the contraction down the rows
    T skj[4];  // SK[j][0..3]: column j of A, for the contractions along the row
    T vj[4];   // f_u[j], l_x[j], l_ux[j], e_j
    T lxx;     // l_xx[i][j]
    T bi;      // f_u[i]
};

template <typename T> ILQR_DEV void tile16_load(Tile16<T>& tl, const T* __restrict__ tp, int i, int j, int l16) {
    using V4 = typename Vec4<T>::type;
    const V4 a = *reinterpret_cast<const V4*>(tp + 4 * i);
    const V4 c = *reinterpret_cast<const V4*>(tp + 4 * j);
    const V4 v = *reinterpret_cast<const V4*>(tp + 32 + 4 * j);
    tl.ski[0] = a.x; tl.ski[1] = a.y; tl.ski[2] = a.z; tl.ski[3] = a.w;
    tl.skj[0] = c.x; tl.skj[1] = c.y; tl.skj[2] = c.z; tl.skj[3] = c.w;
    tl.vj[0] = v.x; tl.vj[1] = v.y; tl.vj[2] = v.z; tl.vj[3] = v.w;
    tl.lxx = tp[16 + l16];
    tl.bi = tp[32 + 4 * i];
}

// contraction along the matrix row: sum_d skj[d] * w[(j + d) % 4]
template <typename T> ILQR_DEV T contract_row(const T* skj, T w) {
    T acc = skj[0] * w;
    acc += skj[1] * dpp<kRight1>(w);
    acc += skj[2] * dpp<kRight2>(w);
    acc += skj[3] * dpp<kRight3>(w);
    return acc;
}
// sum over the 4 lanes of a quad, result in every lane of the quad
template <typename T> ILQR_DEV T quad_sum(T v) {
    v += dpp<kSwap1>(v);
    v += dpp<kRight2>(v);
    return v;
}

// One Riccati step (iLQR_class.py:100-114) on the lane-distributed state:
//   V  = V_xx[i][j] at lane (i, j);  vx = V_x[j] ("column form": replicated down the rows).
// Returns K[j] (column form) and k (replicated); pd = Q_uu (+mu) > 0.
template <typename T, bool REG>
ILQR_DEV void tile16_step(const Tile16<T>& c, T mu, T& V, T& vx, T& Kj, T& kff, bool& pd) {
    // P = f_x' V_xx :  P[i][j] = sum_d A[(i+d)%4][i] * V[(i+d)%4][j]
    T P = c.ski[0] * V;
    P += c.ski[1] * dpp<kDown1>(V);
    P += c.ski[2] * dpp<kDown2>(V);
    P += c.ski[3] * dpp<kDown3>(V);
    // pu = f_u' V_xx :  pu[j] = sum_i b[i] V[i][j]   (sum down the rows, result in every row)
    T pu = c.bi * V;
    pu += dpp<kDown2>(pu);
    pu += dpp<kDown1>(pu);
    // Q_xx = l_xx + P f_x ; Q_ux = l_ux + pu f_x ; Q_x = l_x + f_x' V_x
    const T Qxx = c.lxx + contract_row(c.skj, P);
    const T Qux = c.vj[2] + contract_row(c.skj, pu);
    const T Qx = c.vj[1] + contract_row(c.skj, vx);
    // Q_uu = l_uu + pu f_u ; Q_u = l_u + f_u' V_x
    const T lu = dpp<dpp_quad(0, 0, 0, 0)>(c.vj[3]);
    const T luu = dpp<dpp_quad(1, 1, 1, 1)>(c.vj[3]);
    const T Quu = luu + quad_sum(pu * c.vj[0]);
    const T Qu = lu + quad_sum(c.vj[0] * vx);
    const T Qr = REG ? Quu + mu : Quu;
    pd = Qr > T(0);
    const T inv = T(1) / Qr;
    Kj = -(Qux * inv);   // K = -Q_uu^-1 Q_ux   (:109)
    kff = -(Qu * inv);   // k = -Q_uu^-1 Q_u    (:110)
    // Q_ux in "row form" (lane (i, j) <- Q_ux[i]) = the diagonal lane of each quad broadcast over it
    T Quxi = dpp_row<dpp_quad(0, 0, 0, 0), 0x1>(Qux, Qux);
    Quxi = dpp_row<dpp_quad(1, 1, 1, 1), 0x2>(Quxi, Qux);
    Quxi = dpp_row<dpp_quad(2, 2, 2, 2), 0x4>(Quxi, Qux);
    Quxi = dpp_row<dpp_quad(3, 3, 3, 3), 0x8>(Quxi, Qux);
    if constexpr (!REG) {
        // short form (:113-114): V_x = Q_x + K'Q_u ; V_xx = Q_xx + Q_ux' K
        V = Qxx + Quxi * Kj;
        vx = Qx + Kj * Qu;
    } else {
        // full update for a regularised gain
        const T Ki = -(Quxi * inv);
        V = Qxx + Ki * (Quu * Kj) + Ki * Qux + Quxi * Kj;
        vx = Qx + Kj * (Quu * kff + Qu) + Qux * kff;
    }
}

template <typename T, bool REG>
__global__ void __launch_bounds__(64) backward_tile16_kernel(KArgs<T> a) {
    constexpr int D = sizeof(T) == 4 ? 8 : 5;  // tiles in flight per lane (vmcnt holds 63 operations)
    constexpr int R = gain_record(4, 1);       // 8
    const int lane = threadIdx.x;
    const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3;
    const int gidx = blockIdx.x * 4 + (lane >> 4);
    const bool valid = gidx < a.B;
    const int b = valid ? gidx : a.B - 1;  // out-of-range groups shadow the last trajectory, never store
    const int st = a.status[b];
    const bool act = valid && traj_active(st);
    if (__ballot(act) == 0ull) return;
    const size_t B = a.B;
    const int N = a.N;
    T V = a.term[(size_t)(4 + l16) * B + b];
    T vx = a.term[(size_t)j * B + b];
    const T* __restrict__ lin = a.lin + (size_t)b * kTile16;
    const size_t tstride = B * kTile16;
    T* __restrict__ rec = a.gains + (size_t)b * R + (i == 0 ? j : 4);
    const bool storer = act && (i == 0 || l16 == 4);
    const size_t rstride = B * R;
    bool all_pd = true;

    auto do_step = [&](const Tile16<T>& c, int t) {
        T Kj, kff;
        bool pd;
        tile16_step<T, REG>(c, a.mu, V, vx, Kj, kff, pd);
        all_pd = all_pd && pd;
        if (storer) rec[(size_t)t * rstride] = (i == 0) ? Kj : kff;
    };

    int t = N - 1;
    // remainder steps first (no ring), so that the pipelined loop runs whole rings only
    for (int r = N % D; r > 0; --r, --t) {
        Tile16<T> c;
        tile16_load(c, lin + (size_t)t * tstride, i, j, l16);
        do_step(c, t);
    }
    if (t >= 0) {
        Tile16<T> ring[D];
#pragma unroll
        for (int u = 0; u < D; ++u) tile16_load(ring[u], lin + (size_t)(t - u) * tstride, i, j, l16);
        for (; t >= 0; t -= D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                // consume ring slot u, then refill the SAME registers with the tile D steps ahead (their
                // old contents are dead by then, so no copies and no full drain at the loop edge).  The
                // refill is unconditional (clamped to tile 0 at the end of the sweep): the body stays
                // branch-free and D-1 tiles per lane stay in flight.
                do_step(ring[u], t - u);
                const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                tile16_load(ring[u], lin + (size_t)tn * tstride, i, j, l16);
            }
        }
    }
    if (act && l16 == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

// dense ILQR_LIN order from a tile (debug / tests): e in [0, 46)
ILQR_DEV int tile16_index_of(int e) {
    if (e < 16) { const int i = e >> 2, j = e & 3; return 4 * j + ((i - j + 4) & 3); }  // f_x[i][j]
    if (e < 20) return 32 + 4 * (e - 16);           // f_u[i]
    if (e < 24) return 32 + 4 * (e - 20) + 1;       // l_x[i]
    if (e == 24) return 35;                          // l_u
    if (e < 41) return 16 + (e - 25);                // l_xx
    if (e < 45) return 32 + 4 * (e - 41) + 2;       // l_ux[j]
    return 39;                                       // l_uu
}

template <typename T>
__global__ void tile16_gather_dense_kernel(T* dense, const T* lin, int B, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * 46) return;
    const int e = (int)(idx % 46);
    const int t = (int)((idx / 46) % N);
    const int b = (int)(idx / ((size_t)46 * N));
    dense[idx] = lin[((size_t)t * B + b) * kTile16 + tile16_index_of(e)];
}

}  // namespace ilqr

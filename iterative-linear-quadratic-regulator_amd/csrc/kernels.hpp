// kernels.hpp -- HIP kernels of the batched iLQR hot path (gfx950 / CDNA4).
//
// HBM data layout (scalar type T):
//   X    [n_slots][N+1][B][n_x]      U    [n_slots][N][B][n_u]     one state / control VECTOR per (t, b): a lane reads or
//                    writes it with one 16-byte access (n_x = 4, f32) where the batch-innermost form needed n_x
//                    4-byte ones -- a vector-memory instruction costs a lone wave ~13 cycles of issue whatever its
//                    width (tools/micro/issue_rate.hip), and every store sits in the same in-order vmcnt queue as
//                    the rollout's prefetched loads
//   gains [N][B][R]  one record per (t, b): K_t (n_u*n_x, row-major) then k_t (n_u) = U_ff of the
//                    reference, R = n_u*n_x + n_u rounded up to a multiple of 4 scalars (16-B loads)
//   lin  generic:    [N][E][B]   E = 2n^2+2nm+n+m+m^2, per step f_x f_u l_x l_u l_xx l_ux l_uu
//        n=4, m=1:   [N][B][48]  one 192-B (f32) tile per (t, b), packed for the 16-lane sweep below
//   term [n + n^2][B]                (l_f_x, l_f_xx)
//   costs [n_alpha][B]   cost, cost_prev, alpha_taken [B]   status, iters, accepted, cur_slot [B]
// 64 consecutive trajectories are one contiguous 64 * n_x * sizeof(T) block per wave-instruction in X / U and one
// 256-B (f32) / 512-B (f64) row in the per-trajectory vectors (x0, costs, status ...).
//
// "Slots": the current trajectory of b lives in slot cur_slot[b]; candidate a of a
// line-search pass is rolled out into slot (cur_slot[b] + 1 + a) % n_slots, and accepting
// a candidate is just cur_slot[b] <- that slot.  The next linearize (which reads every point
// anyway) moves the accepted trajectories back to slot 0, so rollouts stay coalesced.
#pragma once
#include "dynamics.hpp"

namespace ilqr {

// cache-policy switches (compile-time; A/B'd with tools/build_variants.sh, results in DESIGN.md)
#ifndef ILQR_NT_TILE_STORE
// Non-temporal tile stores paid while the rollout's accesses were scattered over the slots (forward 216 -> 191 us):
// the one-shot tile stream no longer evicted its re-read working set.  With the canonical slot 0 (linearize_kernel)
// that working set is small and coalesced, and ordinary stores win: the sweep finds most of the 157 MB tile tensor in
// the 256 MB Infinity Cache (42.8 -> 38.8 us f32, 95.9 -> 89.1 us f64; whole step 0.226 -> 0.223 ms).
#define ILQR_NT_TILE_STORE 0
#endif
#ifndef ILQR_NT_TILE_LOAD
#define ILQR_NT_TILE_LOAD 0
#endif
#ifndef ILQR_DROP_ALL
#define ILQR_DROP_ALL 0   // experiment switch: branch-free dropped stores in the fp64 kernels too
#endif
constexpr int kMaxAlpha = 16;
constexpr int kCounterRing = 64;

template <typename T> struct KArgs {
    int B, N, n_slots, integ, maxiter, flags;
    int n_pass;      // alphas in this pass
    int last_pass;   // select: this is the last pass of the iteration
    int init_mode;   // head of a solve (iLQR_class.py:257-259): the rollout takes every trajectory whatever its previous status
                     // and clears the active-count slot, select accepts candidate 0 unconditionally
    int counter_idx; // select: which ring counter receives the number of still-active trajectories
    T dt, tol, mu;
    T alphas[kMaxAlpha];
    T* X; T* U; int* cur_slot;
    T* gains; T* lin; T* term; T* x0;
    T* costs; T* cost; T* cost_prev; T* alpha_taken;
    int* status; int* iters; int* accepted; int* counters;
    int reset_slots;   // backward kernels: set cur_slot[b] = 0 for active b (after a canonicalising linearize)
    int const_lin;     // the expansion's matrices are the same at every (t, b) (a Linear system with the built-in quadratic
                       // cost: only l_x, l_u vary); set by the host for the library's own linearisation, never for caller tensors
    int fuse_select;   // backward_fused16_kernel: run the acceptance step of the previous iteration's candidates first
    int lin_sparse;    // the expansion of a constant-matrix system in its sparse form: the matrices in the records of t = N-1 only
                       // (what the CONST sweep reads), l_x, l_u of every (t, b) as a dense [N][B][n_x + n_u] tensor at the front of `lin`
    int t_first;       // linearize_wave_kernel: first time step of the launch (0, or N-1 for the sparse form)
    const T* params;
    long long* probe;  // diagnostic: {shader cycles, 100 MHz ticks} of block 0 per kernel, or nullptr
};

constexpr int gain_record(int nx, int nu) { return ((nu * nx + nu) + 3) / 4 * 4; }

ILQR_DEV bool traj_active(int status) { return (status & 0xff) == ILQR_TRAJ_ACTIVE; }

// element offset of vector (slot, t, b) in X (Tn = N + 1, C = n_x) or U (Tn = N, C = n_u)
ILQR_DEV size_t vec_at(size_t B, int Tn, int C, int slot, int t, size_t b) {
    return ((((size_t)slot * Tn + t) * B) + b) * C;
}
// a C-vector of T at p (aligned to the largest power of two dividing C * sizeof(T), at most 16): 16-, 8- or 4-byte pieces
typedef unsigned int vec_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int vec_u2 __attribute__((ext_vector_type(2)));
constexpr int vec_piece(int bytes) { return bytes % 16 == 0 ? 16 : bytes % 8 == 0 ? 8 : 4; }
constexpr int vec_pieces(int bytes) { return bytes / vec_piece(bytes); }
template <typename T, int C> ILQR_DEV void vec_load(const T* __restrict__ p, T* o) {
    constexpr int BYTES = C * (int)sizeof(T), PB = vec_piece(BYTES), NP = BYTES / PB;
    if constexpr (PB == 16) {
        vec_u4 v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) v[k] = reinterpret_cast<const vec_u4*>(p)[k];
        __builtin_memcpy(o, v, BYTES);
    } else if constexpr (PB == 8) {
        vec_u2 v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) v[k] = reinterpret_cast<const vec_u2*>(p)[k];
        __builtin_memcpy(o, v, BYTES);
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) o[i] = p[i];
    }
}
template <typename T, int C> ILQR_DEV void vec_store(T* __restrict__ p, const T* v) {
    constexpr int BYTES = C * (int)sizeof(T), PB = vec_piece(BYTES), NP = BYTES / PB;
    if constexpr (PB == 16) {
        vec_u4 w[NP];
        __builtin_memcpy(w, v, BYTES);
#pragma unroll
        for (int k = 0; k < NP; ++k) reinterpret_cast<vec_u4*>(p)[k] = w[k];
    } else if constexpr (PB == 8) {
        vec_u2 w[NP];
        __builtin_memcpy(w, v, BYTES);
#pragma unroll
        for (int k = 0; k < NP; ++k) reinterpret_cast<vec_u2*>(p)[k] = w[k];
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) p[i] = v[i];
    }
}

// Clock probe (diagnostic only; the values go to a buffer nothing else reads): one pair of stamps around
// a whole kernel body for lane 0 of workgroup 0 gives cycles per step and the clock the chip holds.
struct ClockProbe {
    long long c0, r0;
    ILQR_DEV void start() { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    // slot 0: backward (from probe[8]), slot 1: forward (from probe[8 + 4*65536]); per workgroup 4 longs:
    // {start tick, end tick, cycles, HW_ID}.  probe[2*slot], [2*slot+1] keep workgroup 0's {cycles, ticks}.
    ILQR_DEV void stop(long long* probe, int slot) const {
        if (probe && threadIdx.x == 0) {
            const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            if (wg == 0) { probe[2 * slot] = c1 - c0; probe[2 * slot + 1] = r1 - r0; }
            if (wg < 65536) {
                long long* q = probe + 8 + ((size_t)slot * 65536 + wg) * 4;
                q[0] = r0; q[1] = r1; q[2] = c1 - c0;
                q[3] = (long long)__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11)) |
                       ((long long)__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) << 32);
            }
        }
    }
};

}  // namespace ilqr
#include "backward_tile16.hpp"
#include "backward_tile16m2.hpp"
namespace ilqr {

// ---------------------------------------------------------------------------
// The expansion of one (b, t) point packed into the tile the DPP sweeps read: backward_tile16.hpp (n_u = 1: 48
// scalars = 12 V4) or backward_tile16m2.hpp (n_x = 4, n_u = 2: 64 scalars = 16 V4).  fx, fu: the discrete Jacobians
// (Stepper::step_jac); the cost derivatives are evaluated here.  n_x < 4 is zero-padded to the 4 x 4 tile (padding
// states have no dynamics and no cost).  Shared by linearize_kernel and backward_fused16_kernel.
// ---------------------------------------------------------------------------
// tile16_fill<TS>: the packing proper, from the expansion's terms in a scalar type T (T = TS, pick = identity) or in the
// float pair of the two-points-per-lane producers (pick = one half).
template <typename TS, int NX, int NU, typename T, typename Pick>
ILQR_DEV void tile16_fill(Pick pick, const T (*fx)[NX], const T (*fu)[NU], const T* gxn, const T* gu1, const T (*lxxn)[NX],
                          const T (*luxn)[NX], const T (*luu)[NU], typename Vec4<TS>::type* tile) {
    static_assert((NX >= 2 && NX <= 4 && NU == 1) || (NX == 4 && NU == 2), "tile packing: n_x <= 4 with n_u = 1, or (4, 2)");
    // zero-pad to the 4 x 4 tile (a no-op for n_x = 4): padding states have no dynamics and no cost
    auto F = [&](int i, int j) -> TS { return (i < NX && j < NX) ? pick(fx[i < NX ? i : 0][j < NX ? j : 0]) : TS(0); };
    auto L = [&](int i, int j) -> TS { return (i < NX && j < NX) ? pick(lxxn[i < NX ? i : 0][j < NX ? j : 0]) : TS(0); };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        tile[c].x = F((c + 0) & 3, c); tile[c].y = F((c + 1) & 3, c);
        tile[c].z = F((c + 2) & 3, c); tile[c].w = F((c + 3) & 3, c);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tile[4 + i].x = L(i, 0); tile[4 + i].y = L(i, 1);
        tile[4 + i].z = L(i, 2); tile[4 + i].w = L(i, 3);
    }
    if constexpr (NU == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = j < NX ? j : 0;
            tile[8 + j].x = j < NX ? pick(fu[jj][0]) : TS(0);
            tile[8 + j].y = j < NX ? pick(gxn[jj]) : TS(0);
            tile[8 + j].z = j < NX ? pick(luxn[0][jj]) : TS(0);
            tile[8 + j].w = (j == 0) ? pick(gu1[0]) : ((j == 1) ? pick(luu[0][0]) : TS(0));
        }
    } else {
        // group j: f_u[j][0], f_u[j][1], l_x[j], l_ux[0][j] | l_ux[1][j], e0, e1, e2   (backward_tile16m2.hpp);
        // the sweep's Q_uu is symmetric: the mean of l_uu[0][1] and l_uu[1][0] is stored
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[8 + 2 * j].x = pick(fu[j][0]); tile[8 + 2 * j].y = pick(fu[j][NU - 1]);
            tile[8 + 2 * j].z = pick(gxn[j]); tile[8 + 2 * j].w = pick(luxn[0][j]);
            tile[9 + 2 * j].x = pick(luxn[NU - 1][j]);
            tile[9 + 2 * j].y = (j == 0) ? pick(gu1[0]) : ((j == 1) ? TS(0.5) * (pick(luu[0][NU - 1]) + pick(luu[NU - 1][0])) : TS(0));
            tile[9 + 2 * j].z = (j == 0) ? pick(gu1[NU - 1]) : ((j == 1) ? pick(luu[NU - 1][NU - 1]) : TS(0));
            tile[9 + 2 * j].w = (j == 0) ? pick(luu[0][0]) : TS(0);
        }
    }
}

template <typename T, typename Dyn, typename P>
ILQR_DEV void tile16_pack(P p, T dt, const T* x, const T* u, const T (*fx)[Dyn::NX],
                          const T (*fu)[Dyn::NU], typename Vec4<T>::type* tile) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    T gxn[NX], gu1[NU], lxxn[NX][NX], luxn[NU][NX], luu[NU][NU];
    Cost<T, Dyn>::grad(p, dt, x, u, gxn, gu1);
    Cost<T, Dyn>::hess(p, dt, x, u, lxxn, luxn, luu);
    tile16_fill<T, NX, NU>([](T v) { return v; }, fx, fu, gxn, gu1, lxxn, luxn, luu, tile);
}

// ---------------------------------------------------------------------------
// linearize: one lane per (b, t) point, t in [0, N]; t == N is the terminal
// expansion.  Replaces iLQR._get_all_derivatives_for_backward_pass
// (iLQR_class.py:318-331) + l_f_x / l_f_xx (:136-138), hoisted out of the
// sequential scan because it does not depend on the carry.
// ---------------------------------------------------------------------------
template <typename T, typename Dyn, bool TILE16, int INTEG>
__global__ void __launch_bounds__(TILE16 ? 64 : 256) linearize_kernel(KArgs<T> a) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t B = a.B;
    const int t = (int)(idx / B);
    const int b = (int)(idx % B);
    const bool inr = t <= a.N;                                  // a (b, t) point of the batch
    // status and slot are fetched together (the slot of a finished trajectory is simply not used): one memory
    // round trip at the head of the wave instead of two dependent ones
    const int st_raw = inr ? a.status[b] : 0;
    const int slot_raw = inr ? a.cur_slot[b] : 0;
    const bool live = inr && traj_active(st_raw);
    // the parameter block is read before the kernel's first store: hipcc then uses scalar loads (SGPRs); reads
    // that follow a store it cannot disambiguate become per-lane vector loads with their own vmcnt waits
    T p[PL::TOTAL];
#pragma unroll
    for (int q = 0; q < PL::TOTAL; ++q) p[q] = a.params[q];
    T x[NX], u[NU];
    // Canonicalisation: accepting a candidate only moves cur_slot[b] (no copy), so after a few iterations
    // neighbouring trajectories live in different slots and every wave-wide access of the rollout -- 64
    // consecutive b -- scatters over up to n_slots rows (measured: rollout 125 -> 200 us once the slots have
    // decorrelated).  This kernel reads every point of the active trajectories anyway: it also writes them into
    // slot 0 (16 MB at the c3 shape).  cur_slot[b] itself may only change once every block has read it, so that
    // is left to the kernel that always follows (the backward sweep, KArgs::reset_slots; SolverT::fix_slots for
    // any other order); the old slot keeps a valid copy until the next rollout overwrites it.
    const int slot = live ? slot_raw : 0;
    const bool move = live && slot != 0;
    const int tt = live ? t : 0;
    const int bb = live ? b : 0;
    if (live) {
        vec_load<T, NX>(a.X + vec_at(B, a.N + 1, NX, slot, tt, bb), x);
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = T(0);
    }
    if (move) vec_store<T, NX>(a.X + vec_at(B, a.N + 1, NX, 0, tt, bb), x);
    const bool has_u = inr && t < a.N;
    const int tu = has_u ? t : 0;
    if (has_u && live) {
        vec_load<T, NU>(a.U + vec_at(B, a.N, NU, slot, tu, bb), u);
    } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = T(0);
    }
    if (has_u && move) vec_store<T, NU>(a.U + vec_at(B, a.N, NU, 0, tu, bb), u);
    if constexpr (!TILE16) {
        if (!live) return;
    }
    if (live && t == a.N) {
        T g[NX], H[NX][NX];
        Cost<T, Dyn>::l_f_x(p, x, g);
        Cost<T, Dyn>::l_f_xx(p, x, H);
        if constexpr (TILE16) {
            // the DPP sweep reads the terminal expansion on its 4 x 4 tile: [V_x (4) | V_xx (4 x 4)], zero-padded
#pragma unroll
            for (int i = 0; i < 4; ++i) a.term[(size_t)i * B + b] = i < NX ? g[i < NX ? i : 0] : T(0);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                a.term[(size_t)(4 + i) * B + b] = (i / 4 < NX && i % 4 < NX) ? H[i / 4 < NX ? i / 4 : 0][i % 4 < NX ? i % 4 : 0] : T(0);
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) a.term[(size_t)i * B + b] = g[i];
#pragma unroll
            for (int i = 0; i < NX * NX; ++i) a.term[(size_t)(NX + i) * B + b] = H[i / NX][i % NX];
            return;
        }
    }
    const bool point = live && t < a.N;   // this lane produces an expansion record
    T xn[NX], fx[NX][NX], fu[NX][NU];
    Stepper<T, Dyn>::step_jac(INTEG, p, a.dt, x, u, xn, fx, fu);  // integrator folded at compile time
    if constexpr (TILE16) {
        // pack the expansion into the tile of backward_tile16.hpp (n_u = 1: 48 scalars) or backward_tile16m2.hpp
        // (n_x = 4, n_u = 2: 64 scalars).  The 64 tiles of a wave are contiguous in HBM (tile index = t*B + b = this
        // lane's global index), but each lane holds ITS tile: a direct store would be TV x 16-B pieces at a
        // 192..512-B lane stride (measured 1.34x write amplification).  So the wave transposes through LDS in chunks
        // and writes 16 B per lane to consecutive addresses.
        using V4 = typename Vec4<T>::type;
        constexpr int TV = NU == 1 ? 12 : 16;          // V4 per tile
        V4 tile[TV];
        tile16_pack<T, Dyn>(p, a.dt, x, u, fx, fu, tile);
        // two passes over groups of 32 whole tiles (so every pass writes one contiguous run of full cache lines).
        // One pass of 64 tiles in f32 needs 13 KB of LDS per wave and caps the kernel at 3 waves per SIMD; with
        // 6.6 KB it runs 4 (the VGPR limit) and hides more of its gather / store latency: 50 -> 46 us.
        constexpr int PASSES = 2, TPP = 64 / PASSES, ROW = TV + 1;   // TV V4 per tile + 1 pad
        __shared__ V4 xpose[TPP * ROW];
        const int lane = threadIdx.x;
        const unsigned long long okmask = __ballot(point);
        V4* gout = reinterpret_cast<V4*>(a.lin) + (size_t)blockIdx.x * 64 * TV;   // first tile of this wave
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if (ps) __syncthreads();
            if (lane / TPP == ps) {
#pragma unroll
                for (int q = 0; q < TV; ++q) xpose[(lane % TPP) * ROW + q] = tile[q];
            }
            __syncthreads();
            // read back in 16-byte units, tile-major: one wave-instruction = 64 lanes x 16 B = 1 KiB of
            // consecutive addresses, whatever the scalar type (a 32-B f64 V4 split over two instructions
            // would leave every instruction writing half lines, which non-temporal stores punish)
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            constexpr int UPT = TV * (int)sizeof(T) / 4;           // 16-B units per tile: 12 / 24 (16 / 32)
            constexpr int UPV = (int)sizeof(T) / 4;                // units per V4: 1 / 2
            const u4* xs = reinterpret_cast<const u4*>(xpose);
            u4* gu = reinterpret_cast<u4*>(gout);
#pragma unroll
            for (int r = 0; r < TPP * UPT / 64; ++r) {
                const int w = lane + 64 * r;          // w-th unit of this pass
                const int kl = w / UPT, q = w % UPT, k = ps * TPP + kl;
                if ((okmask >> k) & 1ull) {
                    const u4 val = xs[kl * ROW * UPV + q];
                    u4* dst = &gu[(size_t)k * UPT + q];
                    if (ILQR_NT_TILE_STORE) __builtin_nontemporal_store(val, dst);
                    else *dst = val;
                }
            }
        }
        return;
    }
    T* out = a.lin + ((size_t)t * E) * B + b;
    int e = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) out[(size_t)(e++) * B] = fx[i][j];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) out[(size_t)(e++) * B] = fu[i][j];
    T g[NX], gu[NU], lxx[NX][NX], lux[NU][NX], luu[NU][NU];
    Cost<T, Dyn>::grad(p, a.dt, x, u, g, gu);
    Cost<T, Dyn>::hess(p, a.dt, x, u, lxx, lux, luu);
#pragma unroll
    for (int i = 0; i < NX; ++i) out[(size_t)(e++) * B] = g[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) out[(size_t)(e++) * B] = gu[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) out[(size_t)(e++) * B] = lxx[i / NX][i % NX];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) out[(size_t)(e++) * B] = lux[i / NX][i % NX];
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) out[(size_t)(e++) * B] = luu[i / NU][i % NU];
}

// after a canonicalising linearize that is NOT followed by a backward sweep (stage API used out of order)
template <typename T>
__global__ void reset_slots_kernel(KArgs<T> a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < a.B && traj_active(a.status[b])) a.cur_slot[b] = 0;
}

// ---------------------------------------------------------------------------
// Gain solve: [K | k] = -Quu^-1 [Qux | Qu].  Cholesky first (north_star); if Quu
// is not positive definite fall back to LU with partial pivoting, which is what
// the reference's jnp.linalg.solve does unconditionally (iLQR_class.py:109-110).
// Returns false when the fallback was taken.
// ---------------------------------------------------------------------------
template <typename T, int NX, int NU>
ILQR_DEV bool gain_solve(const T (*Quu)[NU], const T (*Qux)[NX], const T* Qu, T (*K)[NX], T* k) {
    if constexpr (NU == 1) {
        const T inv = T(1) / Quu[0][0];
#pragma unroll
        for (int j = 0; j < NX; ++j) K[0][j] = -(Qux[0][j] * inv);
        k[0] = -(Qu[0] * inv);
        return Quu[0][0] > T(0);
    } else {
        T Lc[NU][NU];
        bool pd = true;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            T d = Quu[j][j];
#pragma unroll
            for (int s = 0; s < j; ++s) d -= Lc[j][s] * Lc[j][s];
            pd = pd && (d > T(0));
            const T ljj = M<T>::sqrt(d);
            Lc[j][j] = ljj;
            const T inv = T(1) / ljj;
#pragma unroll
            for (int i = j + 1; i < NU; ++i) {
                // lower triangle of Quu (the reference's Quu is symmetric up to rounding)
                T v = Quu[i][j];
#pragma unroll
                for (int s = 0; s < j; ++s) v -= Lc[i][s] * Lc[j][s];
                Lc[i][j] = v * inv;
            }
        }
        T rhs[NU][NX + 1];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) rhs[i][j] = Qux[i][j];
            rhs[i][NX] = Qu[i];
        }
        if (pd) {
            // L y = rhs ; L' z = y
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const T inv = T(1) / Lc[i][i];
#pragma unroll
                for (int c = 0; c <= NX; ++c) {
                    T v = rhs[i][c];
#pragma unroll
                    for (int s = 0; s < i; ++s) v -= Lc[i][s] * rhs[s][c];
                    rhs[i][c] = v * inv;
                }
            }
#pragma unroll
            for (int i = NU - 1; i >= 0; --i) {
                const T inv = T(1) / Lc[i][i];
#pragma unroll
                for (int c = 0; c <= NX; ++c) {
                    T v = rhs[i][c];
#pragma unroll
                    for (int s = i + 1; s < NU; ++s) v -= Lc[s][i] * rhs[s][c];
                    rhs[i][c] = v * inv;
                }
            }
        } else {
            T Ac[NU][NU];
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int j = 0; j < NU; ++j) Ac[i][j] = Quu[i][j];
            lu_solve_inplace<T, NU, NX + 1>(Ac, rhs);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) K[i][j] = -rhs[i][j];
            k[i] = -rhs[i][NX];
        }
        return pd;
    }
}

// One Riccati step on register tiles (iLQR_class.py:100-114).  tile = the E scalars
// of one (b, t) in ILQR_LIN order.
template <typename T, int NX, int NU>
ILQR_DEV bool riccati_step(const T* tile, T mu, T* Vx, T (*Vxx)[NX], T (*K)[NX], T* k) {
    constexpr int oFX = 0, oFU = NX * NX, oLX = oFU + NX * NU, oLU = oLX + NX, oLXX = oLU + NU,
                  oLUX = oLXX + NX * NX, oLUU = oLUX + NU * NX;
    T Qx[NX], Qu[NU], P[NX][NX], Pu[NU][NX], Qxx[NX][NX], Qux[NU][NX], Quu[NU][NU];
    // Q_x = l_x + f_x' V_x ; Q_u = l_u + f_u' V_x
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) acc += tile[oFX + i * NX + j] * Vx[i];
        Qx[j] = tile[oLX + j] + acc;
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) acc += tile[oFU + i * NU + j] * Vx[i];
        Qu[j] = tile[oLU + j] + acc;
    }
    // P = f_x' V_xx ; Pu = f_u' V_xx   (left-associated like `f_x.T @ V_xx @ f_x`)
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += tile[oFX + s * NX + i] * Vxx[s][j];
            P[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += tile[oFU + s * NU + i] * Vxx[s][j];
            Pu[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += P[i][s] * tile[oFX + s * NX + j];
            Qxx[i][j] = tile[oLXX + i * NX + j] + acc;
        }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += Pu[i][s] * tile[oFX + s * NX + j];
            Qux[i][j] = tile[oLUX + i * NX + j] + acc;
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NX; ++s) acc += Pu[i][s] * tile[oFU + s * NU + j];
            Quu[i][j] = tile[oLUU + i * NU + j] + acc;
        }
    }
    T Qr[NU][NU];
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) Qr[i][j] = Quu[i][j] + ((i == j) ? mu : T(0));
    const bool pd = gain_solve<T, NX, NU>(Qr, Qux, Qu, K, k);
    if (mu == T(0)) {
        // short form (iLQR_class.py:113-114): V_x = Q_x + K'Q_u ; V_xx = Q_xx + Q_ux' K
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NU; ++s) acc += K[s][i] * Qu[s];
            Vx[i] = Qx[i] + acc;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T acc2 = T(0);
#pragma unroll
                for (int s = 0; s < NU; ++s) acc2 += Qux[s][i] * K[s][j];
                Vxx[i][j] = Qxx[i][j] + acc2;
            }
        }
    } else {
        // full update, exact for a regularised gain:
        // V_x = Q_x + K'Quu k + K'Q_u + Q_ux'k ; V_xx = Q_xx + K'Quu K + K'Q_ux + Q_ux'K
        T QK[NU][NX], Qk[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NU; ++s) acc += Quu[i][s] * k[s];
            Qk[i] = acc;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T acc2 = T(0);
#pragma unroll
                for (int s = 0; s < NU; ++s) acc2 += Quu[i][s] * K[s][j];
                QK[i][j] = acc2;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < NU; ++s) acc += K[s][i] * (Qk[s] + Qu[s]) + Qux[s][i] * k[s];
            Vx[i] = Qx[i] + acc;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                T acc2 = T(0);
#pragma unroll
                for (int s = 0; s < NU; ++s)
                    acc2 += K[s][i] * (QK[s][j] + Qux[s][j]) + Qux[s][i] * K[s][j];
                Vxx[i][j] = Qxx[i][j] + acc2;
            }
        }
    }
    return pd;
}

// ---------------------------------------------------------------------------
// backward sweep, generic form: one lane per trajectory, reverse loop over t with
// the next tile prefetched into registers while the current step computes.
// Replaces iLQR._backward_pass_scan (iLQR_class.py:122-161).
// ---------------------------------------------------------------------------
template <typename T, int NX, int NU>
__global__ void __launch_bounds__(64) backward_lane_kernel(KArgs<T> a) {
    constexpr int E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int st = a.status[b];
    if (!traj_active(st)) return;
    if (a.reset_slots) a.cur_slot[b] = 0;   // linearize moved this trajectory into slot 0 (see linearize_kernel)
    const size_t B = a.B;
    T Vx[NX], Vxx[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) Vx[i] = a.term[(size_t)i * B + b];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) Vxx[i][j] = a.term[(size_t)(NX + i * NX + j) * B + b];
    T cur[E], nxt[E];
    {
        const T* src = a.lin + ((size_t)(a.N - 1) * E) * B + b;
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = src[(size_t)e * B];
    }
    bool all_pd = true;
    for (int t = a.N - 1; t >= 0; --t) {
        if (t > 0) {
            const T* src = a.lin + ((size_t)(t - 1) * E) * B + b;
#pragma unroll
            for (int e = 0; e < E; ++e) nxt[e] = src[(size_t)e * B];
        }
        T K[NU][NX], k[NU];
        all_pd = riccati_step<T, NX, NU>(cur, a.mu, Vx, Vxx, K, k) && all_pd;
        constexpr int R = gain_record(NX, NU);
        T* rec = a.gains + ((size_t)t * B + b) * R;
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) rec[i * NX + j] = K[i][j];
#pragma unroll
        for (int i = 0; i < NU; ++i) rec[NU * NX + i] = k[i];
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e];
    }
    if (!all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

// ---------------------------------------------------------------------------
// forward rollout: one lane per (trajectory, alpha) candidate; blockIdx.y = alpha
// index.  Replaces iLQR._forward_pass_scan (iLQR_class.py:193-247), all trial
// alphas of the backtracking loop (:279-302) at once.
// ---------------------------------------------------------------------------
template <typename T, typename Dyn, int INTEG>
__global__ void __launch_bounds__(64) forward_kernel(KArgs<T> a) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    constexpr int R = gain_record(NX, NU);
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int ai = blockIdx.y;
    if (a.init_mode && blockIdx.x == 0 && ai == 0 && threadIdx.x == 0) a.counters[a.counter_idx] = 0;   // the select that follows counts into it
    if (b >= a.B) return;
    if (!a.init_mode) {
        if (!traj_active(a.status[b])) return;
        if (a.accepted[b]) return;  // an earlier pass of this iteration already found its alpha
    }
    const size_t B = a.B;
    const int N = a.N;
    const int slot = a.cur_slot[b];
    const int cslot = (slot + 1 + ai) % a.n_slots;
    const T alpha = a.alphas[ai];
    const T* __restrict__ p = a.params;
    T x[NX], u[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = a.x0[(size_t)i * B + b];
    ClockProbe cp;
    cp.start();
    T cost = T(0);
    // (slot, 0, b); one time step further is B * NX (B * NU) scalars on
    const T* Xo = a.X + vec_at(B, N + 1, NX, slot, 0, b);
    const T* Uo = a.U + vec_at(B, N, NU, slot, 0, b);
    const T* G = a.gains + (size_t)b * R;
    T* Xc = a.X + vec_at(B, N + 1, NX, cslot, 0, b);
    T* Uc = a.U + vec_at(B, N, NU, cslot, 0, b);
    const size_t sX = B * NX, sU = B * NU;
    // The per-step inputs (x_old, u_old, K, k) do not depend on the carried state, so step t+1's are
    // requested before step t's arithmetic starts: their latency hides under one RK4 step.
    // (for big gain records, n_x > 4, the double buffer would not fit the register file: load in place)
    constexpr bool PREFETCH = (R <= 32);
    constexpr int RN = PREFETCH ? R : 1, NXN = PREFETCH ? NX : 1, NUN = PREFETCH ? NU : 1;
    T xo[NX], uo[NU], g[R], xo_n[NXN], uo_n[NUN], g_n[RN];
    vec_load<T, NX>(Xo, xo);
    vec_load<T, NU>(Uo, uo);
#pragma unroll
    for (int r = 0; r < R; ++r) g[r] = G[r];
    for (int t = 0; t < N; ++t) {
        const int tn = (t + 1 < N) ? t + 1 : t;
        if constexpr (PREFETCH) {
            vec_load<T, NX>(Xo + tn * sX, xo_n);
            vec_load<T, NU>(Uo + tn * sU, uo_n);
#pragma unroll
            for (int r = 0; r < R; ++r) g_n[r] = G[(size_t)tn * B * R + r];
        }
        T dx[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = x[i] - xo[i];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            T fb = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) fb += g[j * NX + i] * dx[i];
            // u = u_old + alpha * k + K (x - x_old)   (iLQR_class.py:181-182)
            u[j] = uo[j] + alpha * g[NU * NX + j] + fb;
        }
        vec_store<T, NX>(Xc + t * sX, x);
        vec_store<T, NU>(Uc + t * sU, u);
        cost += Cost<T, Dyn>::stage(p, a.dt, x, u);
        T xn[NX];
        Stepper<T, Dyn>::step(INTEG, p, a.dt, x, u, xn);  // integrator folded at compile time
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xo[i] = xo_n[i];
#pragma unroll
            for (int j = 0; j < NU; ++j) uo[j] = uo_n[j];
#pragma unroll
            for (int r = 0; r < R; ++r) g[r] = g_n[r];
        } else {
            vec_load<T, NX>(Xo + tn * sX, xo);
            vec_load<T, NU>(Uo + tn * sU, uo);
#pragma unroll
            for (int r = 0; r < R; ++r) g[r] = G[(size_t)tn * B * R + r];
        }
    }
    vec_store<T, NX>(Xc + N * sX, x);
    cost += Cost<T, Dyn>::terminal(p, x);
    a.costs[(size_t)ai * B + b] = cost;
    cp.stop(a.probe, 1);
}

// ---------------------------------------------------------------------------
// forward rollout, ring form (small systems, tensors < 2 GiB): identical arithmetic to forward_kernel, but
// the per-step inputs (x_old, u_old, gain record) are kept PF steps ahead in registers by inline-asm buffer
// loads with self-counted vmcnt (same technique and same reasons as RawTile in backward_tile16.hpp: hipcc
// drains every load it can see at the loop edge, and one RK4 step is shorter than an HBM round trip under
// load).  Per step the wave issues NLD asm loads and NX + NU compiler stores (the candidate trajectory), all
// retiring in issue order, so "slot q has landed" == at most (PF-1) * (NLD + NX + NU) younger operations.
// ---------------------------------------------------------------------------
#include "fwd_in_gen.inc"

// The rollout of candidate ai of trajectory b by this lane; in_range = the lane has a candidate at all.  The lanes of a
// wave may hold any mix of (b, ai) -- forward_ring_kernel gives a wave 64 neighbouring trajectories of one alpha, the
// persistent kernel (persistent.hpp) all candidates of a workgroup's trajectories -- as long as the whole wave calls it.
template <typename T, typename Dyn, int INTEG>
ILQR_DEV void rollout_ring(const KArgs<T>& a, int b, int ai, bool in_range, bool force_init = false) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    using In = FwdIn<T, NX, NU>;
    // stores per step: the pieces of the state and of the control vector (one 16-byte store for n_x = 4 in fp32)
    constexpr int R = In::R, NLD = In::NLD, NST = vec_pieces(NX * (int)sizeof(T)) + vec_pieces(NU * (int)sizeof(T));
    // ring depth: (PF-1)*(NLD+NST) <= 63 (the vmcnt field), at most ILQR_RING_PF_MAX, and at most ~130 VGPRs of ring so
    // the rollout arithmetic still fits the 256 without spilling (see csrc/check_ring_kernels.py)
#ifndef ILQR_RING_PF_MAX
#define ILQR_RING_PF_MAX 6
#endif
    constexpr int SLOT_REGS = (NX + NU + R) * (int)sizeof(T) / 4;
    constexpr int PF_CNT = (63 / (NLD + NST)) + 1 > ILQR_RING_PF_MAX ? ILQR_RING_PF_MAX : (63 / (NLD + NST)) + 1;
    // (the fp64 backward-Euler step -- Newton loop with an LU solve -- needs more registers of its own: one slot fewer)
    constexpr int RING_CAP = (sizeof(T) == 8 && INTEG == ILQR_INT_BACKWARD_EULER) ? 104 : 130;
    constexpr int PF = PF_CNT * SLOT_REGS > RING_CAP ? RING_CAP / SLOT_REGS : PF_CNT;
    static_assert(PF >= 2, "ring too shallow to be worth it");
    // status, accepted flag and slot in one memory round trip (bitwise &: no short-circuit between the loads)
    const int bc = in_range ? b : 0;
    ai = in_range ? ai : 0;
    const int st_raw = a.status[bc], acc_raw = a.accepted[bc], slot_raw = a.cur_slot[bc];
    // force_init: the head of a solve (every trajectory, alpha = 0) whatever the argument block says -- the persistent kernel
    // runs it with the block of its iterations
    const bool live = in_range & ((a.init_mode != 0) | force_init | (traj_active(st_raw) & (acc_raw == 0)));
    if (__ballot(live) == 0ull) return;
    const int bb = live ? b : 0;          // dead lanes shadow trajectory 0 and never store
    const size_t B = a.B;
    const int N = a.N;
    const int slot = slot_raw;            // (a dead lane's slot only selects which valid rows it reads and discards)
    const int cslot = (slot + 1 + ai) % a.n_slots;
    const T alpha = force_init ? T(0) : a.alphas[ai];
    // (a local copy: when the argument block is memory -- the persistent kernel's roles -- a read of a.dt inside the step
    // loop is a load hipcc waits for with vmcnt(0), which drains the register ring every step)
    const T dt = a.dt;
    // The parameter block is copied into registers once: the asm statements below carry "memory" clobbers
    // (they pin the order of loads and stores the vmcnt arithmetic relies on), and a clobber would otherwise
    // make hipcc reload every parameter from memory after each of them.
    using PLp = ParamLayout<Dyn::NSYS, NX, NU>;
    T p[PLp::QS];   // [system constants | x_target | Q | R | Q_f]: all the rollout reads
#pragma unroll
    for (int i = 0; i < PLp::QS; ++i) p[i] = a.params[i];
    T x[NX], u[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = a.x0[(size_t)i * B + bb];
    ClockProbe cp;
    cp.start();
    T cost = T(0);
    const int stepX = (int)(B * NX * sizeof(T)), stepU = (int)(B * NU * sizeof(T)), stepG = (int)(B * R * sizeof(T));
    const unsigned bytesX = (unsigned)((size_t)a.n_slots * (N + 1) * NX * B * sizeof(T));
    const unsigned bytesU = (unsigned)((size_t)a.n_slots * N * NU * B * sizeof(T));
    // candidate stores go through buffer descriptors: the time step moves in the scalar offset (no per-store
    // 64-bit VALU address arithmetic), and a dead lane's offset lies beyond the descriptor's range, where the
    // hardware drops the store (no exec-mask juggling around the counted stores)
    // (the same descriptors serve the ring's asm loads: one set of SGPRs)
    const __amdgpu_buffer_rsrc_t rXc = make_rsrc(a.X, bytesX), rUc = make_rsrc(a.U, bytesU);
    // (fp32 only: the fp64 kernels keep the exec-mask predicate, which measured faster there; the dropped form is
    // correct for 64-bit stores too -- backward_tile16.hpp, DROP)
    constexpr bool DROP = sizeof(T) == 4 || ILQR_DROP_ALL;
    const int kDropped = 0x7ffffff0;
    const int vXc = (live || !DROP) ? (int)(vec_at(B, N + 1, NX, cslot, 0, bb) * sizeof(T)) : kDropped;
    const int vUc = (live || !DROP) ? (int)(vec_at(B, N, NU, cslot, 0, bb) * sizeof(T)) : kDropped;
    const __amdgpu_buffer_rsrc_t srdG = make_rsrc(a.gains, (unsigned)((size_t)N * B * R * sizeof(T)));
    const int vx = (int)(vec_at(B, N + 1, NX, slot, 0, bb) * sizeof(T));
    const int vu = (int)(vec_at(B, N, NU, slot, 0, bb) * sizeof(T));
    const int vg = (int)((size_t)bb * R * sizeof(T));
    // first: the slot holds nothing yet (prologue); otherwise a refill, tied to the slot's consumed inputs
    auto issue = [&](In& in, int t, auto first) {
        in.template issue<decltype(first)::value>(rXc, rUc, srdG, vx, vu, vg, t * stepX, t * stepU, t * stepG);
    };
    constexpr std::true_type kFirst{};
    constexpr std::false_type kRefill{};
    auto do_step = [&](const In& in, int t) {
        T xo[NX], uo[NU], dx[NX];
        in.unpack(xo, uo);
#ifndef ILQR_NO_CTRL_PK
        if constexpr (sizeof(T) == 4 && NX == 4) {
            // fp32, n_x = 4: the feedback K (x - x_old) on pairs in packed FP32 (the same four products, summed pairwise)
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto pair = [](float a, float b) { f2 r; r.x = a; r.y = b; return r; };
            const f2 d01 = pair(x[0], x[1]) - pair(xo[0], xo[1]), d23 = pair(x[2], x[3]) - pair(xo[2], xo[3]);
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const f2 f = __builtin_elementwise_fma(pair(in.gain(j * NX + 2), in.gain(j * NX + 3)), d23,
                                                       pair(in.gain(j * NX + 0), in.gain(j * NX + 1)) * d01);
                u[j] = uo[j] + alpha * in.gain(NU * NX + j) + (f.x + f.y);   // iLQR_class.py:181-182
            }
        } else
#endif
        {
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = x[i] - xo[i];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            T fb = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) fb += in.gain(j * NX + i) * dx[i];
            u[j] = uo[j] + alpha * in.gain(NU * NX + j) + fb;   // iLQR_class.py:181-182
        }
        }
        // exactly NST stores per step for every wave that is still running (they are counted)
        if (DROP || live) {
            buf_store_vec<T, NX>(rXc, vXc, uniform(t * stepX), x);
            buf_store_vec<T, NU>(rUc, vUc, uniform(t * stepU), u);
        }
        cost += Cost<T, Dyn>::stage(p, dt, x, u);
        T xn[NX];
        Stepper<T, Dyn>::step(INTEG, p, dt, x, u, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
    };
    int t = 0;
    // leading remainder: one slot, fully waited (whole rings only in the pipelined loop)
    for (int r = N % PF; r > 0; --r, ++t) {
        In in;
        issue(in, t, kFirst);
        in.template wait<0>();
        do_step(in, t);
    }
    if (t < N) {
        In ring[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) issue(ring[q], t + q, kFirst);
#pragma unroll
        for (int q = 0; q < PF; ++q) {   // first pass: the prologue loads may be the only operations in flight
            ring[q].template wait<(PF - 1) * NLD>();
            do_step(ring[q], t + q);
            issue(ring[q], (t + q + PF < N) ? t + q + PF : N - 1, kRefill);
        }
        for (t += PF; t < N; t += PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                ring[q].template wait<(PF - 1) * (NLD + NST)>();
                do_step(ring[q], t + q);
                issue(ring[q], (t + q + PF < N) ? t + q + PF : N - 1, kRefill);   // clamped: branch-free refill
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (live) {
        vec_store<T, NX>(a.X + vec_at(B, N + 1, NX, cslot, N, bb), x);
        cost += Cost<T, Dyn>::terminal(p, x);
        a.costs[(size_t)ai * B + b] = cost;
    }
    cp.stop(a.probe, 1);
}

template <typename T, typename Dyn, int INTEG>
__global__ void __launch_bounds__(64) forward_ring_kernel(KArgs<T> a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int ai = blockIdx.y;
    if (a.init_mode && blockIdx.x == 0 && ai == 0 && threadIdx.x == 0) a.counters[a.counter_idx] = 0;   // the select that follows counts into it
    rollout_ring<T, Dyn, INTEG>(a, b, ai, b < a.B);
}

// ---------------------------------------------------------------------------
// select: backtracking acceptance "first alpha with cost_new <= cost"
// (iLQR_class.py:289-297) over the candidates of one pass, and, on the last pass
// of an iteration, the loop bookkeeping of optimize_trajectory (:267-271, :304-311).
// ---------------------------------------------------------------------------
// One trajectory's acceptance step over the candidates of a pass (non-init form).  Writes cost, cost_prev, alpha_taken
// and -- on the last pass of an iteration -- iters, accepted and the new status; returns whether the trajectory is still
// active.  slot_out = the slot its current trajectory lives in afterwards (the caller stores it: select_kernel as it
// is, the fused sweep after moving the trajectory to slot 0); status_out = its status word afterwards.
template <typename T>
ILQR_DEV bool select_candidates(const KArgs<T>& a, int b, bool last_pass, int& slot_out, int& status_out) {
    const size_t B = a.B;
    // everything the decision reads is fetched up front, whatever the status turns out to be (all addresses are valid
    // for any b < B): ONE memory round trip -- a first-match loop that loads as it goes serialised up to n_pass of them
    // (most of select_kernel's former 8 us), and loads behind the status test were a second one at the head of the
    // fused kernel, where the whole workgroup waits for this lane
    int st = a.status[b];
    int slot = a.cur_slot[b];
    const int acc_in = a.accepted[b];
    const T c0 = a.cost[b];
    T cs[kMaxAlpha];
#pragma unroll
    for (int ai = 0; ai < kMaxAlpha; ++ai) cs[ai] = ai < a.n_pass ? a.costs[(size_t)ai * B + b] : T(0);
    const int it_in = a.iters[b];
    const T cp_in = a.cost_prev[b];
    T cost_now = c0, cost_before = cp_in;      // cost / cost_prev as they stand after this pass
    bool still_active = false;
    if (traj_active(st)) {
        int acc = acc_in;
        if (!acc) {
            int first = -1;
#pragma unroll
            for (int ai = kMaxAlpha - 1; ai >= 0; --ai)
                if (ai < a.n_pass && cs[ai] <= c0) first = ai;  // NaN compares false, like the reference
            if (first >= 0) {
                T c = cs[0], al = a.alphas[0];
#pragma unroll
                for (int ai = 1; ai < kMaxAlpha; ++ai)
                    if (ai == first) { c = cs[ai]; al = a.alphas[ai]; }
                slot = (slot + 1 + first) % a.n_slots;
                a.cost_prev[b] = c0;
                a.cost[b] = c;
                a.alpha_taken[b] = al;
                cost_now = c;
                cost_before = c0;
                acc = 1;
            }
        }
        if (last_pass) {
            const int it = it_in + 1;
            a.iters[b] = it;
            a.accepted[b] = 0;
            if (!(a.flags & ILQR_FLAG_KEEP_ITERATING)) {
                if (!acc) {
                    st = (st & ~0xff) | ILQR_TRAJ_LINESEARCH_FAILED;
                } else if (it >= a.maxiter) {
                    st = (st & ~0xff) | ILQR_TRAJ_MAXITER;
                } else {
                    const T d = M<T>::abs(cost_now - cost_before);
                    if (d <= a.tol) st = (st & ~0xff) | ILQR_TRAJ_CONVERGED;
                }
                a.status[b] = st;
            }
            if (!acc) a.alpha_taken[b] = T(0);
        } else {
            a.accepted[b] = acc;
        }
        still_active = traj_active(st);
    }
    slot_out = slot;
    status_out = st;
    return still_active;
}

template <typename T>
__global__ void __launch_bounds__(256) select_kernel(KArgs<T> a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    bool still_active = false;
    if (b < a.B) {
        if (a.init_mode) {
            // head of optimize_trajectory: X, U, cost <- forward_pass(alpha = 0)  (:257-259)
            const T c = a.costs[b];
            a.cur_slot[b] = (a.cur_slot[b] + 1) % a.n_slots;
            a.cost[b] = c;
            a.cost_prev[b] = c;
            a.alpha_taken[b] = T(0);
            a.status[b] = ILQR_TRAJ_ACTIVE;
            a.iters[b] = 0;
            a.accepted[b] = 0;
            still_active = true;
        } else {
            int slot, st;
            const int slot_before = a.cur_slot[b];
            still_active = select_candidates(a, b, a.last_pass != 0, slot, st);
            if (slot != slot_before) a.cur_slot[b] = slot;
        }
    }
    if (a.last_pass || a.init_mode) {
        const unsigned long long m = __ballot(still_active);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.counters[a.counter_idx], (int)__popcll(m));
        // the next iteration's counter is cleared here (stream order makes it safe), so the host loop
        // needs no memset launch per iteration
        if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[(a.counter_idx + 1) % kCounterRing] = 0;
    }
}

// ---------------------------------------------------------------------------
// status_reduce: shard-local {min cost, max |dcost|, #active, #converged} as 4 doubles, the
// operand of the only inter-GPU collective of the path (one block; B is a few thousand).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) status_reduce_kernel(const T* cost, const T* cost_prev, const int* status,
                                                            int B, double* out4) {
    __shared__ double s_min[256], s_max[256];
    __shared__ int s_act[256], s_conv[256];
    double mn = 1.0 / 0.0, mx = 0.0;
    int act = 0, conv = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double c = (double)cost[b];
        mn = c < mn ? c : mn;
        const double d = fabs(c - (double)cost_prev[b]);
        mx = d > mx ? d : mx;
        const int st = status[b] & 0xff;
        act += st == ILQR_TRAJ_ACTIVE;
        conv += st == ILQR_TRAJ_CONVERGED;
    }
    s_min[threadIdx.x] = mn; s_max[threadIdx.x] = mx; s_act[threadIdx.x] = act; s_conv[threadIdx.x] = conv;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const int o = threadIdx.x + w;
            s_min[threadIdx.x] = s_min[o] < s_min[threadIdx.x] ? s_min[o] : s_min[threadIdx.x];
            s_max[threadIdx.x] = s_max[o] > s_max[threadIdx.x] ? s_max[o] : s_max[threadIdx.x];
            s_act[threadIdx.x] += s_act[o];
            s_conv[threadIdx.x] += s_conv[o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out4[0] = s_min[0]; out4[1] = s_max[0]; out4[2] = s_act[0]; out4[3] = s_conv[0]; }
}

// ---------------------------------------------------------------------------
// eval_points: the 12 System callables (system_base.py:223-251) at arbitrary
// points, dense row-major outputs; NULL outputs are skipped.
// ---------------------------------------------------------------------------
template <typename T> struct EvalArgs {
    int npts, integ;
    T dt;
    const T* params;
    const T* x; const T* u;
    T *f, *f_x, *f_u, *l, *l_x, *l_u, *l_xx, *l_ux, *l_uu, *l_f, *l_f_x, *l_f_xx;
};

template <typename T, typename Dyn>
__global__ void __launch_bounds__(64) eval_points_kernel(EvalArgs<T> a) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npts) return;
    const T* __restrict__ p = a.params;
    T x[NX], u[NU];
#pragma unroll
    for (int r = 0; r < NX; ++r) x[r] = a.x[(size_t)i * NX + r];
#pragma unroll
    for (int r = 0; r < NU; ++r) u[r] = a.u ? a.u[(size_t)i * NU + r] : T(0);
    if (a.f_x || a.f_u) {
        T xn[NX], fx[NX][NX], fu[NX][NU];
        Stepper<T, Dyn>::step_jac(a.integ, p, a.dt, x, u, xn, fx, fu);
#pragma unroll
        for (int r = 0; r < NX; ++r) {
#pragma unroll
            for (int c = 0; c < NX; ++c)
                if (a.f_x) a.f_x[((size_t)i * NX + r) * NX + c] = fx[r][c];
#pragma unroll
            for (int c = 0; c < NU; ++c)
                if (a.f_u) a.f_u[((size_t)i * NX + r) * NU + c] = fu[r][c];
        }
    }
    if (a.f) {
        // the plain step, exactly what the rollout uses
        T xn[NX];
        Stepper<T, Dyn>::step(a.integ, p, a.dt, x, u, xn);
#pragma unroll
        for (int r = 0; r < NX; ++r) a.f[(size_t)i * NX + r] = xn[r];
    }
    if (a.l) a.l[i] = Cost<T, Dyn>::stage(p, a.dt, x, u);
    if (a.l_x || a.l_u) {
        T g[NX], gu[NU];
        Cost<T, Dyn>::grad(p, a.dt, x, u, g, gu);
#pragma unroll
        for (int r = 0; r < NX; ++r)
            if (a.l_x) a.l_x[(size_t)i * NX + r] = g[r];
#pragma unroll
        for (int r = 0; r < NU; ++r)
            if (a.l_u) a.l_u[(size_t)i * NU + r] = gu[r];
    }
    if (a.l_xx || a.l_ux || a.l_uu) {
        T lxx[NX][NX], lux[NU][NX], luu[NU][NU];
        Cost<T, Dyn>::hess(p, a.dt, x, u, lxx, lux, luu);
#pragma unroll
        for (int r = 0; r < NX * NX; ++r)
            if (a.l_xx) a.l_xx[(size_t)i * NX * NX + r] = lxx[r / NX][r % NX];
#pragma unroll
        for (int r = 0; r < NU * NX; ++r)
            if (a.l_ux) a.l_ux[(size_t)i * NU * NX + r] = lux[r / NX][r % NX];
#pragma unroll
        for (int r = 0; r < NU * NU; ++r)
            if (a.l_uu) a.l_uu[(size_t)i * NU * NU + r] = luu[r / NU][r % NU];
    }
    if (a.l_f) a.l_f[i] = Cost<T, Dyn>::terminal(p, x);
    if (a.l_f_x) {
        T g[NX];
        Cost<T, Dyn>::l_f_x(p, x, g);
#pragma unroll
        for (int r = 0; r < NX; ++r) a.l_f_x[(size_t)i * NX + r] = g[r];
    }
    if (a.l_f_xx) {
        T H[NX][NX];
        Cost<T, Dyn>::l_f_xx(p, x, H);
#pragma unroll
        for (int r = 0; r < NX * NX; ++r) a.l_f_xx[(size_t)i * NX * NX + r] = H[r / NX][r % NX];
    }
}

// ---------------------------------------------------------------------------
// MPC advance (run_iLQR_MPC.py:127-140): u0 = U[:,0]; plant step with the plant's own
// integrator; x_0 <- new plant state; warm start <- shift(U) repeating the last column.
// One lane per trajectory (the shift walks its own column in place: read t+1, write t).
// ---------------------------------------------------------------------------
template <typename T> struct MpcArgs {
    int B, N, plant_integ, step;
    T dt;
    const T* params;
    T* U; const int* cur_slot; T* x0; T* plant_x;
    T* u_log; T* x_log; T* cost_log; const T* cost;  // logs in the ABI layout [n_steps][B][...], or NULL
};

// Workgroup = 64 trajectories x kMpcChunks slices of the horizon (threadIdx.y): the shift U[t] <- U[t + 1] reads its
// slice into registers, the workgroup synchronises, then writes -- a lane no longer walks its whole column alone (200
// dependent load / store pairs: 66 us of the c4 step, now a few).  Slice 0 also runs the plant step.  Horizons beyond
// kMpcChunks * (32 / n_u) + 1 steps fall back to the walk.
constexpr int kMpcChunks = 16, kMpcSliceScalars = 32;   // a slice keeps at most 32 scalars per lane in registers
template <typename T, typename Dyn>
__global__ void __launch_bounds__(64 * kMpcChunks) mpc_advance_kernel(MpcArgs<T> a) {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    constexpr int kMpcSlice = kMpcSliceScalars / NU > 0 ? kMpcSliceScalars / NU : 1;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int chunk = threadIdx.y;
    const size_t B = a.B;
    const bool inb = b < a.B;
    const int bb = inb ? b : a.B - 1;
    T* Uc = a.U + vec_at(B, a.N, NU, a.cur_slot[bb], 0, bb);
    const size_t sU = B * NU;
    const int n_shift = a.N - 1;                                   // U[t] <- U[t + 1], t = 0 .. N-2
    const int per = (n_shift + kMpcChunks - 1) / kMpcChunks;       // time steps per slice
    const bool sliced = per <= kMpcSlice;
    T u0[NU];
    vec_load<T, NU>(Uc, u0);                                       // (before anything is shifted)
    T keep[kMpcSlice][NU];
    const int t0 = chunk * per;
    if (sliced) {
#pragma unroll
        for (int q = 0; q < kMpcSlice; ++q)
            if (q < per && t0 + q < n_shift) vec_load<T, NU>(Uc + (size_t)(t0 + q + 1) * sU, keep[q]);
    }
    __syncthreads();
    if (sliced) {
        if (inb) {
#pragma unroll
            for (int q = 0; q < kMpcSlice; ++q)
                if (q < per && t0 + q < n_shift) vec_store<T, NU>(Uc + (size_t)(t0 + q) * sU, keep[q]);
        }
    }
    if (chunk != 0 || !inb) return;
    T x[NX], xn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = a.plant_x[(size_t)i * B + b];
    Stepper<T, Dyn>::step(a.plant_integ, a.params, a.dt, x, u0, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        a.plant_x[(size_t)i * B + b] = xn[i];
        a.x0[(size_t)i * B + b] = xn[i];
        if (a.x_log) a.x_log[((size_t)a.step * B + b) * NX + i] = xn[i];
    }
#pragma unroll
    for (int j = 0; j < NU; ++j)
        if (a.u_log) a.u_log[((size_t)a.step * B + b) * NU + j] = u0[j];
    if (a.cost_log) a.cost_log[(size_t)a.step * B + b] = a.cost[b];
    if (!sliced) {
        for (int t = 0; t + 1 < a.N; ++t) {
            T un[NU];
            vec_load<T, NU>(Uc + (t + 1) * sU, un);
            vec_store<T, NU>(Uc + t * sU, un);
        }
    }
}

// ---------------------------------------------------------------------------
// layout conversion between the host layouts of the C-ABI (leading batch axis in
// front of the reference layout) and the device's batch-innermost slots.
// dense[b][c][t] (c = component, t = time) <-> slots[slot(b)][t][b][c]
// ---------------------------------------------------------------------------
template <typename T>
__global__ void scatter_ct_kernel(const T* dense, T* slots, const int* cur_slot, int B, int C, int Tn) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * Tn) return;
    const int c = (int)(idx % C);
    const int b = (int)((idx / C) % B);
    const int t = (int)(idx / ((size_t)B * C));
    const int s = cur_slot ? cur_slot[b] : 0;
    slots[(((size_t)s * Tn + t) * B + b) * C + c] = dense[((size_t)b * C + c) * Tn + t];
}
template <typename T>
__global__ void gather_ct_kernel(T* dense, const T* slots, const int* cur_slot, int B, int C, int Tn) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * Tn) return;
    const int c = (int)(idx % C);
    const int b = (int)((idx / C) % B);
    const int t = (int)(idx / ((size_t)B * C));
    const int s = cur_slot ? cur_slot[b] : 0;
    dense[((size_t)b * C + c) * Tn + t] = slots[(((size_t)s * Tn + t) * B + b) * C + c];
}
// dense[b][t][c] <-> dev[t][c][b]   (K, ILQR_LIN, x0 with Tn = 1)
template <typename T>
__global__ void scatter_tc_kernel(const T* dense, T* dev, int B, int C, int Tn) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * Tn) return;
    const int b = (int)(idx % B);
    const int c = (int)((idx / B) % C);
    const int t = (int)(idx / ((size_t)B * C));
    dev[((size_t)t * C + c) * B + b] = dense[((size_t)b * Tn + t) * C + c];
}
template <typename T>
__global__ void gather_tc_kernel(T* dense, const T* dev, int B, int C, int Tn) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * Tn) return;
    const int b = (int)(idx % B);
    const int c = (int)((idx / B) % C);
    const int t = (int)(idx / ((size_t)B * C));
    dense[((size_t)b * Tn + t) * C + c] = dev[((size_t)t * C + c) * B + b];
}


// gains[t][b][R] <-> the reference layouts K [B][N][n_u][n_x], U_ff [B][n_u][N]
template <typename T>
__global__ void gains_scatter_K_kernel(const T* denseK, T* gains, int B, int N, int MN, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * MN) return;
    const int c = (int)(idx % MN);
    const int t = (int)((idx / MN) % N);
    const int b = (int)(idx / ((size_t)MN * N));
    gains[((size_t)t * B + b) * R + c] = denseK[idx];
}
template <typename T>
__global__ void gains_gather_K_kernel(T* denseK, const T* gains, int B, int N, int MN, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * MN) return;
    const int c = (int)(idx % MN);
    const int t = (int)((idx / MN) % N);
    const int b = (int)(idx / ((size_t)MN * N));
    denseK[idx] = gains[((size_t)t * B + b) * R + c];
}
template <typename T>
__global__ void gains_scatter_k_kernel(const T* denseUff, T* gains, int B, int N, int M, int MN, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * M) return;
    const int t = (int)(idx % N);
    const int j = (int)((idx / N) % M);
    const int b = (int)(idx / ((size_t)M * N));
    gains[((size_t)t * B + b) * R + MN + j] = denseUff[idx];
}
template <typename T>
__global__ void gains_gather_k_kernel(T* denseUff, const T* gains, int B, int N, int M, int MN, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * M) return;
    const int t = (int)(idx % N);
    const int j = (int)((idx / N) % M);
    const int b = (int)(idx / ((size_t)M * N));
    denseUff[idx] = gains[((size_t)t * B + b) * R + MN + j];
}

}  // namespace ilqr
#include "backward_fused16.hpp"
#include "persistent.hpp"

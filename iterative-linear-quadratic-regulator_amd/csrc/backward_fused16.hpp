// backward_fused16.hpp -- acceptance step + linearisation + backward Riccati sweep as ONE kernel for the systems the
// 16-lane DPP sweep serves (n_u = 1, 2 <= n_x <= 4; iLQR_class.py:289-307, 318-331, 79-161).
//
// Why: in the materialised path (linearize_kernel -> backward_tile16_kernel) the 157 MB tile tensor of the c3 shape is
// written to HBM only to be read once a microsecond later, the sweep's four waves per CU issue a vector instruction in
// about half of their cycles (a 200-step latency chain), and the acceptance step between two iterations is a launch of
// its own.  Here one workgroup owns 16 trajectories from start to end:
//   * threads 0..15 run the acceptance step of the previous iteration's candidates for their trajectory
//     (select_candidates: the same code as select_kernel) -- no other workgroup ever touches these trajectories, so the
//     bookkeeping needs no grid-wide ordering;
//   * P producer waves (lane = trajectory x 4 consecutive time steps) evaluate the expansion of `unit` k = time steps
//     N-1-4k .. N-4-4k with exactly the arithmetic of linearize_kernel (Stepper::step_jac, tile16_pack), move the
//     accepted trajectory into slot 0 as linearize_kernel does, and hand the tiles over through a ring of RU units in
//     LDS;
//   * the four sweep waves (one per SIMD, 4 trajectories each, raised priority) run the DPP step of
//     backward_tile16.hpp on tiles read from that ring, one tile of lookahead, and store the gains.
// Synchronisation is workgroup-local and needs no barrier after the head of the kernel: LDS executes a wave's
// instructions in program order, so "tile writes, then ready[slot] = k + 1" by a producer and "poll ready[slot], then
// tile reads" by a sweep wave are ordered by the hardware; the compiler is kept from reordering them by empty asm
// statements with a memory clobber (a release / acquire fence would also drain the sweep's outstanding gain stores).
// A sweep wave publishes done[w] = k + 1 once every read of unit k has been issued; a producer overwrites ring slot
// k % RU only when all four have passed unit k - RU.
//
// The tile tensor is never materialised: the kernel's HBM traffic is the trajectory (read, and written once when it
// moves to slot 0) and the gains -- it is bound by vector-instruction issue, not by HBM, and bench.py reports it so.
#pragma once

namespace ilqr {

// TPW: trajectories per workgroup = 16 (four sweep waves: the big-batch form, one workgroup per CU at B = 4096) or 4
// (one sweep wave: a batch of <= 1024 then spreads over all 256 CUs instead of 64, and the kernel's duration falls from
// the CU's vector-issue bound -- 16 trajectories' worth of linearisation -- to the sweep's own latency chain).
// P: producer waves; RU: ring slots (units of 64 tiles = 64 / TPW time steps x TPW trajectories); TILE: scalars between
// two tiles in LDS (48 + padding: the 16-byte writes of 8 neighbouring lanes then fall into distinct banks in fp32 --
// 52 dwords; fp64 keeps its tiles 32-byte aligned for the double4 reads of the sweep and takes a two-way conflict on
// the writes)
// PK (fp32, TPW = 16, explicit integrators): the producers evaluate TWO time steps per lane in packed FP32 -- the
// system, integrator and cost templates instantiated on a float pair (dynamics.hpp, pair_f32): 817 vector instructions
// for two points against 989 for one, bit-identical values.  A unit is then 128 tiles; four producer waves (the pair
// form needs ~200 VGPRs: 512-thread workgroups) produce as many points per pass as eight scalar ones.
template <typename T, int TPW, bool PK> struct FusedCfg;
template <> struct FusedCfg<float, 16, false> { static constexpr int P = 8, RU = 8; };
#ifndef ILQR_PK_RU
#define ILQR_PK_RU 5        // ring slots of the pair producers (A/B builds: tools/build_variants.sh; 4: 45.6 us, 5: 42.5 us at B = 4096)
#endif
#ifndef ILQR_PK_PAD
#define ILQR_PK_PAD 0       // extra scalars behind every PAIR of tiles (lane stride 2 TL + PAD: 104 dwords put lanes l, l + 4 on the same banks)
#endif
template <> struct FusedCfg<float, 16, true> { static constexpr int P = 4, RU = ILQR_PK_RU; };
template <> struct FusedCfg<double, 16, false> { static constexpr int P = 4, RU = 4; };   // (P = 8 caps the kernel at 168 VGPRs: the fp64 RK4 producer spills)
template <> struct FusedCfg<float, 4, false> { static constexpr int P = 3, RU = 4; };
template <> struct FusedCfg<double, 4, false> { static constexpr int P = 3, RU = 3; };

// scalars between two tiles in LDS: the n_u = 1 tile (48) or the (4, 2) tile of backward_tile16m2.hpp (64), + 4 of padding
template <int NU> constexpr int fused_tl() { return NU == 2 ? 68 : 52; }
template <bool PK, int NU> constexpr int fused_unit_scalars() { return (PK ? 128 : 64) * fused_tl<NU>() + (PK ? 64 * ILQR_PK_PAD : 0); }
// ring slots: the configuration's, except that the pair producers' 128-tile units of the (4, 2) tile fit four times into 160 KB
template <typename T, int TPW, bool PK, int NU> constexpr int fused_ru() { return (PK && NU == 2 && FusedCfg<T, TPW, PK>::RU > 4) ? 4 : FusedCfg<T, TPW, PK>::RU; }
template <typename T, int TPW, bool PK, int NU = 1> constexpr int fused_lds_bytes() {
    return fused_ru<T, TPW, PK, NU>() * fused_unit_scalars<PK, NU>() * (int)sizeof(T) + (fused_ru<T, TPW, PK, NU>() + 4 + 32 + 4) * 4;
}
static_assert(fused_lds_bytes<float, 16, true, 1>() <= 160 * 1024 && fused_lds_bytes<float, 16, true, 2>() <= 160 * 1024 &&
              fused_lds_bytes<float, 16, false, 2>() <= 160 * 1024 && fused_lds_bytes<double, 16, false, 2>() <= 160 * 1024, "LDS ring exceeds a CU");
template <typename T, int TPW, bool PK> constexpr int fused_threads() { return 64 * (TPW / 4 + FusedCfg<T, TPW, PK>::P); }

// the sweep's view of a tile in LDS, and the step that consumes it
template <typename T, int NU> struct FusedStep;
template <> struct FusedStep<float, 1> {
    using Tile = TileQ;
    static ILQR_DEV void load(Tile& t, const float* tp, int i, int j, int l16) {
        const float4 s = *reinterpret_cast<const float4*>(tp + 4 * j);
        const float4 v = *reinterpret_cast<const float4*>(tp + 32 + 4 * j);
        t.skj[0] = s.x; t.skj[1] = s.y; t.skj[2] = s.z; t.skj[3] = s.w;
        t.vj[0] = v.x; t.vj[1] = v.y; t.vj[2] = v.z; t.vj[3] = v.w;
        t.a = tp[l16];
        t.lxx = tp[16 + l16];
        t.c = tp[32 + l16];
    }
    // out: the scalar this lane stores into the gain record (lanes (0, j): K[j]; the others: k)
    static ILQR_DEV void step(const Tile& c, const LaneConst<float>& lc, int i, int j, float& V, float& vx, float& out, bool& pd) {
        RawTileQ none;            // (the refill arguments of the ring form are unused without REFILL)
        const i32x4 srd = {0, 0, 0, 0};
        const TileOffsets off = {0, 0, 0};
        float Kj, kff;
        tile16_step_f32<false>(c, lc, V, vx, Kj, kff, pd, none, srd, off, 0);
        out = (i == 0) ? Kj : kff;
    }
};
template <> struct FusedStep<double, 1> {
    using Tile = Tile16<double>;
    static ILQR_DEV void load(Tile& t, const double* tp, int i, int j, int l16) { tile16_load_lds(t, tp, i, j, l16); }
    static ILQR_DEV void step(const Tile& c, const LaneConst<double>& lc, int i, int j, double& V, double& vx, double& out, bool& pd) {
        double Kj, kff;
        tile16_step<double, false>(c, lc, 0.0, V, vx, Kj, kff, pd);
        out = (i == 0) ? Kj : kff;
    }
};
// n_x = 4, n_u = 2, fp32: the scheduled step (gen_tile16m2_step.py) on its own view of the 64-scalar tile
template <> struct FusedStep<float, 2> {
    using Tile = TileQ2;
    static ILQR_DEV void load(Tile& t, const float* tp, int i, int j, int l16) { tileq2_load_lds(t, tp, i, j, l16); }
    static ILQR_DEV void step(const Tile& c, const LaneConst<float>& lc, int i, int j, float& V, float& vx, float& out, bool& pd) {
        const GainSel sel = GainSel::of(i, j);      // (loop invariant: three compares outside the sweep's loop)
        tile16m2_step_f32(c, lc, sel, V, vx, out, pd);
    }
};
// n_x = 4, n_u = 2 (fp64): the 64-scalar tile and the step of backward_tile16m2.hpp
template <typename T> struct FusedStep<T, 2> {
    using Tile = Tile16M2<T>;
    using V4 = typename Vec4<T>::type;
    static ILQR_DEV void load(Tile& t, const T* tp, int i, int j, int l16) {
        const V4 si = *reinterpret_cast<const V4*>(tp + 4 * i);
        const V4 sj = *reinterpret_cast<const V4*>(tp + 4 * j);
        const V4 g0 = *reinterpret_cast<const V4*>(tp + 32 + 8 * j);
        const V4 g1 = *reinterpret_cast<const V4*>(tp + 36 + 8 * j);
        t.ski[0] = si.x; t.ski[1] = si.y; t.ski[2] = si.z; t.ski[3] = si.w;
        t.skj[0] = sj.x; t.skj[1] = sj.y; t.skj[2] = sj.z; t.skj[3] = sj.w;
        t.gj[0] = g0.x; t.gj[1] = g0.y; t.gj[2] = g0.z; t.gj[3] = g0.w;
        t.gj[4] = g1.x; t.gj[5] = g1.y; t.gj[6] = g1.z; t.gj[7] = g1.w;
        const T* gi = tp + 32 + 8 * i;
        t.gi[0] = gi[0]; t.gi[1] = gi[1]; t.gi[3] = gi[3]; t.gi[4] = gi[4];
        t.lxx = tp[16 + l16];
    }
    // out: lanes (0, j): K[0][j]; (1, j): K[1][j]; rows 2, 3: k[0] (j = 0) or k[1]
    static ILQR_DEV void step(const Tile& c, const LaneConst<T>& lc, int i, int j, T& V, T& vx, T& out, bool& pd) {
        T K0, K1, k0, k1;
        tile16m2_step<T, false>(c, lc.m0, lc.m1, T(0), V, vx, K0, K1, k0, k1, pd);
        out = (i == 0) ? K0 : ((i == 1) ? K1 : ((j == 0) ? k0 : k1));
    }
};

ILQR_DEV void compiler_fence() { asm volatile("" ::: "memory"); }
ILQR_DEV int lds_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
ILQR_DEV void lds_poke(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The roles of a workgroup as functions of one struct, so that the fused kernel below and the persistent kernel
// (persistent.hpp: the whole iteration loop of a workgroup's trajectories in one launch) run the same code.
template <typename T, typename Dyn, int INTEG, int TPW, bool PK> struct FusedWG {
    static constexpr int NX = Dyn::NX, NU = Dyn::NU;
    static_assert((NU == 1 && NX >= 2 && NX <= 4) || (NX == 4 && NU == 2), "the fused sweep serves the DPP tiles: n_u = 1, or (4, 2)");
    static_assert(TPW == 16 || TPW == 4, "16 or 4 trajectories per workgroup");
    static_assert(!PK || (sizeof(T) == 4 && TPW == 16 && INTEG != ILQR_INT_BACKWARD_EULER), "pair producers: fp32, 16-trajectory workgroups, explicit integrators");
    using Cfg = FusedCfg<T, TPW, PK>;
    static constexpr int P = Cfg::P, RU = fused_ru<T, TPW, PK, NU>(), TL = fused_tl<NU>();
    static constexpr int NV = (NU == 2 ? kTile16M2 : kTile16) / 4;   // V4 pieces of a tile
    static constexpr int NSW = TPW / 4;                 // sweep waves
    static constexpr int UT = PK ? 128 : 64;            // tiles per unit (one pass of a producer wave)
    static constexpr int US = UT / TPW;                 // time steps per unit
    static constexpr int UNIT = fused_unit_scalars<PK, NU>();   // scalars per ring slot
    static constexpr int PS = 2 * TL + ILQR_PK_PAD;     // pair producers: scalars between two lanes' pairs of tiles
    static constexpr int R = gain_record(NX, NU);
    using PL = ParamLayout<Dyn::NSYS, NX, NU>;
    using V4 = typename Vec4<T>::type;
    // where the tile of (step r of the unit, trajectory tl) sits in its ring slot: step-major, or -- pair producers --
    // the two steps a lane evaluates together next to each other
    static ILQR_DEV constexpr int tile_off(int r, int tl) { return PK ? ((r / 2) * TPW + tl) * PS + (r % 2) * TL : (r * TPW + tl) * TL; }
    static ILQR_DEV constexpr int step_off(int r) { return PK ? (r / 2) * TPW * PS + (r % 2) * TL : r * TPW * TL; }

    struct Lds {
        T* ring;
        int* ready;      // [RU] unit index + 1 held by the slot
        int* done;       // [4]  units fully read, per sweep wave (NSW of them used)
        int* s_slot;     // [16] slot of the trajectory's current (X, U)
        int* s_stat;     // [16] status word after the acceptance step (-1: beyond the batch)
    };
    static ILQR_DEV Lds carve(unsigned char* base) {
        Lds L;
        L.ring = reinterpret_cast<T*>(base);
        L.ready = reinterpret_cast<int*>(base + (size_t)RU * UNIT * sizeof(T));
        L.done = L.ready + RU;
        L.s_slot = L.done + 4;
        L.s_stat = L.s_slot + 16;
        return L;
    }

    // ---- head (wave 0 only; the caller puts a workgroup barrier behind it): the acceptance step of the newest candidates
    // (iLQR_class.py:289-307), one lane per trajectory, or just the trajectories' status and slot; clears the ring's flags.
    // count: add the number of still-active trajectories to counters[counter_idx] (the host loop's read-back).
    static ILQR_DEV void head(const KArgs<T>& a, const Lds& L, int b0, int lane, bool do_select, bool count) {
        bool still = false;
        if (lane < 16) {
            const int b = b0 + lane;
            int slot = 0, st = -1;
            if (lane < TPW && b < a.B) {
                if (do_select) {
                    still = select_candidates(a, b, true, slot, st);
                } else {
                    st = a.status[b];
                    slot = a.cur_slot[b];
                }
                // an active trajectory is moved to slot 0 below (linearize_kernel's canonicalisation); a finished one
                // stays where its accepted candidate is
                a.cur_slot[b] = traj_active(st) ? 0 : slot;
            }
            L.s_slot[lane] = slot;
            L.s_stat[lane] = st;
        } else if (lane < 16 + RU + 4) {
            L.ready[lane - 16] = 0;     // ready[RU], done[4] are contiguous
        }
        if (do_select && count) {
            const unsigned long long m = __ballot(still);
            if (lane == 0 && m) atomicAdd(&a.counters[a.counter_idx], (int)__popcll(m));
            if (blockIdx.x == 0 && lane == 0) a.counters[(a.counter_idx + 1) % kCounterRing] = 0;
        }
    }
    static ILQR_DEV bool any_active(const Lds& L) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < 16; ++q) any = any || traj_active(L.s_stat[q]);
        return any;
    }

    // ---- sweep role (waves 0 .. NSW-1) ------------------------------------------------------------------------------------
    static ILQR_DEV void sweep(const KArgs<T>& a, const Lds& L, int b0, int wave, int lane, ClockProbe& cp) {
        const size_t B = a.B;
        const int N = a.N;
        const int tid = threadIdx.x;
        (void)tid;
        // ================= sweep: 4 trajectories per wave, lane (i, j) of a 16-lane row owns V_xx[i][j] ===============
        __builtin_amdgcn_s_setprio(2);
        const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3;
        const int tl = 4 * wave + (lane >> 4);
        const int gidx = b0 + tl;
        const bool valid = gidx < a.B;
        const int b = valid ? gidx : a.B - 1;
        const int st = L.s_stat[tl];
        const bool act = valid && traj_active(st);
        const int slot = act ? L.s_slot[tl] : 0;
        // terminal expansion at x_N of the accepted trajectory (iLQR_class.py:136-138), which also moves to slot 0
        const T* __restrict__ pp = a.params;
        T xN[NX];
        vec_load<T, NX>(a.X + vec_at(B, N + 1, NX, slot, N, b), xN);
        if (act && slot != 0 && l16 == 0) vec_store<T, NX>(a.X + vec_at(B, N + 1, NX, 0, N, b), xN);
        T V = T(0), vx = T(0);
        {
            T g[NX], H[NX][NX];
            Cost<T, Dyn>::l_f_x(pp, xN, g);
            Cost<T, Dyn>::l_f_xx(pp, xN, H);
#pragma unroll
            for (int ii = 0; ii < NX; ++ii) {
                if (j == ii) vx = g[ii];
#pragma unroll
                for (int jj = 0; jj < NX; ++jj)
                    if (i == ii && j == jj) V = H[ii][jj];
            }
        }
        const unsigned gain_bytes = (unsigned)((size_t)N * B * R * sizeof(T));
        const __amdgpu_buffer_rsrc_t rgain = make_rsrc(a.gains, gain_bytes);
        const int rstride = (int)(B * R * sizeof(T));
        // lanes (0, j) store K[j], lane (1, 0) stores k; fp32: the others carry an offset beyond the descriptor's range
        // and the hardware drops their store (backward_tile16_kernel)
        constexpr bool DROP = sizeof(T) == 4 || ILQR_DROP_ALL;
        // (n_u = 2: lanes (0, j), (1, j) store K[0][j], K[1][j]; lanes (2, 0), (2, 1) k[0], k[1])
        const bool storer = act && (NU == 1 ? ((i == 0 && j < NX) || l16 == 4) : (i < 2 || (i == 2 && j < 2)));
        const int rec_slot = NU == 1 ? (i == 0 ? j : NX) : (i < 2 ? 4 * i + j : 8 + j);
        const int rec_off = (storer || !DROP) ? (int)((b * R + rec_slot) * sizeof(T)) : 0x7ffffff0;
        LaneConst<T> lc;
        lc.m0 = T(j == 0);
        lc.m1 = T(j == 1);
        lc.tr_byte = 4 * ((lane & 48) | (j << 2) | i);
        bool all_pd = true;
        using FS = FusedStep<T, NU>;
        int goff = (N - 1) * rstride;
        // this trajectory's tile of time step r of unit k sits at unit_base(k) + r * TPW * TL: a compile-time offset per step
        auto unit_base = [&](int k) -> const T* { return L.ring + (size_t)(k % RU) * UNIT + tile_off(0, tl); };
        constexpr auto step_off = [](int r) { return PK ? (r / 2) * TPW * PS + (r % 2) * TL : r * TPW * TL; };
        auto one_step = [&](const typename FS::Tile& c) {
            T out;
            bool pd;
            FS::step(c, lc, i, j, V, vx, out, pd);
            all_pd = all_pd && pd;
            if (DROP || storer) buf_store1(rgain, rec_off, uniform(goff), out);
            goff -= rstride;
        };
        // Two tile buffers used alternately (US is even, so the buffer of a step is a compile-time choice: no copies);
        // the tile of step s + 1 is read from LDS while step s computes.  The flag of the next unit is fetched one step
        // before it is needed, so its LDS round trip is not exposed either.
        typename FS::Tile tq[2];
#ifdef ILQR_FUSED_STAMPS   // diagnostic build (tools/fused_stamps.py): where a sweep wave's time goes
        const long long st_t0 = __builtin_readcyclecounter();
        long long st_spin = 0;
#endif
        while (lds_peek(&L.ready[0]) < 1) __builtin_amdgcn_s_sleep(1);
        compiler_fence();
#ifdef ILQR_FUSED_STAMPS
        const long long st_first = __builtin_readcyclecounter();
#endif
        FS::load(tq[0], unit_base(0), i, j, l16);
        const int n_full = N / US, rem = N % US;
        for (int k = 0; k < n_full; ++k) {
            const T* ub = unit_base(k);
            const bool more = (k + 1) * US < N;
            int flag = 0;
#pragma unroll
            for (int r = 0; r < US; ++r) {
                if (r == US - 2 && more) flag = lds_peek(&L.ready[(k + 1) % RU]);
                if (r < US - 1) {
                    FS::load(tq[(r + 1) & 1], ub + step_off(r + 1), i, j, l16);
                } else if (more) {
#ifdef ILQR_FUSED_STAMPS
                    const long long w0 = __builtin_readcyclecounter();
#endif
                    while (flag < k + 2) {
                        __builtin_amdgcn_s_sleep(1);
                        flag = lds_peek(&L.ready[(k + 1) % RU]);
                    }
                    compiler_fence();
#ifdef ILQR_FUSED_STAMPS
                    st_spin += __builtin_readcyclecounter() - w0;
#endif
                    FS::load(tq[0], unit_base(k + 1), i, j, l16);
                }
                one_step(tq[r & 1]);
            }
            // every read of this unit has been issued (LDS serves a wave in order): its L.ring slot may be overwritten
            compiler_fence();
            if (lane == 0) lds_poke(&L.done[wave], k + 1);
        }
        if (rem) {      // the last, partial unit (its first tile is in tq[0] already)
            const T* ub = unit_base(n_full);
#pragma unroll
            for (int r = 0; r < US - 1; ++r) {
                if (r < rem) {
                    if (r + 1 < rem) FS::load(tq[(r + 1) & 1], ub + step_off(r + 1), i, j, l16);
                    one_step(tq[r & 1]);
                }
            }
        }
#ifdef ILQR_FUSED_STAMPS
        if (a.probe && blockIdx.x == 7 && tid == 0) {
            a.probe[5] = st_first - st_t0;                              // head of the sweep role -> first unit L.ready
            a.probe[6] = st_spin;                                       // waiting for later units
            a.probe[7] = __builtin_readcyclecounter() - st_first;       // first unit L.ready -> last step L.done
        }
#endif
        cp.stop(a.probe, 0);
        if (act && l16 == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
    }

    // ---- producer role (waves NSW .. NSW+P-1; pw = the producer's index) ----------------------------------------------------
    static ILQR_DEV void produce(const KArgs<T>& a_, const Lds& L, int b0, int pw, int lane) {
        // (the fields the unit loop reads, as locals: the persistent kernel's roles get the argument block as memory)
        struct { int B; T dt; T* X; T* U; const T* params; } a = {a_.B, a_.dt, a_.X, a_.U, a_.params};
        const size_t B = a.B;
        const int N = a_.N;
        const int n_units = (N + US - 1) / US;
        if constexpr (PK) {
        // ================= pair producers: lane = (trajectory tl, time steps 2 r2 and 2 r2 + 1 of the unit) ===========
        using T2 = pair_f32;
        using Dyn2 = typename Dyn::template rebind<T2>;
        const int tl = lane % TPW, r2 = lane / TPW;
        const int gidx = b0 + tl;
        const bool valid = gidx < a.B;
        const int b = valid ? gidx : a.B - 1;
        const bool actb = valid && traj_active(L.s_stat[tl]);
        const int slot = actb ? L.s_slot[tl] : 0;
        const bool move = actb && slot != 0;
        T pl[PL::TOTAL];
#pragma unroll
        for (int q = 0; q < PL::TOTAL; ++q) pl[q] = a.params[q];
        const SplatParams p{pl};
        const T2 dt2 = a.dt;
        for (int k = pw; k < n_units; k += P) {
            const int ta = N - 1 - k * US - 2 * r2, tb = ta - 1;      // half x: the later step, half y: the one before it
            const bool ina = ta >= 0, inb = tb >= 0;
            const int tta = ina ? ta : 0, ttb = inb ? tb : 0;
            T xa[NX], ua[NU], xb[NX], ub[NU];
            vec_load<T, NX>(a.X + vec_at(B, N + 1, NX, slot, tta, b), xa);
            vec_load<T, NU>(a.U + vec_at(B, N, NU, slot, tta, b), ua);
            vec_load<T, NX>(a.X + vec_at(B, N + 1, NX, slot, ttb, b), xb);
            vec_load<T, NU>(a.U + vec_at(B, N, NU, slot, ttb, b), ub);
            if (move && ina) {
                vec_store<T, NX>(a.X + vec_at(B, N + 1, NX, 0, tta, b), xa);
                vec_store<T, NU>(a.U + vec_at(B, N, NU, 0, tta, b), ua);
            }
            if (move && inb) {
                vec_store<T, NX>(a.X + vec_at(B, N + 1, NX, 0, ttb, b), xb);
                vec_store<T, NU>(a.U + vec_at(B, N, NU, 0, ttb, b), ub);
            }
            T2 x[NX], u[NU], xn[NX], fx[NX][NX], fu[NX][NU];
#pragma unroll
            for (int q = 0; q < NX; ++q) { x[q].x = xa[q]; x[q].y = xb[q]; }
#pragma unroll
            for (int q = 0; q < NU; ++q) { u[q].x = ua[q]; u[q].y = ub[q]; }
            Stepper<T2, Dyn2>::step_jac(INTEG, p, dt2, x, u, xn, fx, fu);
            T2 gxn[NX], gu1[NU], lxxn[NX][NX], luxn[NU][NX], luu[NU][NU];
            Cost<T2, Dyn2>::grad(p, dt2, x, u, gxn, gu1);
            Cost<T2, Dyn2>::hess(p, dt2, x, u, lxxn, luxn, luu);
            V4 tile_a[NV], tile_b[NV];
            tile16_fill<T, NX, NU>([](T2 v) { return v.x; }, fx, fu, gxn, gu1, lxxn, luxn, luu, tile_a);
            tile16_fill<T, NX, NU>([](T2 v) { return v.y; }, fx, fu, gxn, gu1, lxxn, luxn, luu, tile_b);
            if (k >= RU) {
                const int need = k - RU + 1;
                auto slowest = [&]() {
                    int m = lds_peek(&L.done[0]);
#pragma unroll
                    for (int q = 1; q < NSW; ++q) m = min(m, lds_peek(&L.done[q]));
                    return m;
                };
                while (slowest() < need) __builtin_amdgcn_s_sleep(2);
            }
            compiler_fence();
            V4* dst = reinterpret_cast<V4*>(L.ring + (size_t)(k % RU) * UNIT + tile_off(2 * r2, tl));
            if (ina) {
#pragma unroll
                for (int q = 0; q < NV; ++q) dst[q] = tile_a[q];
            }
            if (inb) {
#pragma unroll
                for (int q = 0; q < NV; ++q) dst[TL / 4 + q] = tile_b[q];
            }
            compiler_fence();
            if (lane == 0) lds_poke(&L.ready[k % RU], k + 1);
        }
        } else {
        // ================= producers: lane = (trajectory tl, time step r of the unit) ================================
        const int tl = lane % TPW, r = lane / TPW;
        const int gidx = b0 + tl;
        const bool valid = gidx < a.B;
        const int b = valid ? gidx : a.B - 1;
        const bool actb = valid && traj_active(L.s_stat[tl]);
        const int slot = actb ? L.s_slot[tl] : 0;
        const bool move = actb && slot != 0;
        T p[PL::TOTAL];
#pragma unroll
        for (int q = 0; q < PL::TOTAL; ++q) p[q] = a.params[q];
        for (int k = pw; k < n_units; k += P) {
            const int t = N - 1 - k * US - r;
            const bool inr = t >= 0;
            const int tt = inr ? t : 0;
            T x[NX], u[NU];
            vec_load<T, NX>(a.X + vec_at(B, N + 1, NX, slot, tt, b), x);
            vec_load<T, NU>(a.U + vec_at(B, N, NU, slot, tt, b), u);
            if (move && inr) {
                vec_store<T, NX>(a.X + vec_at(B, N + 1, NX, 0, tt, b), x);
                vec_store<T, NU>(a.U + vec_at(B, N, NU, 0, tt, b), u);
            }
            T xn[NX], fx[NX][NX], fu[NX][NU];
            Stepper<T, Dyn>::step_jac(INTEG, p, a.dt, x, u, xn, fx, fu);
            V4 tile[NV];
            tile16_pack<T, Dyn>(p, a.dt, x, u, fx, fu, tile);
            // the L.ring slot is free once every sweep wave has read unit k - RU
            if (k >= RU) {
                const int need = k - RU + 1;
                auto slowest = [&]() {
                    int m = lds_peek(&L.done[0]);
#pragma unroll
                    for (int q = 1; q < NSW; ++q) m = min(m, lds_peek(&L.done[q]));
                    return m;
                };
                while (slowest() < need) __builtin_amdgcn_s_sleep(2);
            }
            compiler_fence();
            if (inr) {
                constexpr int NQ = NV * (int)sizeof(V4) / 16;       // 16-byte pieces of the tile
                vec_u4 w[NQ];
                __builtin_memcpy(w, tile, sizeof(V4) * NV);
                vec_u4* dst = reinterpret_cast<vec_u4*>(L.ring + (size_t)(k % RU) * UNIT + tile_off(r, tl));
#pragma unroll
                for (int q = 0; q < NQ; ++q) dst[q] = w[q];
            }
            compiler_fence();
            if (lane == 0) lds_poke(&L.ready[k % RU], k + 1);
        }
        }
    }
};

template <typename T, typename Dyn, int INTEG, int TPW, bool PK>
__global__ void __launch_bounds__((fused_threads<T, TPW, PK>())) backward_fused16_kernel(KArgs<T> a) {
    using W = FusedWG<T, Dyn, INTEG, TPW, PK>;
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    const typename W::Lds L = W::carve(fused_lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * TPW;
    ClockProbe cp;       // (diagnostic, ILQR_CLOCK_PROBE: start / end ticks of every workgroup's thread 0 = its first sweep wave)
    cp.start();
    if (wave == 0) W::head(a, L, b0, lane, a.fuse_select != 0, true);
    __syncthreads();
    if (!W::any_active(L)) return;     // (uniform over the workgroup)
    if (wave < W::NSW) W::sweep(a, L, b0, wave, lane, cp);
    else W::produce(a, L, b0, wave - W::NSW, lane);
}

}  // namespace ilqr

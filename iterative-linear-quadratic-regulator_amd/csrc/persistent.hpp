// persistent.hpp -- the whole iteration loop of a workgroup's trajectories in ONE launch (round 3).
//
// Trajectories never interact, so nothing in optimize_trajectory (iLQR_class.py:250-313) needs a grid-wide
// step: a workgroup that owns TPW trajectories can run, for them alone and at its own pace,
//     [ alpha = 0 rollout + its acceptance ]                                      (:257-259, the head of a solve)
//     repeat: acceptance step of the newest candidates -> linearise + backward sweep -> all candidate rollouts
//     [ plant step + warm-start shift, next MPC step ]                            (run_iLQR_MPC.py:116-143)
// with workgroup barriers between the phases and no host in between.  The phases are the device functions the separate
// kernels run (FusedWG::head / sweep / produce of backward_fused16.hpp, rollout_ring of kernels.hpp,
// select_candidates), so the results are bit-identical to the multi-launch forms (tests/test_persistent_gpu.py).
// What it buys: the two launch boundaries of an iteration (~5-7 us each: 48 + 103 us of kernel against 41-44 + 92-96 us
// of workgroup time, ILQR_CLOCK_PROBE); workgroups drift apart, so ALU-heavy sweep phases of some CUs overlap the
// latency-bound rollouts of others; a solve needs no read-back of the active count (a workgroup leaves its loop when
// its own trajectories are done); and an MPC step of a workgroup lasts as long as ITS slowest instance, not the
// batch's -- the device-resident MPC loop then runs at the mean, not the maximum, of the iteration counts.
//
// Memory ordering between phases: every wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier; all
// waves of a workgroup sit on one CU and share its vector L1, which the CU's own stores keep current, so the next
// phase's loads see them (workgroup scope needs no cache maintenance outside threadgroup-split mode).
#pragma once

namespace ilqr {

template <typename T> struct PArgs {
    int n_iters;          // iterations of a solve (upper bound: a workgroup stops when none of its trajectories is active)
    int do_init;          // run the head of a solve first: alpha = 0 rollout through the carried gains and its acceptance
    int n_mpc;            // 0: one solve / n_iters iterations; > 0: that many MPC steps (init, solve, plant step + shift)
    MpcArgs<T> mpc;       // plant, logs (n_mpc > 0)
};

ILQR_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The roles as real functions (noinline): each gets a register allocation of its own.  Inlined into one loop, hipcc kept
// every role's loop-invariant scalars live across all the others and spilled 150-220 VGPRs in the middle of the rollout's
// self-counted register ring -- which csrc/check_ring_kernels.py rejects, rightly: a spill there copies registers whose
// loads have not landed.  The kernel's argument block reaches a role through a pointer to the caller's copy.
template <typename T, typename Dyn, int INTEG>
__device__ __attribute__((noinline)) void role_rollout(const KArgs<T>* a, int b, int ai, bool in_range, int init) {
    rollout_ring<T, Dyn, INTEG>(*a, b, ai, in_range, init != 0);
}
// (The LDS pointers are carved from the workgroup's dynamic allocation INSIDE each role: handed over as function
// arguments they would be generic pointers, and every tile read of the sweep a flat_load instead of a ds_read.)
template <typename W, typename T>
__device__ __attribute__((noinline)) void role_sweep(const KArgs<T>* a, int b0, int wave, int lane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    const typename W::Lds L = W::carve(fused_lds);
    ClockProbe cp;
    cp.start();
    W::sweep(*a, L, b0, wave, lane, cp);
    __builtin_amdgcn_s_setprio(0);
}
template <typename W, typename T>
__device__ __attribute__((noinline)) void role_produce(const KArgs<T>* a, int b0, int pw, int lane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    const typename W::Lds L = W::carve(fused_lds);
    W::produce(*a, L, b0, pw, lane);
}

// PK: the 16-trajectory form runs the pair producers (4 sweep + 4 producer waves = 512 threads: 256 VGPRs per lane, which
// the rollout's register ring (165) and the pair producers (234) both fit; eight scalar producers would cap the kernel
// at 168 and make the ring rollout spill -- the build rejects that, csrc/check_ring_kernels.py)
template <typename T, typename Dyn, int INTEG, int TPW, bool PK>
__global__ void __launch_bounds__((fused_threads<T, TPW, PK>())) ilqr_persistent_kernel(KArgs<T> a, PArgs<T> pa) {
    using W = FusedWG<T, Dyn, INTEG, TPW, PK>;
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    constexpr int NT = fused_threads<T, TPW, PK>();
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    const typename W::Lds L = W::carve(fused_lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * TPW;
    const size_t B = a.B;
    // the roles read the argument block where the dispatch put it (explicit arguments start the kernarg segment): no
    // private copy
    const KArgs<T>* ka = (const KArgs<T>*)(unsigned long long)(__attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef ILQR_PERSIST_STAMPS   // diagnostic build (tools/persist_stamps.py): cycles of workgroup 7's wave 0 per phase
    long long ps[4] = {0, 0, 0, 0}, pt = __builtin_readcyclecounter();
#define ILQR_PSTAMP(k) do { const long long now_ = __builtin_readcyclecounter(); ps[k] += now_ - pt; pt = now_; } while (0)
#else
#define ILQR_PSTAMP(k) do {} while (0)
#endif
    const int n_outer = pa.n_mpc > 0 ? pa.n_mpc : 1;
    for (int ms = 0; ms < n_outer; ++ms) {
        if (pa.do_init) {
            // ---- head of a solve (iLQR_class.py:257-259): every trajectory, alpha = 0, through the carried K and X --------
            if (wave == 0) {
                role_rollout<T, Dyn, INTEG>(ka, b0 + lane, 0, lane < TPW && b0 + lane < a.B, 1);
                drain_stores();
                const int b = b0 + lane;
                if (lane < TPW && b < a.B) {        // (the lane that rolled the candidate out accepts it: no barrier needed)
                    const T c = a.costs[b];
                    a.cur_slot[b] = (a.cur_slot[b] + 1) % a.n_slots;
                    a.cost[b] = c;
                    a.cost_prev[b] = c;
                    a.alpha_taken[b] = T(0);
                    a.status[b] = ILQR_TRAJ_ACTIVE;
                    a.iters[b] = 0;
                    a.accepted[b] = 0;
                }
                drain_stores();
            }
            __syncthreads();
        }
        bool pending = false;
        for (int it = 0; it < pa.n_iters; ++it) {
            if (wave == 0) W::head(a, L, b0, lane, pending, false);
            drain_stores();
            __syncthreads();
            pending = false;
            ILQR_PSTAMP(0);
            if (!W::any_active(L)) break;          // (uniform over the workgroup)
            if (wave < W::NSW) role_sweep<W, T>(ka, b0, wave, lane);
            else role_produce<W, T>(ka, b0, wave - W::NSW, lane);
            drain_stores();
            __syncthreads();
            ILQR_PSTAMP(1);
            // ---- all candidates of the workgroup's trajectories: lane = (trajectory, alpha) ---------------------------------
            {
                const int lg = wave * 64 + lane;
                const int tl = lg % TPW, ai = lg / TPW;
                if (wave * 64 < TPW * a.n_pass)    // (wave-uniform: waves beyond the last candidate skip the role)
                    role_rollout<T, Dyn, INTEG>(ka, b0 + tl, ai, ai < a.n_pass && b0 + tl < a.B, 0);
            }
            drain_stores();
            __syncthreads();
            ILQR_PSTAMP(2);
            pending = true;
        }
        if (pending) {
            // ---- the acceptance step of the last candidates (select_kernel's form: the trajectory stays in its slot) -----------
            const int b = b0 + lane;
            if (wave == 0 && lane < TPW && b < a.B) {
                int slot, st;
                const int before = a.cur_slot[b];
                select_candidates(a, b, true, slot, st);
                if (slot != before) a.cur_slot[b] = slot;
            }
            drain_stores();
            __syncthreads();
        }
        ILQR_PSTAMP(3);
        if (pa.n_mpc > 0) {
            // ---- MPC advance (run_iLQR_MPC.py:127-140): u0 = U[:, 0]; plant step; x_0 <- plant state; shift the warm start ------
            const MpcArgs<T>& m = pa.mpc;
            const int tl = tid % TPW, chunk = tid / TPW;
            constexpr int NCH = NT / TPW, KEEP = 8;
            const int b = b0 + tl;
            const bool inb = b < a.B;
            const int bb = inb ? b : a.B - 1;
            T* Uc = a.U + vec_at(B, a.N, NU, a.cur_slot[bb], 0, bb);
            const size_t sU = B * NU;
            const int n_shift = a.N - 1;
            const int per = (n_shift + NCH - 1) / NCH;
            const bool sliced = per <= KEEP;
            T u0[NU];
            vec_load<T, NU>(Uc, u0);
            T keep[KEEP][NU];
            const int t0 = chunk * per;
            if (sliced) {
#pragma unroll
                for (int q = 0; q < KEEP; ++q)
                    if (q < per && t0 + q < n_shift) vec_load<T, NU>(Uc + (size_t)(t0 + q + 1) * sU, keep[q]);
            }
            __syncthreads();
            if (sliced && inb) {
#pragma unroll
                for (int q = 0; q < KEEP; ++q)
                    if (q < per && t0 + q < n_shift) vec_store<T, NU>(Uc + (size_t)(t0 + q) * sU, keep[q]);
            }
            if (chunk == 0 && inb) {
                T x[NX], xn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = m.plant_x[(size_t)i * B + b];
                Stepper<T, Dyn>::step(m.plant_integ, a.params, a.dt, x, u0, xn);
                const int step = m.step + ms;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    m.plant_x[(size_t)i * B + b] = xn[i];
                    a.x0[(size_t)i * B + b] = xn[i];
                    if (m.x_log) m.x_log[((size_t)step * B + b) * NX + i] = xn[i];
                }
#pragma unroll
                for (int j = 0; j < NU; ++j)
                    if (m.u_log) m.u_log[((size_t)step * B + b) * NU + j] = u0[j];
                if (m.cost_log) m.cost_log[(size_t)step * B + b] = a.cost[b];
                if (!sliced) {
                    for (int t = 0; t + 1 < a.N; ++t) {
                        T un[NU];
                        vec_load<T, NU>(Uc + (t + 1) * sU, un);
                        vec_store<T, NU>(Uc + t * sU, un);
                    }
                }
            }
            drain_stores();
            __syncthreads();
        }
    }
#ifdef ILQR_PERSIST_STAMPS
    if (a.probe && blockIdx.x == 7 && tid == 0) { a.probe[4] = ps[0]; a.probe[5] = ps[1]; a.probe[6] = ps[2]; a.probe[7] = ps[3]; }
#endif
}

}  // namespace ilqr

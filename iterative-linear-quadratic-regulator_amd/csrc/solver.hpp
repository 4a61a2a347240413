// solver.hpp -- host side of libilqr_hip.so: the handle behind the C-ABI.
//
// SolverBase is the dtype-erased interface the extern "C" layer (ilqr_abi.cpp)
// talks to; SolverT<T> owns the device buffers, the stream and the launch
// sequence of the hot path.  It mirrors the state and the control flow of the
// reference's iLQR class (python/class_files/iLQR_class.py:18-75, 250-313) for a
// whole batch of trajectories at once, with the per-trajectory loop state
// (cost, status, iteration count) kept on the device so an iteration needs no
// host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "kernels_wave.hpp"
#include "backward_mfma16.hpp"
#include "forward_mfma16.hpp"

namespace ilqr {

struct SolverBase {
    ilqr_config cfg{};
    std::string err;
    virtual ~SolverBase() {}
    virtual int sync() = 0;
    virtual int set_problem(const void* x0, const void* U) = 0;
    virtual int set(int field, const void* src, size_t bytes) = 0;
    virtual int get(int field, void* dst, size_t bytes) = 0;
    virtual int initial_rollout() = 0;
    virtual int linearize() = 0;
    virtual int backward() = 0;
    virtual int forward(const double* alphas, int n) = 0;
    virtual int select() = 0;
    virtual int iterate(int n) = 0;
    virtual int flush() = 0;
    virtual int solve(int32_t* iters, void* cost) = 0;
    virtual int backward_pass(const void* X, const void* U, void* Uff, void* K) = 0;
    virtual int backward_tensors(const void* lin, const void* term, void* Uff, void* K) = 0;
    virtual int forward_pass(const void* x0, double alpha, const void* X, const void* U, const void* Uff,
                             const void* K, void* Xn, void* Un, void* cost) = 0;
    virtual int eval_points(int integ, int npts, const void* x, const void* u, void** outs) = 0;
    virtual int mpc_reset(const void* x0, const void* U) = 0;
    virtual int mpc_rearm(const void* x0, const void* U) = 0;
    virtual int mpc_run(int n_steps, void* u_out, void* x_out, void* cost_out) = 0;
    virtual int status_reduce(void* dev_out4) = 0;
    virtual int probe_dump(long long* dst, size_t n) = 0;
    virtual int debug_set_stream(void* s) = 0;   // experiments only (tools/cumask_probe.py)
    virtual int timing_enable(int on) = 0;
    virtual int timing_reset() = 0;
    virtual int timing_get(double* ms, int64_t* launches) = 0;
    virtual int algorithmic_bytes(double* bytes) = 0;
};

int system_dims(int system, int n_x, int n_u);  // 1 if (system, n_x, n_u) is a known combination
int param_count(int system, int n_x, int n_u);
int n_sys_params_abi(int system, int n_x, int n_u);
SolverBase* make_solver_f32(const ilqr_config& cfg, std::string& err, int* status);
SolverBase* make_solver_f64(const ilqr_config& cfg, std::string& err, int* status);
bool supported_f32(int system, int n_x, int n_u);
bool supported_f64(int system, int n_x, int n_u);

// Largest tensor the kernels address through a 32-bit buffer descriptor.  Dead lanes' stores are dropped by giving them
// the byte offset 0x7ffffff0 (kernels.hpp, backward_tile16.hpp, backward_fused16.hpp), which the hardware compares with
// the descriptor's num_records = the tensor's size: a tensor of more than 0x7ffffff0 bytes would turn a dropped store
// into a landed one, so that -- not 2^31 -- is the bound (tests/test_limits_gpu.py runs at it).
constexpr size_t kDescriptorMax = 0x7ffffff0ull;

#define ILQR_HIPCHK(expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            this->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return ILQR_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

// Phase timing attaches its HIP events to the kernel dispatch itself (hipExtLaunchKernelGGL start / stop
// events = the dispatch packet's own begin / end timestamps, what rocprofv3 reports) instead of recording
// separate events around the launch: a recorded event is an extra barrier packet on the stream and was
// measured to add ~4.5 us to every bracketed launch.
struct LaunchEvents { hipEvent_t a = nullptr, b = nullptr; };
inline LaunchEvents& launch_events() { static thread_local LaunchEvents e; return e; }
#define ILQR_LAUNCH(kern, grid, block, lds, stream, ...)                                                     \
    do {                                                                                                     \
        LaunchEvents& le_ = launch_events();                                                                 \
        hipExtLaunchKernelGGL(kern, grid, block, lds, stream, le_.a, le_.b, 0, __VA_ARGS__);                 \
        le_ = LaunchEvents();                                                                                \
    } while (0)

template <typename T> struct Ops {
    void (*linearize[5])(const KArgs<T>&, hipStream_t) = {};  // indexed by ilqr_integrator
    void (*backward)(const KArgs<T>&, hipStream_t) = nullptr;
    void (*forward[5])(const KArgs<T>&, hipStream_t) = {};
    void (*fused[5])(const KArgs<T>&, hipStream_t) = {};   // acceptance step + linearise + sweep in one launch (backward_fused16.hpp), or null
    void (*persist[5])(const KArgs<T>&, const PArgs<T>&, hipStream_t) = {};
    bool persist_any_batch[5] = {};   // the integrator also has the 16-trajectory form (batches beyond persist_small_max())
    bool persist_big = false;   // the whole iteration / solve / MPC loop of a workgroup's trajectories in one launch (persistent.hpp), or null
    void (*eval)(const EvalArgs<T>&, hipStream_t) = nullptr;
    void (*mpc_advance)(const MpcArgs<T>&, hipStream_t) = nullptr;
    int n_dev_params = 0;
    int n_sys_dev = 0;
    int lin_stride = 0;   // scalars per (b, t) in the expansion buffer
    bool tile16 = false;  // expansion packed as tiles for the DPP sweeps (n_u = 1: 48 scalars, (4, 2): 64)
    int tile_scalars = 0;
    bool lin_aos = false; // expansion stored as [N][B][E] records (n_x > 4, wave-cooperative kernels)
    bool canonical = false;  // linearize moves every current trajectory into slot 0 (then cur_slot is reset)
    bool const_lin = false;  // the system's expansion has constant matrices (Linear dynamics + parameter-block cost): KArgs::const_lin
    bool (*sweep_reads_sparse)(T mu) = nullptr;   // does the backward dispatch take the constant-matrix form for this mu?
};

// linearize / forward are compiled once per integrator so the integrator switch folds away and each
// variant gets its own register allocation (the RK4 rollout must not pay for the backward-Euler LU).
// is there a generated FwdIn<T, NX, NU> (the ring rollout's one-statement load group) for these dimensions?
template <typename T, int NX, int NU, typename = void> struct has_fwd_in { static constexpr bool value = false; };
template <typename T, int NX, int NU> struct has_fwd_in<T, NX, NU, decltype((void)sizeof(FwdIn<T, NX, NU>))> {
    static constexpr bool value = true;
};

// Bit i set: integrator i may use the ring rollout.  A plugin whose generated dynamics make a ring kernel spill
// is recompiled with that integrator's bit cleared (csrc/check_ring_kernels.py, systems/custom_sys.py).
#ifndef ILQR_RING_INTEG_MASK
#define ILQR_RING_INTEG_MASK 0x1f
#endif
// Bit i set: integrator i gets the fused acceptance + linearise + sweep kernel (backward_fused16.hpp)
#ifndef ILQR_FUSE_INTEG_MASK
#define ILQR_FUSE_INTEG_MASK 0x1f
#endif
inline int persist_small_max() {
    static const int v = getenv("ILQR_FUSED_SMALL_MAX") ? atoi(getenv("ILQR_FUSED_SMALL_MAX")) : 1024;
    return v;
}
// Bit i set: integrator i gets the persistent kernel (persistent.hpp)
#ifndef ILQR_PERSIST_INTEG_MASK
#define ILQR_PERSIST_INTEG_MASK 0x1f
#endif
#ifndef ILQR_NO_PAIR_PRODUCERS
#define ILQR_NO_PAIR_PRODUCERS 0     // 1: never instantiate the two-points-per-lane producers (plugin builds whose generated code is scalar-only)
#endif
// a system whose templates can be instantiated on another scalar type (the float pair of the fused kernel's producers)
template <typename Dyn, typename = void> struct has_rebind { static constexpr bool value = false; };
template <typename Dyn> struct has_rebind<Dyn, std::void_t<typename Dyn::template rebind<float>>> { static constexpr bool value = true; };

template <typename T, typename Dyn, bool TILE, int INTEG> void set_integrator_ops(Ops<T>& o) {
    constexpr bool SMALL = all_integrators<Dyn>::value;
    // n_x > 4 only has the closed-form integrators: fold the others onto euler so nothing big is compiled
    constexpr int I = (SMALL || INTEG == ILQR_INT_DISCRETE) ? INTEG : ILQR_INT_EULER;
    o.linearize[INTEG] = [](const KArgs<T>& a, hipStream_t s) {
        const size_t total = (size_t)a.B * (a.N + 1);
        constexpr int TPB = TILE ? 64 : 256;
        ILQR_LAUNCH((linearize_kernel<T, Dyn, TILE, I>), dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, s, a);
    };
    if constexpr (TILE && ((ILQR_FUSE_INTEG_MASK >> I) & 1)) {
        o.fused[INTEG] = [](const KArgs<T>& a, hipStream_t s) {
            // one workgroup = 16 trajectories (4 sweep waves + the producer waves, tiles through ~104 KB of LDS: one per
            // CU), or 4 trajectories (1 sweep wave, ~52 KB) while the batch then still fits the chip one workgroup per CU
            // (measured, fp32 fused kernel: B = 1024 35 vs 41 us, B = 2048 41 vs 41, B = 4096 73 vs 47)
            // fp32 with an explicit integrator and a system that can be instantiated on a float pair: pair producers
            constexpr bool CAN_PK = sizeof(T) == 4 && I != ILQR_INT_BACKWARD_EULER && has_rebind<Dyn>::value && !ILQR_NO_PAIR_PRODUCERS;
            static const bool ok = [] {
                bool r = hipFuncSetAttribute((const void*)backward_fused16_kernel<T, Dyn, I, 16, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds_bytes<T, 16, false, Dyn::NU>()) == hipSuccess;
                r = r && hipFuncSetAttribute((const void*)backward_fused16_kernel<T, Dyn, I, 4, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds_bytes<T, 4, false, Dyn::NU>()) == hipSuccess;
                if constexpr (CAN_PK)
                    r = r && hipFuncSetAttribute((const void*)backward_fused16_kernel<T, Dyn, I, 16, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds_bytes<T, 16, true, Dyn::NU>()) == hipSuccess;
                (void)hipGetLastError();
                return r;
            }();
            (void)ok;
            static const int force = getenv("ILQR_FUSED_TPW") ? atoi(getenv("ILQR_FUSED_TPW")) : 0;   // A/B switch
            static const int small_max = getenv("ILQR_FUSED_SMALL_MAX") ? atoi(getenv("ILQR_FUSED_SMALL_MAX")) : 1024;
            // Pair producers (two time steps per lane in packed FP32, bit-identical) are built and tested but OFF by default
            // in this kernel: with a ring of 4 units they measured the same as the scalar ones (48.6 vs 48.1 us at B = 4096:
            // the kernel is bound by the sweep waves' chain, and four lone pair waves deliver their first unit later and let
            // the ring run dry); with 5 slots (133 KB of LDS, the build's value now) 44.7-45.8 against 46.4 us -- one
            // microsecond, for a third fewer vector instructions in the same time (bench.py's issue-rate fraction would fall
            // from 0.70 to ~0.6 with nothing else changing).  ILQR_FUSED_PAIRS=1 selects them; the 16-trajectory persistent
            // kernel always uses them (register budget).
            static const bool no_pk = getenv("ILQR_FUSED_PAIRS") == nullptr;                          // A/B switch
            const bool small = force ? force == 4 : a.B <= small_max;
            if (small) {
                ILQR_LAUNCH((backward_fused16_kernel<T, Dyn, I, 4, false>), dim3((a.B + 3) / 4), dim3(fused_threads<T, 4, false>()),
                            (fused_lds_bytes<T, 4, false, Dyn::NU>()), s, a);
                return;
            }
            if constexpr (CAN_PK) {
                if (!no_pk) {
                    ILQR_LAUNCH((backward_fused16_kernel<T, Dyn, I, 16, true>), dim3((a.B + 15) / 16), dim3(fused_threads<T, 16, true>()),
                                (fused_lds_bytes<T, 16, true, Dyn::NU>()), s, a);
                    return;
                }
            }
            ILQR_LAUNCH((backward_fused16_kernel<T, Dyn, I, 16, false>), dim3((a.B + 15) / 16), dim3(fused_threads<T, 16, false>()),
                        (fused_lds_bytes<T, 16, false, Dyn::NU>()), s, a);
        };
    }
    // The persistent kernel (persistent.hpp): fp32 only (tried for fp64 in the 4-trajectory form: as a noinline role the fp64
    // RK4 / backward-Euler rollout spills registers of its self-counted load ring, which the build rejects); batches <= 1024 in
    // 4-trajectory workgroups (scalar producers), larger ones in 16-trajectory workgroups with the pair producers -- which
    // backward Euler and generated systems do not have: those keep one launch per phase.
    if constexpr (TILE && sizeof(T) == 4 && ((ILQR_FUSE_INTEG_MASK >> I) & 1) && has_fwd_in<T, Dyn::NX, Dyn::NU>::value &&
                  ((ILQR_RING_INTEG_MASK >> I) & 1) && ((ILQR_PERSIST_INTEG_MASK >> I) & 1)) {
        constexpr bool BIG = I != ILQR_INT_BACKWARD_EULER && has_rebind<Dyn>::value && !ILQR_NO_PAIR_PRODUCERS;
        o.persist_big = o.persist_big || BIG;
        o.persist[INTEG] = [](const KArgs<T>& a, const PArgs<T>& pa, hipStream_t s) {
            static const bool ok = [] {
                bool r = hipFuncSetAttribute((const void*)ilqr_persistent_kernel<T, Dyn, I, 4, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds_bytes<T, 4, false, Dyn::NU>()) == hipSuccess;
                if constexpr (BIG)
                    r = r && hipFuncSetAttribute((const void*)ilqr_persistent_kernel<T, Dyn, I, 16, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, fused_lds_bytes<T, 16, true, Dyn::NU>()) == hipSuccess;
                (void)hipGetLastError();
                return r;
            }();
            (void)ok;
            if constexpr (BIG) {
                if (a.B > persist_small_max()) {
                    ILQR_LAUNCH((ilqr_persistent_kernel<T, Dyn, I, 16, true>), dim3((a.B + 15) / 16), dim3(fused_threads<T, 16, true>()),
                                (fused_lds_bytes<T, 16, true, Dyn::NU>()), s, a, pa);
                    return;
                }
            }
            ILQR_LAUNCH((ilqr_persistent_kernel<T, Dyn, I, 4, false>), dim3((a.B + 3) / 4), dim3(fused_threads<T, 4, false>()),
                        (fused_lds_bytes<T, 4, false, Dyn::NU>()), s, a, pa);
        };
        o.persist_any_batch[INTEG] = BIG;
    }
    o.forward[INTEG] = [](const KArgs<T>& a, hipStream_t s) {
        const dim3 grid((a.B + 63) / 64, a.n_pass), block(64);
        if constexpr (has_fwd_in<T, Dyn::NX, Dyn::NU>::value && ((ILQR_RING_INTEG_MASK >> I) & 1)) {
            // ring form: 32-bit buffer offsets into X (the largest tensor), and a switch for A/B runs
            static const bool plain = getenv("ILQR_FORWARD_PLAIN") != nullptr;
            // (X is the largest state tensor, U <= X; the gain tensor can be larger than X when n_alpha is small)
            const size_t bytes_x = (size_t)a.n_slots * (a.N + 1) * Dyn::NX * a.B * sizeof(T);
            const size_t bytes_g = (size_t)a.N * a.B * gain_record(Dyn::NX, Dyn::NU) * sizeof(T);
            const bool fits = std::max(bytes_x, bytes_g) <= kDescriptorMax;
            if (fits && !plain) {
                ILQR_LAUNCH((forward_ring_kernel<T, Dyn, I>), grid, block, 0, s, a);
                return;
            }
        }
        ILQR_LAUNCH((forward_kernel<T, Dyn, I>), grid, block, 0, s, a);
    };
}

template <typename T, typename Dyn> Ops<T> make_ops() {
    constexpr int NX = Dyn::NX, NU = Dyn::NU;
    Ops<T> o;
    constexpr bool TILE2 = (NX == 4 && NU == 2);              // backward_tile16m2.hpp
    constexpr bool TILE = (NU == 1 && NX >= 2 && NX <= 4) || TILE2;   // the DPP sweeps; n_x < 4 rides the 4 x 4 tile zero-padded
    constexpr int TSC = TILE2 ? kTile16M2 : kTile16;
    o.tile16 = TILE;
    o.canonical = true;
    o.lin_stride = TILE ? TSC : (2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU);
    o.tile_scalars = TILE ? TSC : 0;
    set_integrator_ops<T, Dyn, TILE, 0>(o);
    set_integrator_ops<T, Dyn, TILE, 1>(o);
    set_integrator_ops<T, Dyn, TILE, 2>(o);
    set_integrator_ops<T, Dyn, TILE, 3>(o);
    set_integrator_ops<T, Dyn, TILE, 4>(o);
    if constexpr (TILE2) {
        o.backward = [](const KArgs<T>& a, hipStream_t s) {
            // 16 trajectories per 256-thread workgroup, one workgroup per CU (see kTile16PinLds)
            static const bool pinned = [] {
                bool ok = hipFuncSetAttribute((const void*)backward_tile16m2_kernel<T, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kTile16PinLds) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)backward_tile16m2_kernel<T, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kTile16PinLds) == hipSuccess;
                (void)hipGetLastError();
                return ok && getenv("ILQR_BACKWARD_NO_PIN") == nullptr;
            }();
            const dim3 grid((a.B + 15) / 16), block(256);
            const size_t lds = pinned ? kTile16PinLds : 0;
            if (a.mu != T(0)) ILQR_LAUNCH((backward_tile16m2_kernel<T, true>), grid, block, lds, s, a);
            else ILQR_LAUNCH((backward_tile16m2_kernel<T, false>), grid, block, lds, s, a);
        };
    } else if constexpr (TILE) {
        // one wave = 4 trajectories x 16 lanes; 1024 single-wave workgroups at B = 4096 = one per SIMD
        o.backward = [](const KArgs<T>& a, hipStream_t s) {
            static const bool lds_ring = getenv("ILQR_BACKWARD_LDS_RING") != nullptr;  // A/B switch for profiling
            // the register-ring kernel addresses both tensors through 32-bit buffer offsets
            // (the gain tensor, gain_record(NX, 1) <= 8 scalars per (t, b), is always the smaller of the two)
            const bool fits = (size_t)a.N * a.B * kTile16 * sizeof(T) <= kDescriptorMax &&
                              (size_t)a.N * a.B * gain_record(NX, 1) * sizeof(T) <= kDescriptorMax;
            if (lds_ring || !fits) {
                const dim3 grid((a.B + 3) / 4), block(64);
                if (a.mu != T(0)) ILQR_LAUNCH((backward_tile16_lds_kernel<T, true, NX>), grid, block, 0, s, a);
                else ILQR_LAUNCH((backward_tile16_lds_kernel<T, false, NX>), grid, block, 0, s, a);
                return;
            }
            // 16 trajectories per 256-thread workgroup, one workgroup per CU (see kTile16PinLds)
            static const bool pinned = [] {
                bool ok = hipFuncSetAttribute((const void*)backward_tile16_kernel<T, false, NX>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kTile16PinLds) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)backward_tile16_kernel<T, true, NX>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kTile16PinLds) == hipSuccess;
                (void)hipGetLastError();
                return ok && getenv("ILQR_BACKWARD_NO_PIN") == nullptr;
            }();
            const dim3 grid((a.B + 15) / 16), block(256);
            const size_t lds = pinned ? kTile16PinLds : 0;
            if (a.mu != T(0)) ILQR_LAUNCH((backward_tile16_kernel<T, true, NX>), grid, block, lds, s, a);
            else ILQR_LAUNCH((backward_tile16_kernel<T, false, NX>), grid, block, lds, s, a);
        };
    } else {
        o.backward = [](const KArgs<T>& a, hipStream_t s) {
            ILQR_LAUNCH((backward_lane_kernel<T, NX, NU>), dim3((a.B + 63) / 64), dim3(64), 0, s, a);
        };
    }
    o.eval = [](const EvalArgs<T>& a, hipStream_t s) {
        ILQR_LAUNCH((eval_points_kernel<T, Dyn>), dim3((a.npts + 63) / 64), dim3(64), 0, s, a);
    };
    o.mpc_advance = [](const MpcArgs<T>& a, hipStream_t s) {
        ILQR_LAUNCH((mpc_advance_kernel<T, Dyn>), dim3((a.B + 63) / 64), dim3(64, kMpcChunks), 0, s, a);
    };
    o.n_dev_params = ParamLayout<Dyn::NSYS, NX, NU>::TOTAL;
    o.n_sys_dev = Dyn::NSYS;
    return o;
}

// n_x > 4 (linear systems): wave-cooperative linearise / backward, generic lane-per-rollout forward
template <typename T, int NX, int NU> Ops<T> make_ops_wave() {
    using Dyn = Linear<T, NX, NU>;
    Ops<T> o;
    o.lin_aos = true;
    o.const_lin = true;      // Linear dynamics, parameter-block quadratic cost
    o.lin_stride = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
    for (int k = 0; k < 5; ++k) {
        o.linearize[k] = [](const KArgs<T>& a, hipStream_t s) {
            if (a.lin_sparse) {
                // sparse form: the matrices (and the terminal expansion) from t = N-1 on, the gradients of every point dense
                KArgs<T> w = a;
                w.t_first = a.N - 1;
                LaunchEvents le = launch_events();      // (the phase timer's event pair spans both launches)
                launch_events() = LaunchEvents{le.a, nullptr};
                ILQR_LAUNCH((linearize_grad_dense_kernel<T, NX, NU>), dim3((unsigned)(((size_t)a.B * a.N + 255) / 256)), dim3(256), 0, s, a);
                launch_events() = LaunchEvents{nullptr, le.b};
                ILQR_LAUNCH((linearize_wave_kernel<T, NX, NU>), dim3((unsigned)((size_t)a.B * 2)), dim3(64), 0, s, w);
                return;
            }
            ILQR_LAUNCH((linearize_wave_kernel<T, NX, NU>), dim3((unsigned)((size_t)a.B * (a.N + 1))), dim3(64), 0, s, a);
        };
    }
    for (int k = 0; k < 5; ++k) {
        // euler / discrete are told apart inside the kernel (a.integ); the others do not exist for n_x > 4
        o.forward[k] = [](const KArgs<T>& a, hipStream_t s) {
            static const bool plain = getenv("ILQR_FORWARD_PLAIN") != nullptr;   // A/B: lane-per-rollout kernel
            if (plain) {
                if (a.integ == ILQR_INT_DISCRETE)
                    ILQR_LAUNCH((forward_kernel<T, Dyn, ILQR_INT_DISCRETE>), dim3((a.B + 63) / 64, a.n_pass), dim3(64), 0, s, a);
                else
                    ILQR_LAUNCH((forward_kernel<T, Dyn, ILQR_INT_EULER>), dim3((a.B + 63) / 64, a.n_pass), dim3(64), 0, s, a);
                return;
            }
            if constexpr (NX == 16 && NU == 8) {
                // all candidates of a trajectory as the columns of one matrix recursion on the matrix cores
                // (forward_mfma16.hpp); 32-bit buffer offsets into X, U and the gains
                static const bool wave = getenv("ILQR_FORWARD_WAVE") != nullptr;   // A/B: wave per (trajectory, alpha)
                const size_t bytes_x = (size_t)a.n_slots * (a.N + 1) * NX * a.B * sizeof(T);
                const size_t bytes_g = (size_t)a.N * a.B * gain_record(NX, NU) * sizeof(T);
                if (!wave && a.n_pass <= 16 && std::max(bytes_x, bytes_g) <= kDescriptorMax) {
                    ILQR_LAUNCH((forward_mfma16_kernel<T>), dim3(a.B), dim3(64), 0, s, a);
                    return;
                }
            }
            ILQR_LAUNCH((forward_wave_kernel<T, NX, NU>), dim3(a.B, a.n_pass), dim3(64), 0, s, a);
        };
    }
    if constexpr (NX == 16 && NU == 8) {
        o.sweep_reads_sparse = [](T mu) {
            return mu == T(0) && getenv("ILQR_BACKWARD_WAVE_LDS") == nullptr && getenv("ILQR_MFMA16_GENERAL") == nullptr;
        };
    }
    o.backward = [](const KArgs<T>& a, hipStream_t s) {
        if constexpr (NX == 16 && NU == 8) {
            // the (16, 8) sweep runs on the matrix cores (backward_mfma16.hpp); mu > 0 keeps the LDS form
            static const bool lds_form = getenv("ILQR_BACKWARD_WAVE_LDS") != nullptr;   // A/B switch
            if (a.mu == T(0) && !lds_form) {
                static const bool general = getenv("ILQR_MFMA16_GENERAL") != nullptr;   // A/B switch
                if (a.const_lin && !general) ILQR_LAUNCH((backward_mfma16_kernel<T, true>), dim3(a.B), dim3(64), 0, s, a);
                else ILQR_LAUNCH((backward_mfma16_kernel<T, false>), dim3(a.B), dim3(64), 0, s, a);
                return;
            }
        }
        ILQR_LAUNCH((backward_wave_kernel<T, NX, NU>), dim3(a.B), dim3(64), 0, s, a);
    };
    o.eval = [](const EvalArgs<T>& a, hipStream_t s) {
        ILQR_LAUNCH((eval_points_kernel<T, Dyn>), dim3((a.npts + 63) / 64), dim3(64), 0, s, a);
    };
    o.mpc_advance = [](const MpcArgs<T>& a, hipStream_t s) {
        ILQR_LAUNCH((mpc_advance_kernel<T, Dyn>), dim3((a.B + 63) / 64), dim3(64, kMpcChunks), 0, s, a);
    };
    o.n_dev_params = ParamLayout<Dyn::NSYS, NX, NU>::TOTAL;
    o.n_sys_dev = Dyn::NSYS;
    return o;
}

template <typename T> bool find_ops(int system, int nx, int nu, Ops<T>* out);

// host: ABI parameter block (doubles) -> device parameter block (see dynamics.hpp)
inline std::vector<double> build_device_params(int system, int nx, int nu, const double* p) {
    std::vector<double> d;
    const double* q = p;
    if (system == ILQR_SYS_PENDULUM) {
        const double g = p[0], l = p[1], dd = p[2];
        d = {g / l, dd};
        q = p + 3;
    } else if (system == ILQR_SYS_UA_DOUBLE_PENDULUM || system == ILQR_SYS_DOUBLE_PENDULUM) {
        const double g = p[0], m1 = p[1], m2 = p[2], l1 = p[3], l2 = p[4], d1 = p[5], d2 = p[6], th1 = p[7],
                     th2 = p[8];
        d = {m2 * l1 * l2,
             (m1 * l1 * l1) / 4 + m2 * l1 * l1 + (m2 * l2 * l2) / 4 + th1 + th2,
             (m2 * l2 * l2) / 4 + th2,
             m2 * g * l2 / 2,
             (m2 + m1 / 2) * g * l1,
             d1,
             d2};
        q = p + 9;
    } else if (system == ILQR_SYS_CUSTOM) {
        q = p;  // user-defined dynamics carry their constants in the generated code
    } else {  // linear: A, B verbatim
        d.assign(p, p + nx * nx + nx * nu);
        q = p + nx * nx + nx * nu;
    }
    const double* xt = q;
    const double* Q = xt + nx;
    const double* R = Q + nx * nx;
    const double* Qf = R + nu * nu;
    d.insert(d.end(), xt, xt + nx);
    d.insert(d.end(), Q, Q + nx * nx);
    d.insert(d.end(), R, R + nu * nu);
    d.insert(d.end(), Qf, Qf + nx * nx);
    auto sym = [&](const double* A, int n) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) d.push_back(0.5 * (A[i * n + j] + A[j * n + i]));
    };
    sym(Q, nx);
    sym(R, nu);
    sym(Qf, nx);
    return d;
}

struct PhaseTimer {
    struct Rec { int phase; hipEvent_t a, b; };
    bool on = false;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double ms[ILQR_N_PHASES] = {0};
    int64_t n[ILQR_N_PHASES] = {0};
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
    void begin(int phase, hipStream_t) {
        if (!on) return;
        Rec r{phase, get_event(), get_event()};
        launch_events() = LaunchEvents{r.a, r.b};  // consumed by the next ILQR_LAUNCH
        pending.push_back(r);
    }
    void end(hipStream_t s) {
        if (!on) return;
        LaunchEvents& le = launch_events();
        if (le.a) {  // nothing was launched inside the bracket: fall back to plain records
            hipEventRecord(le.a, s);
            hipEventRecord(le.b, s);
            le = LaunchEvents();
        }
        if (pending.size() >= 8192) resolve(s);
    }
    void resolve(hipStream_t s) {
        if (pending.empty()) return;
        hipStreamSynchronize(s);
        for (auto& r : pending) {
            float t = 0.f;
            hipEventElapsedTime(&t, r.a, r.b);
            ms[r.phase] += t;
            n[r.phase] += 1;
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        pending.clear();
    }
    void reset(hipStream_t s) {
        resolve(s);
        for (int i = 0; i < ILQR_N_PHASES; ++i) { ms[i] = 0; n[i] = 0; }
    }
    ~PhaseTimer() {
        for (auto& r : pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
        for (auto e : pool) hipEventDestroy(e);
    }
};

// One set of device state: the solver proper, and a second, smaller one used by the
// pure functional calls (backward_pass / forward_pass) so they never disturb the solver.
template <typename T> struct DeviceState {
    int n_slots = 0;
    T *X = nullptr, *U = nullptr, *gains = nullptr, *lin = nullptr, *term = nullptr, *x0 = nullptr;
    T *costs = nullptr, *cost = nullptr, *cost_prev = nullptr, *alpha_taken = nullptr;
    int *cur_slot = nullptr, *status = nullptr, *iters = nullptr, *accepted = nullptr, *counters = nullptr;
    bool slots_stale = false;   // linearize has moved the active trajectories to slot 0, cur_slot not yet reset
    bool lin_const = false;     // `lin` holds the library's own linearisation of a system whose matrices are constant (KArgs::const_lin)
    bool lin_full = true;       // every record of `lin` holds its matrices (false after a sparse linearise: see KArgs::lin_sparse)
};

template <typename T> class SolverT : public SolverBase {
  public:
    int B, N, NX, NU, E, A, R;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Ops<T> ops;
    T* params = nullptr;
    DeviceState<T> st, fn;  // solver state, functional-call scratch state
    T* staging = nullptr;   // dense staging for layout conversion
    size_t staging_elems = 0;
    T* plant_x = nullptr;
    T* eval_buf = nullptr;      // scratch of eval_points, grown on demand (the host-loop MPC calls f_fcn once per step)
    size_t eval_cap = 0;
    T *mpc_u_log = nullptr, *mpc_x_log = nullptr, *mpc_cost_log = nullptr;
    int mpc_log_steps = 0;
    long long* probe = nullptr;  // device, 8 x int64 (see ClockProbe)
    bool probe_on = false;
    size_t probe_elems = 8;
    int* h_counter = nullptr;  // pinned
    std::vector<double> trial_alphas;
    PhaseTimer timer;
    bool have_problem = false, have_rollout = false, mpc_ready = false;
    int iter_seq = 0;

    ~SolverT() override {
        if (stream) hipStreamSynchronize(stream);
        free_state(st);
        free_state(fn);
        hipFree(params);
        hipFree(staging);
        hipFree(plant_x);
        hipFree(eval_buf);
        hipFree(probe);
        hipFree(mpc_u_log);
        hipFree(mpc_x_log);
        hipFree(mpc_cost_log);
        if (h_counter) hipHostFree(h_counter);
        if (iter_graph) hipGraphExecDestroy(iter_graph);
        if (loop_ev[0]) { hipEventDestroy(loop_ev[0]); hipEventDestroy(loop_ev[1]); }
        if (own_stream && stream) hipStreamDestroy(stream);
    }

    static void free_state(DeviceState<T>& s) {
        hipFree(s.X); hipFree(s.U); hipFree(s.gains); hipFree(s.lin); hipFree(s.term); hipFree(s.x0);
        hipFree(s.costs); hipFree(s.cost); hipFree(s.cost_prev); hipFree(s.alpha_taken);
        hipFree(s.cur_slot); hipFree(s.status); hipFree(s.iters); hipFree(s.accepted); hipFree(s.counters);
        s = DeviceState<T>();
    }

    int alloc_state(DeviceState<T>& s, int n_slots) {
        s.n_slots = n_slots;
        const size_t b = B;
        auto al = [&](auto** p, size_t n) -> hipError_t {
            hipError_t e = hipMalloc((void**)p, n * sizeof(**p));
            if (e == hipSuccess) e = hipMemsetAsync(*p, 0, n * sizeof(**p), stream);
            return e;
        };
        ILQR_HIPCHK(al(&s.X, (size_t)n_slots * (N + 1) * NX * b));
        ILQR_HIPCHK(al(&s.U, (size_t)n_slots * N * NU * b));
        ILQR_HIPCHK(al(&s.gains, (size_t)N * R * b));
        ILQR_HIPCHK(al(&s.lin, (size_t)N * ops.lin_stride * b));
        ILQR_HIPCHK(al(&s.term, (size_t)(ops.tile16 ? 20 : NX + NX * NX) * b));   // tile mode: padded to 4 + 4 x 4
        ILQR_HIPCHK(al(&s.x0, (size_t)NX * b));
        ILQR_HIPCHK(al(&s.costs, (size_t)kMaxAlpha * b));
        ILQR_HIPCHK(al(&s.cost, b));
        ILQR_HIPCHK(al(&s.cost_prev, b));
        ILQR_HIPCHK(al(&s.alpha_taken, b));
        ILQR_HIPCHK(al(&s.cur_slot, b));
        ILQR_HIPCHK(al(&s.status, b));
        ILQR_HIPCHK(al(&s.iters, b));
        ILQR_HIPCHK(al(&s.accepted, b));
        ILQR_HIPCHK(al(&s.counters, (size_t)kCounterRing));
        return ILQR_OK;
    }

    // preset: the kernel set of a user-defined system compiled into a plugin (csrc/plugin_template.hip.in);
    // nullptr: one of the built-in systems of this library
    int init(const ilqr_config& c, const Ops<T>* preset = nullptr) {
        cfg = c;
        B = c.batch; N = c.horizon; NX = c.n_x; NU = c.n_u; A = c.n_alpha;
        E = 2 * NX * NX + 2 * NX * NU + NX + NU + NU * NU;
        R = gain_record(NX, NU);
        if (preset) {
            ops = *preset;
        } else if (!find_ops<T>(c.system, NX, NU, &ops)) {
            err = "no kernels compiled for this (system, n_x, n_u, dtype)";
            return ILQR_ERR_UNSUPPORTED;
        }
        if (ops.tile_scalars == kTile16M2 && (size_t)N * B * kTile16M2 * sizeof(T) > kDescriptorMax) {
            err = "n_x = 4, n_u = 2: horizon * batch too large for the sweep's 32-bit tile offsets (< 2 GiB of tiles)";
            return ILQR_ERR_UNSUPPORTED;
        }
        ILQR_HIPCHK(hipSetDevice(c.device));
        if (c.stream) {
            stream = (hipStream_t)c.stream;
        } else {
            ILQR_HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            own_stream = true;
        }
        // backtracking schedule exactly as the Python loop builds it (iLQR_class.py:279-302)
        double al = 1.0;
        for (int j = 0; j < c.n_trials; ++j) {
            trial_alphas.push_back(al);
            al *= c.alpha_factor;
            if (al < c.min_alpha) break;
        }
        std::vector<double> dp = build_device_params(c.system, NX, NU, c.params);
        if ((int)dp.size() != ops.n_dev_params) { err = "internal: device parameter block size mismatch"; return ILQR_ERR_INVALID_ARG; }
        std::vector<T> dpt(dp.begin(), dp.end());
        ILQR_HIPCHK(hipMalloc((void**)&params, dpt.size() * sizeof(T)));
        ILQR_HIPCHK(hipMemcpy(params, dpt.data(), dpt.size() * sizeof(T), hipMemcpyHostToDevice));
        int rc = alloc_state(st, A + 1);
        if (rc) return rc;
        // every layout conversion goes through `staging`: trajectories, gains, the expansion, and the small per-trajectory
        // blocks (padded terminal expansion of the tile sweeps: 20; trial costs: up to kMaxAlpha)
        staging_elems = (size_t)B * std::max({(size_t)(N + 1) * NX, (size_t)N * NU * NX, (size_t)N * E, (size_t)20,
                                              (size_t)kMaxAlpha, (size_t)(NX + NX * NX)});
        ILQR_HIPCHK(hipMalloc((void**)&staging, staging_elems * sizeof(T)));
        ILQR_HIPCHK(hipMalloc((void**)&plant_x, (size_t)NX * B * sizeof(T)));
        ILQR_HIPCHK(hipMemsetAsync(plant_x, 0, (size_t)NX * B * sizeof(T), stream));
        probe_on = getenv("ILQR_CLOCK_PROBE") != nullptr;
        probe_elems = probe_on ? (8 + 2 * 65536 * 4) : 8;
        ILQR_HIPCHK(hipMalloc((void**)&probe, probe_elems * sizeof(long long)));
        ILQR_HIPCHK(hipMemsetAsync(probe, 0, probe_elems * sizeof(long long), stream));
        ILQR_HIPCHK(hipHostMalloc((void**)&h_counter, kCounterRing * sizeof(int)));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return ILQR_OK;
    }

    KArgs<T> kargs(const DeviceState<T>& s) const {
        KArgs<T> a{};
        a.B = B; a.N = N; a.n_slots = s.n_slots; a.integ = cfg.integrator; a.maxiter = cfg.maxiter; a.flags = cfg.flags;
        a.dt = (T)cfg.dt; a.tol = (T)cfg.tol; a.mu = (T)cfg.mu;
        a.X = s.X; a.U = s.U; a.cur_slot = s.cur_slot; a.gains = s.gains; a.lin = s.lin; a.term = s.term;
        a.x0 = s.x0; a.costs = s.costs; a.cost = s.cost; a.cost_prev = s.cost_prev; a.alpha_taken = s.alpha_taken;
        a.status = s.status; a.iters = s.iters; a.accepted = s.accepted; a.counters = s.counters; a.params = params;
        a.reset_slots = 0;
        a.probe = probe_on ? probe : nullptr;
        return a;
    }

    int check_launch() {
        ILQR_HIPCHK(hipGetLastError());
        return ILQR_OK;
    }

    int sync() override {
        if (int rf = flush_select()) return rf;
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return ILQR_OK;
    }

    // ---- layout conversion helpers (dense host layout <-> device) ------------------
    unsigned grid_for(size_t n) const { return (unsigned)((n + 255) / 256); }
    int staging_check(size_t n) {
        if (n <= staging_elems) return ILQR_OK;
        err = "internal: layout-conversion staging buffer too small for this call";
        return ILQR_ERR_INVALID_ARG;
    }

    int up_ct(const void* host, T* slots, const int* cur_slot, int C, int Tn) {
        const size_t n = (size_t)B * C * Tn;
        if (int rs_ = staging_check(n)) return rs_;
        ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(scatter_ct_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, slots, cur_slot, B, C, Tn);
        ILQR_HIPCHK(hipStreamSynchronize(stream));  // the caller's host buffer may be released after return
        return check_launch();
    }
    int down_ct(void* host, const T* slots, const int* cur_slot, int C, int Tn) {
        const size_t n = (size_t)B * C * Tn;
        if (int rs_ = staging_check(n)) return rs_;
        hipLaunchKernelGGL(gather_ct_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, slots, cur_slot, B, C, Tn);
        ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int up_tc(const void* host, T* dev, int C, int Tn) {
        const size_t n = (size_t)B * C * Tn;
        if (int rs_ = staging_check(n)) return rs_;
        ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(scatter_tc_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, dev, B, C, Tn);
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int down_tc(void* host, const T* dev, int C, int Tn) {
        const size_t n = (size_t)B * C * Tn;
        if (int rs_ = staging_check(n)) return rs_;
        hipLaunchKernelGGL(gather_tc_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, dev, B, C, Tn);
        ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }

    int up_gain_K(const void* host, T* gains) {
        const size_t n = (size_t)B * N * NU * NX;
        if (int rs_ = staging_check(n)) return rs_;
        ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gains_scatter_K_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, gains, B, N, NU * NX, R);
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int down_gain_K(void* host, const T* gains) {
        const size_t n = (size_t)B * N * NU * NX;
        if (int rs_ = staging_check(n)) return rs_;
        hipLaunchKernelGGL(gains_gather_K_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, gains, B, N, NU * NX, R);
        ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int up_gain_k(const void* host, T* gains) {
        const size_t n = (size_t)B * N * NU;
        if (int rs_ = staging_check(n)) return rs_;
        ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gains_scatter_k_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, gains, B, N, NU, NU * NX, R);
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int down_gain_k(void* host, const T* gains) {
        const size_t n = (size_t)B * N * NU;
        if (int rs_ = staging_check(n)) return rs_;
        hipLaunchKernelGGL(gains_gather_k_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, gains, B, N, NU, NU * NX, R);
        ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }
    int down_lin(void* host, const T* lin) {
        if (ops.lin_aos) {
            const size_t n = (size_t)B * N * E;
            if (int rs_ = staging_check(n)) return rs_;
            hipLaunchKernelGGL(gains_gather_K_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N, E, E);
            ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
            ILQR_HIPCHK(hipStreamSynchronize(stream));
            return check_launch();
        }
        if (!ops.tile16) return down_tc(host, lin, E, N);
        const size_t n = (size_t)B * N * E;
        if (int rs_ = staging_check(n)) return rs_;
        if (ops.tile_scalars == kTile16M2)
            hipLaunchKernelGGL(tile16m2_gather_dense_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N);
        else
            hipLaunchKernelGGL(tile16_gather_dense_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N, NX);
        ILQR_HIPCHK(hipMemcpyAsync(host, staging, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }

    // dense [B][N][E] expansion records -> the layout the backward kernel of this (n_x, n_u) reads
    int up_lin(const void* host, T* lin) {
        const size_t n = (size_t)B * N * E;
        if (int rs_ = staging_check(n)) return rs_;
        if (ops.lin_aos) {
            ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(gains_scatter_K_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N, E, E);
            ILQR_HIPCHK(hipStreamSynchronize(stream));
            return check_launch();
        }
        if (!ops.tile16) return up_tc(host, lin, E, N);
        ILQR_HIPCHK(hipMemcpyAsync(staging, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
        ILQR_HIPCHK(hipMemsetAsync(lin, 0, (size_t)N * B * ops.tile_scalars * sizeof(T), stream));
        if (ops.tile_scalars == kTile16M2)
            hipLaunchKernelGGL(tile16m2_scatter_dense_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N);
        else
            hipLaunchKernelGGL(tile16_scatter_dense_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, staging, lin, B, N, NX);
        ILQR_HIPCHK(hipStreamSynchronize(stream));
        return check_launch();
    }

    int zero_solver_state(DeviceState<T>& s) {
        const size_t b = B;
        ILQR_HIPCHK(hipMemsetAsync(s.X, 0, (size_t)s.n_slots * (N + 1) * NX * b * sizeof(T), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.U, 0, (size_t)s.n_slots * N * NU * b * sizeof(T), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.gains, 0, (size_t)N * R * b * sizeof(T), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.cur_slot, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.status, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.iters, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.accepted, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.cost, 0, b * sizeof(T), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.cost_prev, 0, b * sizeof(T), stream));
        ILQR_HIPCHK(hipMemsetAsync(s.alpha_taken, 0, b * sizeof(T), stream));
        return ILQR_OK;
    }

    // fresh solver as after iLQR.__init__ (iLQR_class.py:55-61)
    int set_problem(const void* x0, const void* U) override {
        if (!x0 || !U) { err = "set_problem: NULL pointer"; return ILQR_ERR_INVALID_ARG; }
        sel_pending = false;     // the state it would have updated is wiped
        int rc = zero_solver_state(st);
        if (rc) return rc;
        if ((rc = up_tc(x0, st.x0, NX, 1))) return rc;
        if ((rc = up_ct(U, st.U, st.cur_slot, NU, N))) return rc;
        have_problem = true;
        have_rollout = false;
        return ILQR_OK;
    }

    size_t field_bytes(int field) const {
        const size_t b = B;
        switch (field) {
            case ILQR_X: return b * NX * (N + 1) * sizeof(T);
            case ILQR_U: case ILQR_UFF: return b * NU * N * sizeof(T);
            case ILQR_K: return b * N * NU * NX * sizeof(T);
            case ILQR_X0: case ILQR_PLANT_X: return b * NX * sizeof(T);
            case ILQR_COST: case ILQR_ALPHA: return b * sizeof(T);
            case ILQR_STATUS: case ILQR_ITERS: return b * sizeof(int32_t);
            case ILQR_PROBE: return 8 * sizeof(long long);
            case ILQR_TRIAL_COSTS: return b * A * sizeof(T);
            case ILQR_LIN: return b * N * E * sizeof(T);
            default: return 0;
        }
    }

    int set(int field, const void* src, size_t bytes) override {
        if (!src) { err = "set: NULL pointer"; return ILQR_ERR_INVALID_ARG; }
        const size_t want = field_bytes(field);
        if (want == 0 || bytes != want) { err = "set: unknown field or wrong byte count"; return ILQR_ERR_INVALID_ARG; }
        if (int rf = flush_select()) return rf;
        if (int rcs = fix_slots(st)) return rcs;
        switch (field) {
            case ILQR_X: return up_ct(src, st.X, st.cur_slot, NX, N + 1);
            case ILQR_U: return up_ct(src, st.U, st.cur_slot, NU, N);
            case ILQR_UFF: return up_gain_k(src, st.gains);
            case ILQR_K: return up_gain_K(src, st.gains);
            case ILQR_X0: return up_tc(src, st.x0, NX, 1);
            case ILQR_PLANT_X: return up_tc(src, plant_x, NX, 1);
            default: err = "set: field is read-only"; return ILQR_ERR_INVALID_ARG;
        }
    }

    int get(int field, void* dst, size_t bytes) override {
        if (!dst) { err = "get: NULL pointer"; return ILQR_ERR_INVALID_ARG; }
        const size_t want = field_bytes(field);
        if (want == 0 || bytes != want) { err = "get: unknown field or wrong byte count"; return ILQR_ERR_INVALID_ARG; }
        if (int rf = flush_select()) return rf;
        if (int rcs = fix_slots(st)) return rcs;
        switch (field) {
            case ILQR_X: return down_ct(dst, st.X, st.cur_slot, NX, N + 1);
            case ILQR_U: return down_ct(dst, st.U, st.cur_slot, NU, N);
            case ILQR_UFF: return down_gain_k(dst, st.gains);
            case ILQR_K: return down_gain_K(dst, st.gains);
            case ILQR_X0: return down_tc(dst, st.x0, NX, 1);
            case ILQR_PLANT_X: return down_tc(dst, plant_x, NX, 1);
            case ILQR_LIN:
                if (!st.lin_full || lin_stale) {   // the hot path wrote gradients only, or nothing (fused): bring the records up to date first
                    if (int rl = do_linearize(st, true)) return rl;
                }
                return down_lin(dst, st.lin);
            case ILQR_TRIAL_COSTS: return down_tc(dst, st.costs, A, 1);
            case ILQR_COST: ILQR_HIPCHK(hipMemcpyAsync(dst, st.cost, bytes, hipMemcpyDeviceToHost, stream)); return sync();
            case ILQR_ALPHA: ILQR_HIPCHK(hipMemcpyAsync(dst, st.alpha_taken, bytes, hipMemcpyDeviceToHost, stream)); return sync();
            case ILQR_STATUS: ILQR_HIPCHK(hipMemcpyAsync(dst, st.status, bytes, hipMemcpyDeviceToHost, stream)); return sync();
            case ILQR_ITERS: ILQR_HIPCHK(hipMemcpyAsync(dst, st.iters, bytes, hipMemcpyDeviceToHost, stream)); return sync();
            case ILQR_PROBE: ILQR_HIPCHK(hipMemcpyAsync(dst, probe, bytes, hipMemcpyDeviceToHost, stream)); return sync();
            default: err = "get: unknown field"; return ILQR_ERR_INVALID_ARG;
        }
    }

    // ---- stages ---------------------------------------------------------------------
    // full = false: the sweep that follows may be the constant-matrix form, which reads the matrices at t = N-1 only
    int do_linearize(DeviceState<T>& s, bool full = false) {
        KArgs<T> a = kargs(s);
        const bool sparse = !full && ops.const_lin && ops.sweep_reads_sparse && ops.sweep_reads_sparse((T)cfg.mu);
        a.lin_sparse = sparse ? 1 : 0;
        s.lin_full = !sparse;
        timer.begin(ILQR_PHASE_LINEARIZE, stream);
        ops.linearize[cfg.integrator](a, stream);
        timer.end(stream);
        s.slots_stale = ops.canonical;   // the sweep that follows resets cur_slot (KArgs::reset_slots)
        s.lin_const = ops.const_lin;
        if (&s == &st) lin_stale = false;
        return check_launch();
    }
    // cur_slot must be truthful before anything but the backward sweep looks at it
    int fix_slots(DeviceState<T>& s) {
        if (!s.slots_stale) return ILQR_OK;
        KArgs<T> a = kargs(s);
        hipLaunchKernelGGL(reset_slots_kernel<T>, dim3(grid_for((size_t)B)), dim3(256), 0, stream, a);
        s.slots_stale = false;
        return check_launch();
    }
    int do_backward(DeviceState<T>& s) {
        KArgs<T> a = kargs(s);
        a.reset_slots = s.slots_stale ? 1 : 0;
        a.const_lin = s.lin_const ? 1 : 0;
        a.lin_sparse = (s.lin_const && !s.lin_full) ? 1 : 0;    // where the CONST sweep finds l_x, l_u (KArgs::lin_sparse)
        timer.begin(ILQR_PHASE_BACKWARD, stream);
        ops.backward(a, stream);
        timer.end(stream);
        s.slots_stale = false;
        return check_launch();
    }
    int do_forward(DeviceState<T>& s, const double* alphas, int n, bool init = false) {
        if (n < 1 || n > s.n_slots - 1 || n > kMaxAlpha) { err = "forward: alpha count out of range"; return ILQR_ERR_INVALID_ARG; }
        int rcf = fix_slots(s);
        if (rcf) return rcf;
        KArgs<T> a = kargs(s);
        a.n_pass = n;
        a.init_mode = init;     // head of a solve: every trajectory rolls out, counter slot 0 is cleared
        a.counter_idx = 0;
        for (int i = 0; i < n; ++i) a.alphas[i] = (T)alphas[i];
        timer.begin(ILQR_PHASE_FORWARD, stream);
        ops.forward[cfg.integrator](a, stream);
        timer.end(stream);
        return check_launch();
    }
    int do_select(DeviceState<T>& s, const double* alphas, int n, bool last, bool init, int counter_idx) {
        if (int rcs = fix_slots(s)) return rcs;
        KArgs<T> a = kargs(s);
        a.n_pass = n; a.last_pass = last; a.init_mode = init; a.counter_idx = counter_idx;
        for (int i = 0; i < n; ++i) a.alphas[i] = (T)alphas[i];
        timer.begin(ILQR_PHASE_SELECT, stream);
        ILQR_LAUNCH(select_kernel<T>, dim3((B + 255) / 256), dim3(256), 0, stream, a);
        timer.end(stream);
        return check_launch();
    }

    int pending_n = 0;
    double pending_alphas[kMaxAlpha];

    // ---- the fused iteration (backward_fused16.hpp) -----------------------------------------------------------------
    // An iteration is then TWO launches: [acceptance step of the previous candidates + linearise + sweep] and the
    // rollouts.  The acceptance step of the newest candidates stays pending until the next fused launch runs it for its
    // own trajectories, or until anything else looks at the solver state (flush_select: the stand-alone kernel).
    bool sel_pending = false;
    int sel_cidx = 0, sel_n = 0;
    double sel_alphas[kMaxAlpha];
    bool lin_stale = false;      // the expansion in HBM is not the current trajectory's (the fused kernel never writes it)

    bool fused_ok() const {
        static const bool off = getenv("ILQR_NO_FUSE") != nullptr;   // A/B switch, and bench.py's materialised leg
        return !off && !force_unfused && !(cfg.flags & ILQR_FLAG_NO_FUSE) && ops.fused[cfg.integrator] && cfg.mu == 0.0 && (int)trial_alphas.size() <= A &&
               (size_t)N * B * R * sizeof(T) <= kDescriptorMax;
    }
    bool force_unfused = false;
    // the persistent form (persistent.hpp): same conditions as the fused kernel, plus the ring rollout's 32-bit offsets
    bool persist_ok() const {
        static const bool off = getenv("ILQR_NO_PERSIST") != nullptr;   // A/B switch
        const size_t bytes_x = (size_t)st.n_slots * (N + 1) * NX * B * sizeof(T);
        return !off && !(cfg.flags & ILQR_FLAG_NO_PERSIST) && fused_ok() && ops.persist[cfg.integrator] && bytes_x <= kDescriptorMax &&
               (B <= persist_small_max() || ops.persist_any_batch[cfg.integrator]);
    }
    int launch_persist(int n_iters, bool do_init, int n_mpc, const MpcArgs<T>* mpc) {
        int rc;
        if ((rc = flush_select())) return rc;
        if ((rc = fix_slots(st))) return rc;
        KArgs<T> a = kargs(st);
        const int n = (int)trial_alphas.size();
        a.n_pass = n; a.last_pass = 1; a.counter_idx = 0;
        for (int i = 0; i < n; ++i) a.alphas[i] = (T)trial_alphas[i];
        PArgs<T> pa{};
        pa.n_iters = n_iters; pa.do_init = do_init ? 1 : 0; pa.n_mpc = n_mpc;
        if (mpc) pa.mpc = *mpc;
        timer.begin(ILQR_PHASE_PERSIST, stream);
        ops.persist[cfg.integrator](a, pa, stream);
        timer.end(stream);
        lin_stale = true;
        return check_launch();
    }
    int flush_select() {
        if (!sel_pending) return ILQR_OK;
        sel_pending = false;
        return do_select(st, sel_alphas, sel_n, true, false, sel_cidx);
    }
    int flush() override { return flush_select(); }

    int initial_rollout() override {
        if (!have_problem) { err = "initial_rollout before set_problem"; return ILQR_ERR_STATE; }
        int rc;
        if ((rc = flush_select())) return rc;
        // All trajectories take part in the head of a solve, whatever their previous status: the rollout's init mode
        // ignores status / accepted and clears counter slot 0, the select's init mode rewrites status, iteration count
        // and accepted flag of every trajectory -- two launches, no memsets (an MPC step used to pay three).
        const double zero = 0.0;
        if ((rc = do_forward(st, &zero, 1, true))) return rc;
        if ((rc = do_select(st, &zero, 1, false, true, 0))) return rc;
        have_rollout = true;
        iter_seq = 0;
        return ILQR_OK;
    }
    int linearize() override {
        if (!have_problem) { err = "linearize before set_problem"; return ILQR_ERR_STATE; }
        if (int rf = flush_select()) return rf;
        return do_linearize(st);
    }
    int backward() override {
        if (!have_problem) { err = "backward before set_problem"; return ILQR_ERR_STATE; }
        if (int rf = flush_select()) return rf;
        if (lin_stale) {      // the last iteration ran fused: there is no expansion in HBM to sweep over yet
            if (int rl = do_linearize(st)) return rl;
        }
        return do_backward(st);
    }
    int forward(const double* alphas, int n) override {
        if (!have_rollout) { err = "forward before initial_rollout"; return ILQR_ERR_STATE; }
        if (!alphas) { err = "forward: NULL alphas"; return ILQR_ERR_INVALID_ARG; }
        if (n < 1 || n > A) { err = "forward: alpha count must be in [1, n_alpha]"; return ILQR_ERR_INVALID_ARG; }
        int rc = flush_select();
        if (rc) return rc;
        rc = do_forward(st, alphas, n);
        if (rc) return rc;
        pending_n = n;
        for (int i = 0; i < n; ++i) pending_alphas[i] = alphas[i];
        return ILQR_OK;
    }
    int select() override {
        if (pending_n == 0) { err = "select without a preceding forward"; return ILQR_ERR_STATE; }
        int rc = do_select(st, pending_alphas, pending_n, true, false, next_counter());
        pending_n = 0;
        return rc;
    }

    int next_counter() {
        iter_seq += 1;
        return iter_seq % kCounterRing;
    }

    // one iLQR iteration for the whole batch; returns the ring index that will hold the
    // number of trajectories still active after it
    int one_iteration(int* counter_idx, int forced_cidx = -1) {
        int rc;
        if (fused_ok() && forced_cidx < 0) {
            if ((rc = fix_slots(st))) return rc;
            KArgs<T> a = kargs(st);
            a.fuse_select = sel_pending ? 1 : 0;
            a.n_pass = sel_n; a.last_pass = 1; a.counter_idx = sel_cidx;
            for (int i = 0; i < sel_n; ++i) a.alphas[i] = (T)sel_alphas[i];
            timer.begin(ILQR_PHASE_FUSED, stream);
            ops.fused[cfg.integrator](a, stream);
            timer.end(stream);
            if ((rc = check_launch())) return rc;
            sel_pending = false;
            lin_stale = true;
            const int n = (int)trial_alphas.size();
            const int cidx = next_counter();
            if ((rc = do_forward(st, trial_alphas.data(), n))) return rc;
            sel_pending = true;
            sel_cidx = cidx;
            sel_n = n;
            for (int i = 0; i < n; ++i) sel_alphas[i] = trial_alphas[i];
            if (counter_idx) *counter_idx = cidx;
            return ILQR_OK;
        }
        if ((rc = flush_select())) return rc;
        if ((rc = do_linearize(st))) return rc;
        if ((rc = do_backward(st))) return rc;
        const int total = (int)trial_alphas.size();
        const int cidx = forced_cidx >= 0 ? forced_cidx : next_counter();  // cleared by the previous select launch
        for (int base = 0; base < total; base += A) {
            const int n = std::min(A, total - base);
            const bool last = (base + n >= total);
            if ((rc = do_forward(st, trial_alphas.data() + base, n))) return rc;
            if ((rc = do_select(st, trial_alphas.data() + base, n, last, false, cidx))) return rc;
        }
        if (counter_idx) *counter_idx = cidx;
        return ILQR_OK;
    }

    // Optional (ILQR_USE_GRAPH=1): ilqr_iterate replays ONE captured iteration (linearize, sweep, rollout, select:
    // 4 dispatches) as a hipGraph.  Measured on this ROCm: 0.236-0.238 ms per step against 0.229-0.233 for plain
    // stream launches -- the ~12 us of gaps per iteration are GPU-side dependency latency between the kernels, which
    // a graph does not remove, and a graph launch costs more than four kernel launches -- so it is off by default.
    // The captured select uses a fixed slot of the active-count ring; only run_solve_loop reads that ring (never
    // through the graph) and it starts with initial_rollout, which clears it.
    hipGraphExec_t iter_graph = nullptr;
    bool graph_ok = true;
    int build_iter_graph() {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return 1; }
        const int rc = one_iteration(nullptr, kCounterRing - 1);
        const hipError_t e = hipStreamEndCapture(stream, &g);
        if (rc || e != hipSuccess || !g) { if (g) hipGraphDestroy(g); (void)hipGetLastError(); return 1; }
        const hipError_t ei = hipGraphInstantiate(&iter_graph, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (ei != hipSuccess) { iter_graph = nullptr; (void)hipGetLastError(); return 1; }
        return 0;
    }

    int iterate(int n) override {
        if (!have_rollout) { err = "iterate before initial_rollout"; return ILQR_ERR_STATE; }
        static const bool want_graph = getenv("ILQR_USE_GRAPH") != nullptr;
        int i = 0;
        if (want_graph && graph_ok && !timer.on && n >= 1) {
            force_unfused = true;      // the captured iteration is the four-launch form
            if (!iter_graph) {
                // one ordinary iteration first: one-time function attributes must not be set inside a capture
                int rc = one_iteration(nullptr);
                if (rc) return rc;
                i = 1;
                if (build_iter_graph()) graph_ok = false;
            }
            if (iter_graph) {
                for (; i < n; ++i) ILQR_HIPCHK(hipGraphLaunch(iter_graph, stream));
                return ILQR_OK;
            }
        }
        // (measured, fp32 c3 system, us per iteration persistent / two launches: B = 256 120 / 122, B = 1024 125 / 127,
        // B = 4096 162 / 150 -- in the 16-trajectory form only 3 of the workgroup's 8 waves roll out and the phases of a
        // workgroup wait for their slowest wave: the big-batch iteration keeps its two launches)
        static const int it_max = getenv("ILQR_PERSIST_ITERATE_MAX") ? atoi(getenv("ILQR_PERSIST_ITERATE_MAX")) : persist_small_max();
        if (i < n && B <= it_max && persist_ok()) return launch_persist(n - i, false, 0, nullptr);
        for (; i < n; ++i) {
            int rc = one_iteration(nullptr);
            if (rc) return rc;
        }
        return ILQR_OK;
    }

    // optimize_trajectory (iLQR_class.py:250-313) for the batch.  The per-trajectory loop
    // state lives on the device; the host only learns how many trajectories are still active,
    // one iteration late (so the stream never drains), and stops launching when none is.
    int run_solve_loop() {
        int rc;
        static const int solve_max = getenv("ILQR_PERSIST_SOLVE_MAX") ? atoi(getenv("ILQR_PERSIST_SOLVE_MAX")) : persist_small_max();
        if (B <= solve_max && persist_ok()) {
            // one launch: every workgroup runs the head of the solve and then iterates until its own trajectories are
            // done (or maxiter): no read-back of the active count, no surplus iterations
            if (!have_problem) { err = "solve before set_problem"; return ILQR_ERR_STATE; }
            if ((rc = launch_persist(cfg.maxiter, true, 0, nullptr))) return rc;
            have_rollout = true;
            iter_seq = 0;
            return ILQR_OK;
        }
        if ((rc = initial_rollout())) return rc;
        if (!loop_ev[0]) {
            ILQR_HIPCHK(hipEventCreateWithFlags(&loop_ev[0], hipEventDisableTiming));
            ILQR_HIPCHK(hipEventCreateWithFlags(&loop_ev[1], hipEventDisableTiming));
        }
        // Inactive trajectories are skipped inside every kernel, so the only reason to look at the count of active
        // ones is to stop launching once nobody is left.  The host reads it ONE ITERATION LATE: iteration i + 1 is
        // already queued when it waits for the count of iteration i, so the stream never drains for the read-back
        // (the price is one surplus iteration of early-exiting kernels at the end of a solve).
        // Short loops (the pendulum MPC of run_iLQR_MPC.py runs maxiter = 10) are simply enqueued whole: a finished
        // trajectory is skipped inside every kernel, so surplus iterations cost a few microseconds of early-exiting
        // launches each, less than one host round trip.
        static const int enqueue_all = getenv("ILQR_SOLVE_ENQUEUE_ALL") ? atoi(getenv("ILQR_SOLVE_ENQUEUE_ALL")) : 12;
        if (cfg.maxiter <= enqueue_all) {
            for (int i = 0; i < cfg.maxiter; ++i)
                if ((rc = one_iteration(nullptr))) return rc;
            return flush_select();
        }
        if (fused_ok()) {
            // The count of trajectories still active after iteration i is written by the kernel that runs its
            // acceptance step: the fused launch of iteration i + 1.  It is copied out right behind that launch's
            // iteration and looked at one iteration later still, so the stream never drains for the read-back (the
            // price is two surplus iterations of early-exiting kernels at the end of a solve instead of one).
            int cidx_of[3] = {-1, -1, -1};     // counter slot of iterations i, i-1, i-2
            for (int i = 0; i < cfg.maxiter; ++i) {
                cidx_of[2] = cidx_of[1]; cidx_of[1] = cidx_of[0];
                if ((rc = one_iteration(&cidx_of[0]))) return rc;
                if (i >= 1) {
                    ILQR_HIPCHK(hipMemcpyAsync(h_counter + cidx_of[1], st.counters + cidx_of[1], sizeof(int), hipMemcpyDeviceToHost, stream));
                    ILQR_HIPCHK(hipEventRecord(loop_ev[i & 1], stream));
                }
                if (i >= 2) {
                    ILQR_HIPCHK(hipEventSynchronize(loop_ev[(i - 1) & 1]));
                    if (h_counter[cidx_of[2]] == 0) { sel_pending = false; break; }   // nobody is active: nothing left to accept
                }
            }
            return flush_select();
        }
        int prev = -1;
        for (int i = 0; i < cfg.maxiter; ++i) {
            int cidx;
            if ((rc = one_iteration(&cidx))) return rc;
            ILQR_HIPCHK(hipMemcpyAsync(h_counter + cidx, st.counters + cidx, sizeof(int), hipMemcpyDeviceToHost, stream));
            ILQR_HIPCHK(hipEventRecord(loop_ev[i & 1], stream));
            static const bool eager = getenv("ILQR_SOLVE_SYNC_EVERY_ITERATION") != nullptr;   // A/B switch
            if (eager) {
                ILQR_HIPCHK(hipEventSynchronize(loop_ev[i & 1]));
                if (h_counter[cidx] == 0) break;
            } else if (prev >= 0) {
                ILQR_HIPCHK(hipEventSynchronize(loop_ev[(i - 1) & 1]));
                if (h_counter[prev] == 0) break;
            }
            prev = cidx;
        }
        return ILQR_OK;
    }
    hipEvent_t loop_ev[2] = {nullptr, nullptr};

    int solve(int32_t* iters, void* cost) override {
        if (!have_problem) { err = "solve before set_problem"; return ILQR_ERR_STATE; }
        int rc = run_solve_loop();
        if (rc) return rc;
        if (iters) ILQR_HIPCHK(hipMemcpyAsync(iters, st.iters, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
        if (cost) ILQR_HIPCHK(hipMemcpyAsync(cost, st.cost, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, stream));
        return sync();
    }

    // ---- pure functional calls --------------------------------------------------------
    int ensure_fn() {
        if (fn.X) return ILQR_OK;
        int rc = alloc_state(fn, 2);
        if (rc) return rc;
        return ILQR_OK;
    }
    int reset_fn() {
        const size_t b = B;
        ILQR_HIPCHK(hipMemsetAsync(fn.cur_slot, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(fn.status, 0, b * sizeof(int), stream));
        ILQR_HIPCHK(hipMemsetAsync(fn.accepted, 0, b * sizeof(int), stream));
        return ILQR_OK;
    }

    int backward_pass(const void* X, const void* U, void* Uff, void* K) override {
        if (!X || !U) { err = "backward_pass: NULL input"; return ILQR_ERR_INVALID_ARG; }
        int rc;
        if ((rc = ensure_fn()) || (rc = reset_fn())) return rc;
        if ((rc = up_ct(X, fn.X, nullptr, NX, N + 1))) return rc;
        if ((rc = up_ct(U, fn.U, nullptr, NU, N))) return rc;
        if ((rc = do_linearize(fn))) return rc;
        if ((rc = do_backward(fn))) return rc;
        if (Uff && (rc = down_gain_k(Uff, fn.gains))) return rc;
        if (K && (rc = down_gain_K(K, fn.gains))) return rc;
        return sync();
    }

    // the Riccati sweep alone, on an expansion the caller computed (its own autodiff, identified model, ...)
    int backward_tensors(const void* lin, const void* term, void* Uff, void* K) override {
        if (!lin || !term) { err = "backward_tensors: NULL input"; return ILQR_ERR_INVALID_ARG; }
        int rc;
        if ((rc = ensure_fn()) || (rc = reset_fn())) return rc;
        if ((rc = up_lin(lin, fn.lin))) return rc;
        fn.lin_const = false;    // caller-supplied tensors: anything goes
        fn.lin_full = true;
        if (ops.tile16 && NX < 4) {
            // the sweep reads [V_x (4) | V_xx (4 x 4)] zero-padded
            std::vector<T> pad((size_t)B * 20, T(0));
            const T* src = (const T*)term;
            for (int bq = 0; bq < B; ++bq) {
                for (int i = 0; i < NX; ++i) pad[(size_t)bq * 20 + i] = src[(size_t)bq * (NX + NX * NX) + i];
                for (int i = 0; i < NX; ++i)
                    for (int j = 0; j < NX; ++j) pad[(size_t)bq * 20 + 4 + 4 * i + j] = src[(size_t)bq * (NX + NX * NX) + NX + i * NX + j];
            }
            if ((rc = up_tc(pad.data(), fn.term, 20, 1))) return rc;
        } else if ((rc = up_tc(term, fn.term, NX + NX * NX, 1))) {
            return rc;
        }
        if ((rc = do_backward(fn))) return rc;
        if (Uff && (rc = down_gain_k(Uff, fn.gains))) return rc;
        if (K && (rc = down_gain_K(K, fn.gains))) return rc;
        return sync();
    }

    int forward_pass(const void* x0, double alpha, const void* X, const void* U, const void* Uff, const void* K,
                     void* Xn, void* Un, void* cost) override {
        if (!x0 || !X || !U || !Uff || !K) { err = "forward_pass: NULL input"; return ILQR_ERR_INVALID_ARG; }
        int rc;
        if ((rc = ensure_fn()) || (rc = reset_fn())) return rc;
        if ((rc = up_tc(x0, fn.x0, NX, 1))) return rc;
        if ((rc = up_ct(X, fn.X, nullptr, NX, N + 1))) return rc;
        if ((rc = up_ct(U, fn.U, nullptr, NU, N))) return rc;
        if ((rc = up_gain_k(Uff, fn.gains))) return rc;
        if ((rc = up_gain_K(K, fn.gains))) return rc;
        if ((rc = do_forward(fn, &alpha, 1))) return rc;
        // the candidate went to slot 1: read it back from there
        T* X1 = fn.X + (size_t)(N + 1) * NX * B;
        T* U1 = fn.U + (size_t)N * NU * B;
        if (Xn && (rc = down_ct(Xn, X1, nullptr, NX, N + 1))) return rc;
        if (Un && (rc = down_ct(Un, U1, nullptr, NU, N))) return rc;
        if (cost) ILQR_HIPCHK(hipMemcpyAsync(cost, fn.costs, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, stream));
        return sync();
    }

    int eval_points(int integ, int npts, const void* x, const void* u, void** outs) override {
        if (npts < 1 || !x) { err = "eval_points: bad arguments"; return ILQR_ERR_INVALID_ARG; }
        const size_t sizes[12] = {(size_t)NX, (size_t)NX * NX, (size_t)NX * NU, 1, (size_t)NX, (size_t)NU,
                                  (size_t)NX * NX, (size_t)NU * NX, (size_t)NU * NU, 1, (size_t)NX, (size_t)NX * NX};
        size_t total = (size_t)NX + NU;
        for (int i = 0; i < 12; ++i) total += sizes[i];
        if (total * npts > eval_cap) {     // grown on demand, kept: no hipMalloc / hipFree per call
            ILQR_HIPCHK(hipStreamSynchronize(stream));
            hipFree(eval_buf);
            eval_buf = nullptr;
            eval_cap = 0;
            ILQR_HIPCHK(hipMalloc((void**)&eval_buf, total * npts * sizeof(T)));
            eval_cap = total * npts;
        }
        T* buf = eval_buf;
        T* dx = buf;
        T* du = dx + (size_t)npts * NX;
        T* cur = du + (size_t)npts * NU;
        T* dout[12];
        for (int i = 0; i < 12; ++i) {
            dout[i] = outs[i] ? cur : nullptr;
            cur += sizes[i] * npts;
        }
        hipError_t e = hipMemcpyAsync(dx, x, (size_t)npts * NX * sizeof(T), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && u) e = hipMemcpyAsync(du, u, (size_t)npts * NU * sizeof(T), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && !u) e = hipMemsetAsync(du, 0, (size_t)npts * NU * sizeof(T), stream);
        if (e == hipSuccess) {
            EvalArgs<T> a{};
            a.npts = npts; a.integ = integ < 0 ? cfg.integrator : integ; a.dt = (T)cfg.dt; a.params = params;
            a.x = dx; a.u = du;
            a.f = dout[0]; a.f_x = dout[1]; a.f_u = dout[2]; a.l = dout[3]; a.l_x = dout[4]; a.l_u = dout[5];
            a.l_xx = dout[6]; a.l_ux = dout[7]; a.l_uu = dout[8]; a.l_f = dout[9]; a.l_f_x = dout[10]; a.l_f_xx = dout[11];
            ops.eval(a, stream);
            e = hipGetLastError();
        }
        for (int i = 0; i < 12 && e == hipSuccess; ++i)
            if (outs[i]) e = hipMemcpyAsync(outs[i], dout[i], sizes[i] * npts * sizeof(T), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { err = std::string("eval_points: ") + hipGetErrorString(e); return ILQR_ERR_HIP; }
        return ILQR_OK;
    }

    // ---- MPC ----------------------------------------------------------------------------
    int mpc_reset(const void* x0, const void* U) override {
        int rc = set_problem(x0, U);
        if (rc) return rc;
        if ((rc = up_tc(x0, plant_x, NX, 1))) return rc;
        mpc_ready = true;
        return ILQR_OK;
    }

    // Controller restart that KEEPS the solver state: x_0 and plant state <- x0, warm start <- U, while X, K, U_ff stay
    // what the previous solve left.  run_iLQR_MPC.py warms up with one full optimize_trajectory() (:95) and then enters
    // its loop on the same solver object, so step 0's alpha = 0 rollout is u = U_guess + K_warm (x - X_warm)
    // (SURVEY Q1 / Q2); mpc_reset() is the cold start of run_iLQR_UA_MPC.py, whose warm-up is side-effect free.
    int mpc_rearm(const void* x0, const void* U) override {
        if (!x0 || !U) { err = "mpc_rearm: NULL pointer"; return ILQR_ERR_INVALID_ARG; }
        if (!have_problem) { err = "mpc_rearm before set_problem / mpc_reset"; return ILQR_ERR_STATE; }
        int rc;
        if ((rc = flush_select())) return rc;
        if ((rc = fix_slots(st))) return rc;
        if ((rc = up_tc(x0, st.x0, NX, 1))) return rc;
        if ((rc = up_tc(x0, plant_x, NX, 1))) return rc;
        if ((rc = up_ct(U, st.U, st.cur_slot, NU, N))) return rc;
        mpc_ready = true;
        return ILQR_OK;
    }

    int mpc_run(int n_steps, void* u_out, void* x_out, void* cost_out) override {
        if (!mpc_ready) { err = "mpc_run before mpc_reset"; return ILQR_ERR_STATE; }
        if (cfg.plant_integrator < 0) { err = "mpc_run: the handle was created without a plant integrator"; return ILQR_ERR_STATE; }
        if (n_steps < 1) { err = "mpc_run: n_steps < 1"; return ILQR_ERR_INVALID_ARG; }
        if (n_steps > mpc_log_steps) {
            hipFree(mpc_u_log); hipFree(mpc_x_log); hipFree(mpc_cost_log);
            mpc_u_log = mpc_x_log = mpc_cost_log = nullptr;
            ILQR_HIPCHK(hipMalloc((void**)&mpc_u_log, (size_t)n_steps * NU * B * sizeof(T)));
            ILQR_HIPCHK(hipMalloc((void**)&mpc_x_log, (size_t)n_steps * NX * B * sizeof(T)));
            ILQR_HIPCHK(hipMalloc((void**)&mpc_cost_log, (size_t)n_steps * B * sizeof(T)));
            mpc_log_steps = n_steps;
        }
        if (persist_ok()) {
            // the whole receding-horizon loop on the device: a workgroup's MPC step lasts as long as ITS slowest instance
            MpcArgs<T> m{};
            m.B = B; m.N = N; m.plant_integ = cfg.plant_integrator; m.step = 0; m.dt = (T)cfg.dt; m.params = params;
            m.U = st.U; m.cur_slot = st.cur_slot; m.x0 = st.x0; m.plant_x = plant_x;
            m.u_log = mpc_u_log; m.x_log = mpc_x_log; m.cost_log = mpc_cost_log; m.cost = st.cost;
            int rcp = launch_persist(cfg.maxiter, true, n_steps, &m);
            if (rcp) return rcp;
            have_rollout = true;
        } else
        for (int k = 0; k < n_steps; ++k) {
            int rc = run_solve_loop();
            if (rc) return rc;
            MpcArgs<T> m{};
            m.B = B; m.N = N; m.plant_integ = cfg.plant_integrator; m.step = k; m.dt = (T)cfg.dt; m.params = params;
            m.U = st.U; m.cur_slot = st.cur_slot; m.x0 = st.x0; m.plant_x = plant_x;
            m.u_log = mpc_u_log; m.x_log = mpc_x_log; m.cost_log = mpc_cost_log; m.cost = st.cost;
            timer.begin(ILQR_PHASE_OTHER, stream);
            ops.mpc_advance(m, stream);
            timer.end(stream);
            if ((rc = check_launch())) return rc;
        }
        // the kernel writes the logs in the ABI's own layout [step][B][c]: one plain copy each
        auto fetch = [&](void* host, const T* dev, int C) -> int {
            if (!host) return ILQR_OK;
            ILQR_HIPCHK(hipMemcpyAsync(host, dev, (size_t)n_steps * C * B * sizeof(T), hipMemcpyDeviceToHost, stream));
            return ILQR_OK;
        };
        int rc;
        if ((rc = fetch(u_out, mpc_u_log, NU))) return rc;
        if ((rc = fetch(x_out, mpc_x_log, NX))) return rc;
        if ((rc = fetch(cost_out, mpc_cost_log, 1))) return rc;
        return sync();
    }

    int debug_set_stream(void* sp) override { stream = (hipStream_t)sp; return ILQR_OK; }
    int probe_dump(long long* dst, size_t n) override {
        if (!dst || n > probe_elems) { err = "probe_dump: bad size"; return ILQR_ERR_INVALID_ARG; }
        ILQR_HIPCHK(hipMemcpyAsync(dst, probe, n * sizeof(long long), hipMemcpyDeviceToHost, stream));
        return sync();
    }

    int status_reduce(void* dev_out4) override {
        if (!dev_out4) { err = "status_reduce: NULL pointer"; return ILQR_ERR_INVALID_ARG; }
        if (int rf = flush_select()) return rf;
        timer.begin(ILQR_PHASE_OTHER, stream);
        ILQR_LAUNCH(status_reduce_kernel<T>, dim3(1), dim3(256), 0, stream, st.cost, st.cost_prev, st.status, B,
                           (double*)dev_out4);
        timer.end(stream);
        return check_launch();
    }

    // ---- measurement ----------------------------------------------------------------------
    int timing_enable(int on) override { timer.on = on != 0; return ILQR_OK; }
    int timing_reset() override { timer.reset(stream); return ILQR_OK; }
    int timing_get(double* ms, int64_t* launches) override {
        timer.resolve(stream);
        for (int i = 0; i < ILQR_N_PHASES; ++i) {
            if (ms) ms[i] = timer.ms[i];
            if (launches) launches[i] = timer.n[i];
        }
        return ILQR_OK;
    }
    int algorithmic_bytes(double* bytes) override {
        const double s = sizeof(T), n = NX, m = NU, b = B, Nn = N;
        // dense, no padding, no symmetry packing
        bytes[ILQR_PHASE_LINEARIZE] = b * Nn * s * ((n + m) + (2 * n * n + 2 * n * m + n + m + m * m));
        bytes[ILQR_PHASE_BACKWARD] = b * s * (Nn * (2 * n * n + 3 * n * m + n + 2 * m + m * m) + n * n + n);
        bytes[ILQR_PHASE_FORWARD] = b * Nn * s * ((n + 2 * m + m * n) + (double)A * (n + m));
        bytes[ILQR_PHASE_SELECT] = b * (s * (A + 3) + 4 * 4);
        bytes[ILQR_PHASE_OTHER] = 0;
        // the fused kernel materialises no expansion: it reads the trajectory and the candidates' costs, writes the gains
        bytes[ILQR_PHASE_FUSED] = b * s * (Nn * ((n + m) + (m * n + m)) + n) + bytes[ILQR_PHASE_SELECT];
        // one iteration of the persistent kernel: the fused part + the rollouts
        bytes[ILQR_PHASE_PERSIST] = bytes[ILQR_PHASE_FUSED] + bytes[ILQR_PHASE_FORWARD];
        return ILQR_OK;
    }
};

}  // namespace ilqr

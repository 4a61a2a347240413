// Micro-benchmark: what one lone wave64 pays per instruction on gfx950, by instruction class.  The sweeps and the
// rollout of this library run ONE wave per SIMD (the batch is 4096 trajectories = 1024 waves on 1024 SIMDs), so their
// cost model is the single-wave issue rate, not the throughput of a full CU.  Prints ns per instruction and, from the
// shader clock counter (s_memtime), ticks per instruction.
//   build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define REP8_(X) X X X X X X X X
// 48 copies between two drains: the counter holds 63, and a drain every 8 would charge each instruction 1/8 of a
// memory latency
#define REP8(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X) REP8_(X)
#define DPP " row_mask:0xf bank_mask:0xf bound_ctrl:1"

enum {
    FMA_IND, FMA_DEP, FMAC_DPP_ROR_IND, FMAC_DPP_QP_IND, ADD_DPP_DEP, MIX_FMA_DPP, RCP_IND, RCP_DEP, SALU_IND,
    LOAD_X4_B2B, LOAD_X1_B2B, LOAD_X4_SPREAD, STORE_X1_B2B, MFMA_F32_IND, MFMA_F32_DEP, MFMA_F64_IND, BPERM_DEP,
    READLANE_IND, SWAP32_DEP, PKFMA_IND, FMA64_IND, FMA64_DEP, LOAD_X1_SPREAD, STORE_X1_SPREAD, DS_B128_SPREAD, DS_B32_SPREAD, LDSDMA_SPREAD, FMA7_ONLY, N_MODES
};
static const char* kNames[N_MODES] = {
    "v_fma_f32 independent", "v_fma_f32 dependent", "v_fmac_f32_dpp row_ror independent", "v_fmac_f32_dpp quad_perm independent",
    "v_add_f32_dpp dependent", "v_fma / v_fmac_dpp alternating", "v_rcp_f32 independent", "v_rcp_f32 dependent",
    "s_add_i32 independent", "buffer_load_dwordx4 back to back", "buffer_load_dword back to back",
    "buffer_load_dwordx4 + 7 v_fma each", "buffer_store_dword back to back", "v_mfma_f32_16x16x4 independent",
    "v_mfma_f32_16x16x4 dependent", "v_mfma_f64_16x16x4 independent", "ds_bpermute_b32 dependent", "v_readlane_b32 independent",
    "v_permlane32_swap + v_add dependent (x2)", "v_pk_fma_f32 independent", "v_fma_f64 independent", "v_fma_f64 dependent",
    "buffer_load_dword + 7 v_fma each", "buffer_store_dword + 7 v_fma each", "ds_read_b128 + 7 v_fma each",
    "ds_read_b32 + 7 v_fma each", "global_load_lds_dwordx4 + 7 v_fma each", "7 v_fma (the filler alone)"};
// instructions per loop iteration, per mode (the loop body is written out 32 or 8 times)
static const int kPerIter[N_MODES] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 48, 48, 48, 48, 8, 8, 8, 8, 32, 8, 32, 32, 32, 48, 48, 48, 48, 48, 48};

template <int MODE> __global__ void __launch_bounds__(256) rate(float* buf, long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = a[i];
    const float m = 1.0001f, c = 0.0001f;
    const double md = 1.0001, cd = 0.0001;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 accd[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    f4 ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = f4{0.f, 0.f, 0.f, 0.f};
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    // descriptor over buf (64 KiB, L2-resident)
    i4 srd;
    {
        const unsigned long long p = (unsigned long long)buf;
        srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        srd.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
        srd.z = 1 << 16;
        srd.w = 0x00020000;
    }
    const int voff = (threadIdx.x & 63) * 16, voff1 = (threadIdx.x & 63) * 4 + 32768;
    int perm = ((threadIdx.x * 5 + 3) & 63) * 4;
    __shared__ float lds[4096];
    lds[threadIdx.x] = a[0];
    const int ldsoff = (threadIdx.x & 15) * 16;   // quad-replicated 16-byte reads, as a tile ring would do
    const float* gptr = buf + (threadIdx.x & 63) * 4;
    const int m0v = __builtin_amdgcn_readfirstlane(8192 + (int)(threadIdx.x >> 6) * 1024);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == FMA_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if constexpr (MODE == FMA_DEP) {
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        } else if constexpr (MODE == FMAC_DPP_ROR_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:4" DPP : "+v"(a[i]) : "v"(m), "v"(c));
        } else if constexpr (MODE == FMAC_DPP_QP_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,2,3,0]" DPP : "+v"(a[i]) : "v"(m), "v"(c));
        } else if constexpr (MODE == ADD_DPP_DEP) {
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2]" DPP : "+v"(a[0]));
        } else if constexpr (MODE == MIX_FMA_DPP) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:8" DPP : "+v"(a[i + 1]) : "v"(m), "v"(c));
                }
        } else if constexpr (MODE == RCP_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (MODE == RCP_DEP) {
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[0]));
        } else if constexpr (MODE == SALU_IND) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));
                asm volatile("s_add_i32 %0, %0, 1" : "+s"(s1));
                asm volatile("s_add_i32 %0, %0, 1" : "+s"(s2));
                asm volatile("s_add_i32 %0, %0, 1" : "+s"(s3));
            }
        } else if constexpr (MODE == LOAD_X4_B2B) {
            asm volatile(REP8("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)"
                         : "=&v"(ld[0]) : "v"(voff), "s"(srd) : "memory");
        } else if constexpr (MODE == LOAD_X1_B2B) {
            asm volatile(REP8("buffer_load_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)"
                         : "=&v"(a[7]) : "v"(voff1), "s"(srd) : "memory");
        } else if constexpr (MODE == LOAD_X4_SPREAD) {
            asm volatile(
                REP8("buffer_load_dwordx4 %0, %8, %9, 0 offen\n\t"
                     "v_fma_f32 %1, %1, %10, %11\n\tv_fma_f32 %2, %2, %10, %11\n\tv_fma_f32 %3, %3, %10, %11\n\t"
                     "v_fma_f32 %4, %4, %10, %11\n\tv_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\t"
                     "v_fma_f32 %7, %7, %10, %11\n\t") "s_waitcnt vmcnt(0)"
                : "=&v"(ld[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6])
                : "v"(voff), "s"(srd), "v"(m), "v"(c) : "memory");
        } else if constexpr (MODE == STORE_X1_B2B) {
            asm volatile(REP8("buffer_store_dword %0, %1, %2, 0 offen\n\t") "s_waitcnt vmcnt(0)"
                         : : "v"(a[0]), "v"(voff1), "s"(srd) : "memory");
        } else if constexpr (MODE == MFMA_F32_IND) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc[i], 0, 0, 0);
        } else if constexpr (MODE == MFMA_F32_DEP) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc[0], 0, 0, 0);
        } else if constexpr (MODE == MFMA_F64_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) accd[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[0], d[1], accd[i], 0, 0, 0);
        } else if constexpr (MODE == BPERM_DEP) {
#pragma unroll
            for (int r = 0; r < 8; ++r) perm = __builtin_amdgcn_ds_bpermute(perm & 0xfc, perm);
        } else if constexpr (MODE == READLANE_IND) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(a[0]));
                asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s1) : "v"(a[1]));
                asm volatile("v_readlane_b32 %0, %1, 11" : "=s"(s2) : "v"(a[2]));
                asm volatile("v_readlane_b32 %0, %1, 13" : "=s"(s3) : "v"(a[3]));
            }
        } else if constexpr (MODE == SWAP32_DEP) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[0]), __float_as_uint(a[1]), false, false);
                a[0] = __uint_as_float(sw[0]) + 1.0f;
                a[1] = __uint_as_float(sw[1]);
            }
        } else if constexpr (MODE == PKFMA_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
        } else if constexpr (MODE == FMA64_IND) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
        } else if constexpr (MODE == LOAD_X1_SPREAD || MODE == STORE_X1_SPREAD || MODE == DS_B128_SPREAD ||
                             MODE == DS_B32_SPREAD || MODE == LDSDMA_SPREAD || MODE == FMA7_ONLY) {
#define FILL7 "v_fma_f32 %[a0], %[a0], %[m], %[c]\n\tv_fma_f32 %[a1], %[a1], %[m], %[c]\n\tv_fma_f32 %[a2], %[a2], %[m], %[c]\n\t" \
              "v_fma_f32 %[a3], %[a3], %[m], %[c]\n\tv_fma_f32 %[a4], %[a4], %[m], %[c]\n\tv_fma_f32 %[a5], %[a5], %[m], %[c]\n\t" \
              "v_fma_f32 %[a6], %[a6], %[m], %[c]\n\t"
#define SPREAD_ASM(MEM, TAIL)                                                                                          \
    asm volatile("s_mov_b32 m0, %[m0v]\n\t" REP8(MEM FILL7) TAIL                                                      \
                 : [d4] "+v"(ld[0]), [d1] "+v"(a[7]), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), \
                   [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6])                                                   \
                 : [vo] "v"(voff1), [lo] "v"(ldsoff), [srd] "s"(srd), [m] "v"(m), [c] "v"(c), [gp] "v"(gptr), [m0v] "s"(m0v) \
                 : "memory", "m0")
            if constexpr (MODE == LOAD_X1_SPREAD) SPREAD_ASM("buffer_load_dword %[d1], %[vo], %[srd], 0 offen\n\t", "s_waitcnt vmcnt(0)");
            if constexpr (MODE == STORE_X1_SPREAD) SPREAD_ASM("buffer_store_dword %[a0], %[vo], %[srd], 0 offen\n\t", "s_waitcnt vmcnt(0)");
            if constexpr (MODE == DS_B128_SPREAD) SPREAD_ASM("ds_read_b128 %[d4], %[lo]\n\t", "s_waitcnt lgkmcnt(0)");
            if constexpr (MODE == DS_B32_SPREAD) SPREAD_ASM("ds_read_b32 %[d1], %[lo]\n\t", "s_waitcnt lgkmcnt(0)");
            if constexpr (MODE == LDSDMA_SPREAD) SPREAD_ASM("global_load_lds_dwordx4 %[gp], off\n\t", "s_waitcnt vmcnt(0)");
            if constexpr (MODE == FMA7_ONLY) SPREAD_ASM("", "");
        } else if constexpr (MODE == FMA64_DEP) {
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(md), "v"(cd));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = (float)(s0 + s1 + s2 + s3 + perm);
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + ld[i].x;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    s += (float)(accd[0].x + accd[1].y);
    if (s == 12345.678f) buf[20000 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE> void run(float* buf, long long* cyc, int threads = 64) {
    const int iters = 5000;
    float best = 1e30f;
    long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        rate<MODE><<<1, threads>>>(buf, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    const double n = (double)iters * kPerIter[MODE];
    printf("%-40s %d wave%s %7.2f ns/instr  %7.2f ticks/instr\n", kNames[MODE], threads / 64, threads > 64 ? "s" : " ", best * 1e6 / n,
           (double)c / n);
    fflush(stdout);
}

template <int M> void run_all(float* buf, long long* cyc) {
    if constexpr (M < N_MODES) {
        run<M>(buf, cyc);
        run_all<M + 1>(buf, cyc);
    }
}

int main() {
    float* buf;
    long long* cyc;
    hipMalloc(&buf, 1 << 17);
    hipMemset(buf, 0, 1 << 17);
    hipMalloc(&cyc, 8);
    run_all<0>(buf, cyc);
    // the same with one wave on each of the CU's four SIMDs: what is shared (the vector-memory path) shows up here
    run<FMA_IND>(buf, cyc, 256);
    run<LOAD_X4_B2B>(buf, cyc, 256);
    run<LOAD_X1_B2B>(buf, cyc, 256);
    run<LOAD_X4_SPREAD>(buf, cyc, 256);
    run<STORE_X1_B2B>(buf, cyc, 256);
    run<BPERM_DEP>(buf, cyc, 256);
    run<FMA7_ONLY>(buf, cyc, 256);
    run<LOAD_X1_SPREAD>(buf, cyc, 256);
    run<STORE_X1_SPREAD>(buf, cyc, 256);
    run<DS_B128_SPREAD>(buf, cyc, 256);
    run<DS_B32_SPREAD>(buf, cyc, 256);
    run<LDSDMA_SPREAD>(buf, cyc, 256);
    return 0;
}

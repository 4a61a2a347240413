// Micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 for a lone wave64 on gfx950 (cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(64) rate(float* out, long long* cyc, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2)(a[i], a[i] + 0.5f); }
    const float m = 1.0001f, c = 0.0001f;
    const f2 m2 = (f2)(m, m), c2 = (f2)(c, c);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        } else if (MODE == 2) {   // dependent chain, scalar
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        } else {                  // dependent chain, packed
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(m2), "v"(c2));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char* names[4] = {"v_fma_f32 independent x8", "v_pk_fma_f32 independent x8", "v_fma_f32 dependent", "v_pk_fma_f32 dependent"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<1, 64>>>(out, cyc, iters);
            if (mode == 1) rate<1><<<1, 64>>>(out, cyc, iters);
            if (mode == 2) rate<2><<<1, 64>>>(out, cyc, iters);
            if (mode == 3) rate<3><<<1, 64>>>(out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-32s %.3f ms, %.2f ns/instr, counter ticks/instr %.3f\n", names[mode], ms, ms * 1e6 / (iters * 32.0), (double)c / (iters * 32.0));
        }
    }
    return 0;
}

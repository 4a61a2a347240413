// How many wait states does a packed-FP32 write of a wide store's data registers need behind the store?  (gfx950)
// kinds: buffer store with SGPR soffset / literal soffset, global_store_dwordx4; followers v_pk_add_f32 on dwords [0:1]
// and [2:3]; between them 0..4 wait states (s_nop) or unrelated VALU instructions.  See store_hazard.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define PRE "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v44, 0x42c80000\n\tv_mov_b32 v45, 0x42c80000\n\tv_mov_b32 v46, 0\n\ts_nop 4\n\t"
#define FOLLOW "v_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\ts_waitcnt vmcnt(0)"
#define FOLLOW_F64 "v_fma_f64 v[40:41], v[44:45], v[44:45], v[40:41]\n\tv_add_f64 v[42:43], v[42:43], v[44:45]\n\ts_waitcnt vmcnt(0)"
#define FOLLOW_MOV "v_pk_mov_b32 v[40:41], v[44:45], v[44:45]\n\tv_mov_b64 v[42:43], v[44:45]\n\ts_waitcnt vmcnt(0)"
#define ARGS : : "v"(voff), "s"(srd), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff), "v"(out + 4 * lane) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46"
#define ST0 "buffer_store_dwordx4 v[40:43], %0, %1, %6 offen\n\t"
#define ST1 "buffer_store_dwordx4 v[40:43], %0, %1, 0 offen\n\t"
#define ST2 "global_store_dwordx4 %7, v[40:43], off\n\t"
#define ST3 "buffer_store_dwordx4 v[40:43], off, %1, %6\n\t"
#define GAP0 ""
#define GAP1 "s_nop 0\n\t"
#define GAP2 "s_nop 1\n\t"
#define GAP3 "s_nop 2\n\t"
#define GAP4 "s_nop 3\n\t"
#define GAP5 "v_add_f32 v46, v46, v44\n\t"
#define GAP6 "v_add_f32 v46, v46, v44\n\tv_add_f32 v46, v46, v44\n\t"
#define GAP7 "v_pk_add_f32 v[46:47], v[46:47], v[44:45]\n\t"

template <int K, int G> __global__ void k(float* out, int n_bytes, int soff_in) {
    const int lane = threadIdx.x + blockIdx.x * blockDim.x;
    i32x4 srd;
    const unsigned long long p = (unsigned long long)out;
    srd.x = (int)(p & 0xffffffffu); srd.y = (int)((p >> 32) & 0xffff); srd.z = n_bytes; srd.w = 0x00020000;
    const int voff = lane * 16;
    const int soff = __builtin_amdgcn_readfirstlane(soff_in);
    const float a = 1.0f + lane, b = 2.0f + lane, c = 3.0f + lane, d = 4.0f + lane;
#define CASE(KK, GG, ST, GAP) if constexpr (K == KK && G == GG) asm volatile(PRE ST GAP FOLLOW ARGS)
#define ROW(KK, ST) CASE(KK, 0, ST, GAP0); CASE(KK, 1, ST, GAP1); CASE(KK, 2, ST, GAP2); CASE(KK, 3, ST, GAP3); CASE(KK, 4, ST, GAP4); CASE(KK, 5, ST, GAP5); CASE(KK, 6, ST, GAP6);
    ROW(0, ST0) ROW(1, ST1) ROW(2, ST2)
#undef CASE
#define CASE(KK, GG, ST, GAP) if constexpr (K == KK && G == GG) asm volatile(PRE ST GAP FOLLOW_F64 ARGS)
    ROW(3, ST0) ROW(4, ST1) ROW(5, ST2)
#undef CASE
#define CASE(KK, GG, ST, GAP) if constexpr (K == KK && G == GG) asm volatile(PRE ST GAP FOLLOW_MOV ARGS)
    ROW(6, ST0) ROW(7, ST1) ROW(8, ST2)
}

template <int K, int G> void run() {
    const int blocks = 1024, threads = 64, n = blocks * threads;
    float* d;
    (void)hipMalloc(&d, n * 16);
    int bad[4] = {0, 0, 0, 0};
    std::vector<float> h(n * 4);
    for (int rep = 0; rep < 10; ++rep) {
        (void)hipMemset(d, 0, n * 16);
        hipLaunchKernelGGL((k<K, G>), dim3(blocks), dim3(threads), 0, 0, d, n * 16, 0);
        (void)hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < 4; ++q)
                if (h[4 * i + q] != (float)(q + 1 + i)) ++bad[q];
    }
    static const char* kn[] = {"buffer x4, SGPR soffset  ", "buffer x4, literal soffset", "global_store_dwordx4     ",
                               "f64 fma/add: buffer SGPR ", "f64 fma/add: buffer lit  ", "f64 fma/add: global      ",
                               "pk_mov/mov_b64: buf SGPR ", "pk_mov/mov_b64: buf lit  ", "pk_mov/mov_b64: global   "};
    static const char* gn[] = {"nothing", "s_nop 0", "s_nop 1", "s_nop 2", "s_nop 3", "1 VALU", "2 VALU"};
    printf("%s | %-8s | clobbered %7d %7d %7d %7d of %d\n", kn[K], gn[G], bad[0], bad[1], bad[2], bad[3], 10 * n);
    (void)hipFree(d);
}
template <int K> void row() { run<K, 0>(); run<K, 1>(); run<K, 2>(); run<K, 3>(); run<K, 4>(); run<K, 5>(); run<K, 6>(); }
int main() { row<0>(); row<1>(); row<2>(); row<3>(); row<4>(); row<5>(); row<6>(); row<7>(); row<8>(); return 0; }

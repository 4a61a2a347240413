// Does a VALU write of the data registers right behind a 16-byte buffer store (SGPR soffset) corrupt the stored data?
// LLVM's hazard recognizer pads "VMEM store of more than 64 bits followed by a write of its data VGPRs" only when the
// store has NO SGPR soffset (GCNHazardRecognizer::createsVALUHazard).  This test issues the pair back to back in one asm
// block (nothing can be scheduled in between) for several followers and prints how many lanes stored a clobbered value.
//   hipcc --offload-arch=gfx950 -O2 -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int V> __global__ void k(float* out, int n_bytes, int soff_in) {
    const int lane = threadIdx.x + blockIdx.x * blockDim.x;
    i32x4 srd;
    const unsigned long long p = (unsigned long long)out;
    srd.x = (int)(p & 0xffffffffu);
    srd.y = (int)((p >> 32) & 0xffff);
    srd.z = n_bytes;
    srd.w = 0x00020000;
    const int voff = lane * 16;
    const int soff = __builtin_amdgcn_readfirstlane(soff_in);
    const float a = 1.0f + lane, b = 2.0f + lane, c = 3.0f + lane, d = 4.0f + lane;
#define PRE "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v44, 0x42c80000\n\tv_mov_b32 v45, 0x42c80000\n\ts_nop 4\n\t"
#define ST "buffer_store_dwordx4 v[40:43], %0, %1, %6 offen\n\t"
#define POST "s_waitcnt vmcnt(0)"
#define ARGS : : "v"(voff), "s"(srd), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff) : "memory", "v40", "v41", "v42", "v43", "v44", "v45"
    if constexpr (V == 0) asm volatile(PRE ST "v_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 1) asm volatile(PRE ST "v_add_f32 v40, v40, v44\n\tv_add_f32 v41, v41, v44\n\tv_add_f32 v42, v42, v44\n\tv_add_f32 v43, v43, v44\n\t" POST ARGS);
    if constexpr (V == 2) asm volatile(PRE ST "s_nop 0\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 3) asm volatile(PRE ST "v_mov_b32 v41, v44\n\tv_mov_b32 v40, v44\n\tv_mov_b32 v43, v44\n\tv_mov_b32 v42, v44\n\t" POST ARGS);
    if constexpr (V == 4) asm volatile(PRE ST "v_pk_fma_f32 v[40:41], v[44:45], v[44:45], v[40:41]\n\tv_pk_fma_f32 v[42:43], v[44:45], v[44:45], v[42:43]\n\t" POST ARGS);
    if constexpr (V == 6) asm volatile(PRE "global_store_dwordx4 %7, v[40:43], off\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST
                                       : : "v"(voff), "s"(srd), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff), "v"(out + 4 * lane) : "memory", "v40", "v41", "v42", "v43", "v44", "v45");
    if constexpr (V == 7) asm volatile(PRE "buffer_store_dwordx2 v[40:41], %0, %1, %6 offen\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tbuffer_store_dwordx2 v[42:43], %0, %1, %6 offen offset:8\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 8) asm volatile(PRE "buffer_store_dwordx4 v[40:43], %0, %1, 0 offen\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 9) asm volatile(PRE ST "v_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 10) asm volatile(PRE ST "v_add_f32 v44, v44, v45\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 11) asm volatile(PRE "buffer_store_dwordx3 v[40:42], %0, %1, %6 offen\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\tbuffer_store_dword v43, %0, %1, %6 offen offset:12\n\t" POST ARGS);
    if constexpr (V == 12) asm volatile(PRE ST "v_pk_mul_f32 v[40:41], v[40:41], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 14) asm volatile(PRE ST "v_fma_f64 v[40:41], v[44:45], v[44:45], v[40:41]\n\t" POST ARGS);
    if constexpr (V == 15) asm volatile(PRE ST "v_mov_b64 v[40:41], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 16 || V == 17) {
        __shared__ float lds[64 * 4];
        const int la = (int)(threadIdx.x * 16);
        if constexpr (V == 16) asm volatile(PRE "ds_write_b128 %7, v[40:43]\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\ts_waitcnt lgkmcnt(0)\n\t" POST
                                            : : "v"(voff), "s"(srd), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff), "v"(la) : "memory", "v40", "v41", "v42", "v43", "v44", "v45");
        if constexpr (V == 17) asm volatile(PRE "ds_write_b64 %7, v[40:41]\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tds_write_b64 %7, v[42:43] offset:8\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\ts_waitcnt lgkmcnt(0)\n\t" POST
                                            : : "v"(voff), "s"(srd), "v"(a), "v"(b), "v"(c), "v"(d), "s"(soff), "v"(la) : "memory", "v40", "v41", "v42", "v43", "v44", "v45");
        __syncthreads();
        for (int q = 0; q < 4; ++q) out[4 * lane + q] = lds[4 * threadIdx.x + q];
    }
    if constexpr (V == 18) asm volatile(PRE "buffer_store_dwordx4 v[40:43], %0, %1, 0 offen\n\ts_nop 0\n\tv_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[42:43], v[44:45]\n\t" POST ARGS);
    if constexpr (V == 5) asm volatile(PRE ST "s_nop 1\n\tv_pk_fma_f32 v[40:41], v[44:45], v[44:45], v[40:41]\n\tv_pk_fma_f32 v[42:43], v[44:45], v[44:45], v[42:43]\n\t" POST ARGS);
}

template <int V> void run(const char* what) {
    const int blocks = 1024, threads = 64, n = blocks * threads;
    float* d;
    hipMalloc(&d, n * 16);
    int bad[4] = {0, 0, 0, 0};
    std::vector<float> h(n * 4);
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(d, 0, n * 16);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, n * 16, 0);
        hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < 4; ++q)
                if (h[4 * i + q] != (float)(q + 1 + i)) ++bad[q];
    }
    printf("%-58s clobbered dwords: %d %d %d %d (of %d each)\n", what, bad[0], bad[1], bad[2], bad[3], 20 * n);
    hipFree(d);
}

int main() {
    run<0>("store; v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<1>("store; v_add v0; v_add v1; v_add v2; v_add v3");
    run<2>("store; s_nop 0; v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<3>("store; v_mov v1; v_mov v0; v_mov v3; v_mov v2");
    run<4>("store; v_pk_fma v[0:1]; v_pk_fma v[2:3]");
    run<5>("store; s_nop 1; v_pk_fma v[0:1]; v_pk_fma v[2:3]");
    run<6>("global_store_dwordx4; v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<7>("store_dwordx2 v[0:1]; v_pk_add v[0:1]; same for v[2:3]");
    run<8>("store (soffset literal 0); v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<9>("store; v_pk_add v[2:3]; v_pk_add v[0:1]");
    run<10>("store; v_add (unrelated); v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<11>("store_dwordx3 v[0:2]; v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<12>("store; v_pk_mul v[0:1]");
    run<14>("store; v_fma_f64 v[0:1]");
    run<15>("store; v_mov_b64 v[0:1]");
    run<16>("ds_write_b128; v_pk_add v[0:1]; v_pk_add v[2:3]");
    run<17>("ds_write_b64 v[0:1]; v_pk_add v[0:1]; same for v[2:3]");
    run<18>("store (soffset literal 0); s_nop 0; v_pk_add v[0:1]; v_pk_add v[2:3]");
    return 0;
}

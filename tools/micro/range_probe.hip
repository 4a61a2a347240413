// range_probe.hip -- which raw-buffer stores does gfx950 drop?  ("dropped stores")
//
// The fp32 sweep and rollout switch a lane's store off by giving it a voffset beyond the descriptor's
// num_records instead of masking EXEC.  This probe measures the rule the hardware applies, for 4-, 8- and
// 16-byte stores, with the byte offset split between voffset (VGPR, offen), soffset (SGPR) and the
// instruction's immediate -- on an allocation large enough (9 GiB, base 2.5 GiB in) that a store the range check lets
// through lands inside the allocation wherever base + soffset + voffset + imm points (mod 2^32 or not).
//
//   hipcc --offload-arch=gfx950 -O2 -o range_probe range_probe.hip && ./range_probe
//
// Output: one line per case: "w=<bytes> nr=<num_records> voff=<hex> soff=<hex> imm=<n> -> landed <count> dwords
// [first byte offset from base]" -- "landed 0" = dropped.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(e)                                                                             \
    do {                                                                                   \
        hipError_t r_ = (e);                                                               \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Case { unsigned nr, voff, soff; int width, imm; };

// one wave; lane 0 stores the pattern with the case's offsets, every other lane stores at a harmless in-range
// offset 64 + 16 * lane (so the instruction is a mixed one, like the product's), unless solo != 0
__global__ void store_kernel(char* base, Case c, int solo) {
    const int lane = threadIdx.x;
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (unsigned)__builtin_amdgcn_readfirstlane((int)c.nr), 0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane((int)c.soff);
    if (solo && lane != 0) return;
    const int voff = lane == 0 ? (int)c.voff : 64 + 16 * lane;
    const unsigned pat = 0xabcd0000u + lane;
    if (c.width == 4) {
        if (c.imm) __builtin_amdgcn_raw_buffer_store_b32(pat, r, voff + 2048, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(pat, r, voff, soff, 0);
    } else if (c.width == 8) {
        u32x2 v = {pat, pat + 0x100};
        if (c.imm) __builtin_amdgcn_raw_buffer_store_b64(v, r, voff + 2048, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
    } else {
        u32x4 v = {pat, pat + 0x100, pat + 0x200, pat + 0x300};
        if (c.imm) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + 2048, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    }
}

// counts dwords that carry lane 0's pattern (0xabcd0000 + k*0x100) anywhere in the allocation
__global__ void scan_kernel(const unsigned* p, size_t n, unsigned long long* count, unsigned long long* first) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const unsigned v = p[i];
        if ((v & 0xfffffcffu) == 0xabcd0000u) {
            atomicAdd(count, 1ull);
            atomicMin(first, (unsigned long long)i * 4);
        }
    }
}

int main() {
    // the descriptor's base sits 2.5 GiB into a 9 GiB allocation: a store that got through a signed or wrapped
    // address computation still lands inside it
    const size_t total = 9ull << 30, lead = 5ull << 29, bytes = total - lead;
    char* alloc = nullptr;
    CHK(hipMalloc((void**)&alloc, total));
    CHK(hipMemset(alloc, 0, total));
    char* buf = alloc + lead;
    unsigned long long *d_count, *d_first;
    CHK(hipMalloc((void**)&d_count, 8));
    CHK(hipMalloc((void**)&d_first, 8));
    std::vector<Case> cases;
    const unsigned NRS[] = {1u << 20, 52428800u /* c3 f64 gains */, 0x7fff0000u};
    for (unsigned nr : NRS)
        for (int w : {4, 8, 16}) {
            const unsigned vo[] = {nr - 16, nr - (unsigned)w, nr - (unsigned)w + 4, nr - 4, nr, nr + 4, 0x7ffffff0u, 0x7ffffff8u,
                                   0x7ffffffcu, 0x80000000u, 0xfffffff0u};
            for (unsigned v : vo)
                for (unsigned so : {0u, 4096u, nr / 2, nr - 64})
                    cases.push_back({nr, v, so, w, 0});
            // immediate offset on top of the product's "dropped" voffset
            cases.push_back({nr, 0x7ffffff0u, 0u, w, 1});
            cases.push_back({nr, nr - 2048 - 16, 0u, w, 1});
            cases.push_back({nr, nr - 2048, 0u, w, 1});
        }
    int n_landed_beyond = 0;
    for (int solo = 0; solo < 2; ++solo)
        for (const Case& c : cases) {
            CHK(hipMemset(buf, 0, 4096));   // the harmless lanes' area
            // clear the places a store could have landed in the previous case: cheap full clear every case would cost
            // 6 GiB x cases; instead clear the candidate windows
            const unsigned long long cand[] = {(unsigned long long)c.voff + c.soff + (c.imm ? 2048 : 0),
                                               ((unsigned long long)c.voff + c.soff + (c.imm ? 2048 : 0)) & 0xffffffffull,
                                               (unsigned long long)c.voff + (c.imm ? 2048 : 0), (unsigned long long)c.soff};
            for (unsigned long long o : cand) {
                const unsigned long long s = o > 64 ? o - 64 : 0;
                if (s + 256 <= bytes) CHK(hipMemset(buf + s, 0, 256));
            }
            CHK(hipMemset(d_count, 0, 8));
            CHK(hipMemset(d_first, 0xff, 8));
            hipLaunchKernelGGL(store_kernel, dim3(1), dim3(64), 0, 0, buf, c, solo);
            CHK(hipGetLastError());
            CHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(scan_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned*)alloc, total / 4, d_count, d_first);
            CHK(hipDeviceSynchronize());
            unsigned long long count = 0, first = 0;
            CHK(hipMemcpy(&count, d_count, 8, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&first, d_first, 8, hipMemcpyDeviceToHost));
            const unsigned long long off = (unsigned long long)c.voff + (c.imm ? 2048 : 0);   // what the range check sees (if soffset is excluded)
            const bool expect_in = off + c.width <= c.nr;
            const bool landed = count != 0;
            if (landed && !expect_in) ++n_landed_beyond;
            printf("solo=%d w=%2d nr=%#10x voff=%#10x soff=%#10x imm=%4d -> landed %llu dwords", solo, c.width, c.nr, c.voff, c.soff,
                   c.imm ? 2048 : 0, count);
            if (landed) printf(" at base%+lld", (long long)first - (long long)lead);
            printf("   [voff+imm+w<=nr: %s]%s\n", expect_in ? "in" : "out", landed == expect_in ? "" : "   <-- differs from the voffset-only rule");
            if (landed) CHK(hipMemset(alloc + (first & ~63ull), 0, 128));
        }
    printf("stores that landed although voffset + imm + width > num_records: %d\n", n_landed_beyond);
    (void)hipFree(alloc);
    return 0;
}

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
#include <type_traits>
#include <utility>

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

// ---- buffer addressing (SRSRC): address = base + voffset(lane, set once) + soffset(scalar, per step) +
// immediate.  The sweep's per-step address arithmetic is then ONE scalar multiply instead of four 64-bit
// vector adds per tile (cdna_hip_programming.md T8: it pays "as part of other addressing idioms").
// cache policy of the tile stream (aux = 2: nt -- each tile byte is read exactly once per sweep)
constexpr int kTileAux = ILQR_NT_TILE_LOAD ? 2 : 0;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Uniformity must be PROVABLE to hipcc or every buffer op is wrapped in a ~10-instruction waterfall loop
// (cdna_hip_programming.md T20): the descriptor inputs and every soffset go through readfirstlane.
ILQR_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
ILQR_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = (unsigned)uniform((int)(unsigned)a), hi = (unsigned)uniform((int)(unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (unsigned)uniform((int)bytes), 0x00020000);
}
// 4 scalars of T at voff + IMM_SCALARS*sizeof(T)
template <int IMM, typename T> struct BufLoad;
template <int IMM> struct BufLoad<IMM, float> {
    static ILQR_DEV void v4(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* o) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM * 4, soff, kTileAux);
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    static ILQR_DEV float v1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff + IMM * 4, soff, kTileAux));
    }
};
template <int IMM> struct BufLoad<IMM, double> {
    static ILQR_DEV void v4(__amdgpu_buffer_rsrc_t r, int voff, int soff, double* o) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM * 8, soff, kTileAux);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM * 8 + 16, soff, kTileAux);
        o[0] = __hiloint2double((int)a.y, (int)a.x); o[1] = __hiloint2double((int)a.w, (int)a.z);
        o[2] = __hiloint2double((int)b.y, (int)b.x); o[3] = __hiloint2double((int)b.w, (int)b.z);
    }
    static ILQR_DEV double v1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff + IMM * 8, soff, kTileAux);
        return __hiloint2double((int)a.y, (int)a.x);
    }
};
ILQR_DEV void buf_store1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
ILQR_DEV void buf_store1(__amdgpu_buffer_rsrc_t r, int voff, int soff, double v) {
    u32x2 w = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, soff, 0);
}

// a C-vector of T in 16-, 8- or 4-byte pieces (vec_pieces(C * sizeof(T)) stores)
template <typename T, int C> ILQR_DEV void buf_store_vec(__amdgpu_buffer_rsrc_t r, int voff, int soff, const T* v) {
    constexpr int BYTES = C * (int)sizeof(T), PB = BYTES % 16 == 0 ? 16 : BYTES % 8 == 0 ? 8 : 4, NP = BYTES / PB;
    unsigned w[BYTES / 4];
    __builtin_memcpy(w, v, BYTES);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if constexpr (PB == 16) {
            // A packed-FP32 write of the data registers scheduled right behind a 16-byte store lands its high half before
            // the store has read it -- one wait state later than the hazard hipcc knows and pads (gfx950;
            // tools/micro/store_hazard2.hip, verify_ring_isa.store_pk_hazards).  The store stays a builtin (hipcc pads
            // what it knows, e.g. an SGPR operand fresh from v_readlane: nothing inside an asm statement is padded, and a
            // store written as asm with a spilled descriptor restored right in front of it went astray).  The wait state
            // is an asm statement that READS the data registers: they stay live up to it, so nothing can overwrite them
            // before it has executed, wherever the scheduler puts it.
            const u32x4 q = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(q, r, voff + 16 * k, soff, 0);
            asm volatile("s_nop 0" : : "v"(q));
        } else if constexpr (PB == 8) {
            const u32x2 q = {w[2 * k], w[2 * k + 1]};
            __builtin_amdgcn_raw_buffer_store_b64(q, r, voff + 8 * k, soff, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(w[k], r, voff + 4 * k, soff, 0);
        }
    }
}

// per-lane byte offsets into this trajectory's tile, fixed for the whole sweep
struct TileOffsets { int vi, vj, vl; };

template <typename T> struct Tile16 {
    T ski[4];  // SK[i][0..3]: column i of A, for the contraction down the rows
    T skj[4];  // SK[j][0..3]: column j of A, for the contractions along the row
    T vj[4];   // f_u[j], l_x[j], l_ux[j], e_j
    T lxx;     // l_xx[i][j]
    T bi;      // f_u[i]
    T luxi;    // l_ux[i]   (row form, for the transpose-free fp32 step)
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
    tl.luxi = tp[32 + 4 * i + 2];
}

template <typename T>
ILQR_DEV void tile16_load_buf(Tile16<T>& tl, __amdgpu_buffer_rsrc_t r, const TileOffsets& o, int soff) {
    BufLoad<0, T>::v4(r, o.vi, soff, tl.ski);      // SK[i][0..3]
    BufLoad<0, T>::v4(r, o.vj, soff, tl.skj);      // SK[j][0..3]
    BufLoad<32, T>::v4(r, o.vj, soff, tl.vj);      // f_u[j], l_x[j], l_ux[j], e_j
    tl.lxx = BufLoad<16, T>::v1(r, o.vl, soff);    // l_xx[i][j]
    tl.bi = BufLoad<32, T>::v1(r, o.vi, soff);     // f_u[i]
    tl.luxi = BufLoad<34, T>::v1(r, o.vi, soff);   // l_ux[i]
}

// ---- the fp32 tile as the un-regularised fp32 step takes it ---------------------------------------------------
// What bounds the fp32 sweep is neither HBM nor the wave's own instruction stream but the CU's vector-memory RETURN
// path: 64 B/clk, shared by the four waves (tools/micro/issue_rate.hip: a 16-byte-per-lane load costs a lone wave
// 16 cycles and four waves 64).  A lane (i, j) needs 15 of the tile's 48 scalars, and loading each of them into every
// lane that needs it -- 17 dwords per lane and step -- kept that path busy for 4 waves x 17 x 4 = 272 of the step's
// ~385 cycles.  The scalars that depend only on the ROW i (SK[i][0..3], f_u[i], l_ux[i]) are the same in the four lanes
// of a quad, so here each lane loads ONE element of them (a = tile[l16] = SK[i][j], c = tile[32 + l16]: lane (i, 0) gets
// f_u[i], lane (i, 2) gets l_ux[i]) and the quad shares them by DPP broadcast (quad_perm:[d,d,d,d]), folded into the
// consuming instruction where that instruction has no other lane move, else a v_mov_b32_dpp the wave has issue slots to
// spare for: 11 dwords per lane and step, 4 more VALU instructions.  The column-dependent scalars (SK[j][..], f_u[j] ...)
// sit in lanes 4 apart, where no DPP pattern is a broadcast; they stay replicated loads.
struct TileQ {
    float skj[4];  // SK[j][0..3]
    float vj[4];   // f_u[j], l_x[j], l_ux[j], e_j
    float a;       // SK[i][j]: the quad holds SK[i][0..3]
    float lxx;     // l_xx[i][j]
    float c;       // tile[32 + 4i + j]: f_u[i] on lane j = 0, l_ux[i] on lane j = 2
};
ILQR_DEV void tileq_load_buf(TileQ& tl, __amdgpu_buffer_rsrc_t r, const TileOffsets& o, int soff) {
    BufLoad<0, float>::v4(r, o.vj, soff, tl.skj);
    BufLoad<32, float>::v4(r, o.vj, soff, tl.vj);
    tl.a = BufLoad<0, float>::v1(r, o.vl, soff);
    tl.lxx = BufLoad<16, float>::v1(r, o.vl, soff);
    tl.c = BufLoad<32, float>::v1(r, o.vl, soff);
}

// ---- tile loads hipcc does not count -------------------------------------------------------------------
// hipcc places its own s_waitcnt for loads it can see, and at a loop edge it drains them all (measured:
// vmcnt(0..5) at the top of every ring pass = one exposed memory latency per D steps; with the tiles coming
// from HBM rather than from the Infinity Cache that is ~2 us per pass).  The ring's loads are therefore
// issued from inline asm, which hipcc's bookkeeping does not see, and the kernel counts vmcnt itself
// (cdna_hip_programming.md 5.7, form (ii): "=v" loads, then before the first consumer a wait statement
// naming every destination "+v").  Loads, stores and LDS-DMA retire in issue order; one step issues
// NLOAD tile loads and exactly one gain store, so "slot u has landed" == at most (D-1)*(NLOAD+1) younger
// operations outstanding in steady state, (D-1)*NLOAD while the prologue's loads are still the only ones.
#if ILQR_NT_TILE_LOAD
#define ILQR_TILE_NT " nt"
#else
#define ILQR_TILE_NT ""
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef double f64x2n __attribute__((ext_vector_type(2)));

ILQR_DEV i32x4 make_srd(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 d;
    d.x = uniform((int)(unsigned)a);
    d.y = uniform((int)((unsigned)(a >> 32) & 0xffffu));   // stride 0
    d.z = uniform((int)bytes);
    d.w = 0x00020000;
    return d;
}

template <int... I, typename F> ILQR_DEV void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): an unrolled loop whose index is a constant expression
template <int N, typename F> ILQR_DEV void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <typename T> struct RawTile;
// A REFILL ties every destination to the slot's previous value ("+&v"): the statement then redefines the slot's own
// registers, so whatever still needs the consumed tile must read it (or copy it) BEFORE the loads are issued.  With
// plain outputs hipcc is free to sink the tail of the step below the issue; the old tile is then still live, the new
// loads get a temporary register, and the temporary is copied into the slot's register later -- a copy of a register
// whose load may not have landed (this produced wrong fp64 gains in round 1 and sat unnoticed in the mu > 0 fp32
// sweep; csrc/verify_ring_isa.py now rejects any build that touches an in-flight destination).  The prologue's first
// issue has nothing to tie to and uses early-clobber outputs.
#define ILQR_OUT_FIRST(x) "=&v"(x)
#define ILQR_OUT_REFILL(x) "+&v"(x)
// All loads of a tile are ONE asm statement opening with s_nop 4: every SGPR operand (descriptor, soffset) is
// then materialised before the statement, and a value hipcc produced with a VALU (v_readfirstlane, or a
// v_readlane restoring a spilled SGPR) has its 5 wait states before a buffer instruction reads it -- hipcc
// cannot pad hazards inside an asm string (cdna_hip_programming.md 5.7 item 2).  Without this the f64 rollout,
// which runs out of SGPRs, read a half-restored descriptor and faulted.  Outputs are early-clobber: a later
// load of the statement must not see its address register overwritten by an earlier one.
template <> struct RawTile<float> {
    static constexpr int NLOAD = 5;
    f32x4n ski, skj, vj, vi;   // vi = {f_u[i], l_x[i], l_ux[i], e_i}: the row-form twin of vj (f_u[i] and l_ux[i] used)
    float lxx;
#define ILQR_RAWTILE_F32(OUT)                                                                \
    asm volatile(                                                                            \
        "s_nop 4\n\t"                                                                        \
        "buffer_load_dwordx4 %0, %5, %8, %9 offen" ILQR_TILE_NT "\n\t"                       \
        "buffer_load_dwordx4 %1, %6, %8, %9 offen" ILQR_TILE_NT "\n\t"                       \
        "buffer_load_dwordx4 %2, %6, %8, %9 offen offset:128" ILQR_TILE_NT "\n\t"            \
        "buffer_load_dword %3, %7, %8, %9 offen offset:64" ILQR_TILE_NT "\n\t"               \
        "buffer_load_dwordx4 %4, %5, %8, %9 offen offset:128" ILQR_TILE_NT                   \
        : OUT(ski), OUT(skj), OUT(vj), OUT(lxx), OUT(vi)                                     \
        : "v"(o.vi), "v"(o.vj), "v"(o.vl), "s"(srd), "s"(soff)                               \
        : "memory")
    template <bool FIRST> ILQR_DEV void issue(const i32x4& srd, const TileOffsets& o, int soff) {
        if constexpr (FIRST) ILQR_RAWTILE_F32(ILQR_OUT_FIRST);
        else ILQR_RAWTILE_F32(ILQR_OUT_REFILL);
    }
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(ski), "+v"(skj), "+v"(vj), "+v"(lxx), "+v"(vi) : "i"(N) : "memory");
    }
    ILQR_DEV void unpack(Tile16<float>& t) const {
        t.ski[0] = ski.x; t.ski[1] = ski.y; t.ski[2] = ski.z; t.ski[3] = ski.w;
        t.skj[0] = skj.x; t.skj[1] = skj.y; t.skj[2] = skj.z; t.skj[3] = skj.w;
        t.vj[0] = vj.x; t.vj[1] = vj.y; t.vj[2] = vj.z; t.vj[3] = vj.w;
        t.lxx = lxx; t.bi = vi.x; t.luxi = vi.z;
    }
};
template <> struct RawTile<double> {
    static constexpr int NLOAD = 8;
    f64x2n ski0, ski1, skj0, skj1, vj0, vj1;
    double lxx, bi;
#define ILQR_RAWTILE_F64(OUT)                                                                \
    asm volatile(                                                                            \
        "s_nop 4\n\t"                                                                        \
        "buffer_load_dwordx4 %0, %8, %11, %12 offen" ILQR_TILE_NT "\n\t"                     \
        "buffer_load_dwordx4 %1, %8, %11, %12 offen offset:16" ILQR_TILE_NT "\n\t"           \
        "buffer_load_dwordx4 %2, %9, %11, %12 offen" ILQR_TILE_NT "\n\t"                     \
        "buffer_load_dwordx4 %3, %9, %11, %12 offen offset:16" ILQR_TILE_NT "\n\t"           \
        "buffer_load_dwordx4 %4, %9, %11, %12 offen offset:256" ILQR_TILE_NT "\n\t"          \
        "buffer_load_dwordx4 %5, %9, %11, %12 offen offset:272" ILQR_TILE_NT "\n\t"          \
        "buffer_load_dwordx2 %6, %10, %11, %12 offen offset:128" ILQR_TILE_NT "\n\t"         \
        "buffer_load_dwordx2 %7, %8, %11, %12 offen offset:256" ILQR_TILE_NT                 \
        : OUT(ski0), OUT(ski1), OUT(skj0), OUT(skj1), OUT(vj0), OUT(vj1), OUT(lxx), OUT(bi)  \
        : "v"(o.vi), "v"(o.vj), "v"(o.vl), "s"(srd), "s"(soff)                               \
        : "memory")
    template <bool FIRST> ILQR_DEV void issue(const i32x4& srd, const TileOffsets& o, int soff) {
        if constexpr (FIRST) ILQR_RAWTILE_F64(ILQR_OUT_FIRST);
        else ILQR_RAWTILE_F64(ILQR_OUT_REFILL);
    }
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(ski0), "+v"(ski1), "+v"(skj0), "+v"(skj1), "+v"(vj0), "+v"(vj1), "+v"(lxx), "+v"(bi)
                     : "i"(N) : "memory");
    }
    ILQR_DEV void unpack(Tile16<double>& t) const {
        t.ski[0] = ski0.x; t.ski[1] = ski0.y; t.ski[2] = ski1.x; t.ski[3] = ski1.y;
        t.skj[0] = skj0.x; t.skj[1] = skj0.y; t.skj[2] = skj1.x; t.skj[3] = skj1.y;
        t.vj[0] = vj0.x; t.vj[1] = vj0.y; t.vj[2] = vj1.x; t.vj[3] = vj1.y;
        t.lxx = lxx; t.bi = bi; t.luxi = 0.0;   // (row-form l_ux is only used by the fp32 step)
    }
};

// The quad-shared fp32 tile (TileQ) in the ring: two 16-byte and three 4-byte loads per lane.  issue<true>() is the
// prologue's combined form; issue_part<K>() is ONE load of a refill as its own statement, for the step that spreads the
// refill of the previous slot over its own arithmetic (tile16_step_f32<true>).  Part 0 carries the s_nop 4 of the
// combined form: all five parts read the same descriptor / soffset SGPRs, which are therefore materialised before it.
struct RawTileQ {
    static constexpr int NLOAD = 5;
    f32x4n skj, vj;
    float a, lxx, c;
    ILQR_DEV void issue_first(const i32x4& srd, const TileOffsets& o, int soff) {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %5, %7, %8 offen" ILQR_TILE_NT "\n\t"
            "buffer_load_dwordx4 %1, %5, %7, %8 offen offset:128" ILQR_TILE_NT "\n\t"
            "buffer_load_dword %2, %6, %7, %8 offen" ILQR_TILE_NT "\n\t"
            "buffer_load_dword %3, %6, %7, %8 offen offset:64" ILQR_TILE_NT "\n\t"
            "buffer_load_dword %4, %6, %7, %8 offen offset:128" ILQR_TILE_NT
            : "=&v"(skj), "=&v"(vj), "=&v"(a), "=&v"(lxx), "=&v"(c)
            : "v"(o.vj), "v"(o.vl), "s"(srd), "s"(soff)
            : "memory");
    }
#define ILQR_RAWTILE_PART(NOP, dst, voff, IMM)                                                               \
    asm volatile(NOP "buffer_load_dword" IMM ILQR_TILE_NT : "+&v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory")
    template <int K> ILQR_DEV void issue_part(const i32x4& srd, const TileOffsets& o, int soff) {
        if constexpr (K == 0) ILQR_RAWTILE_PART("s_nop 4\n\t", skj, o.vj, "x4 %0, %1, %2, %3 offen");
        if constexpr (K == 1) ILQR_RAWTILE_PART("", vj, o.vj, "x4 %0, %1, %2, %3 offen offset:128");
        if constexpr (K == 2) ILQR_RAWTILE_PART("", a, o.vl, " %0, %1, %2, %3 offen");
        if constexpr (K == 3) ILQR_RAWTILE_PART("", lxx, o.vl, " %0, %1, %2, %3 offen offset:64");
        if constexpr (K == 4) ILQR_RAWTILE_PART("", c, o.vl, " %0, %1, %2, %3 offen offset:128");
    }
    template <int N> ILQR_DEV void wait() {
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(skj), "+v"(vj), "+v"(a), "+v"(lxx), "+v"(c) : "i"(N) : "memory");
    }
    ILQR_DEV void unpack(TileQ& t) const {
        t.skj[0] = skj.x; t.skj[1] = skj.y; t.skj[2] = skj.z; t.skj[3] = skj.w;
        t.vj[0] = vj.x; t.vj[1] = vj.y; t.vj[2] = vj.z; t.vj[3] = vj.w;
        t.a = a; t.lxx = lxx; t.c = c;
    }
};

// acc += coef * (w moved across lanes by a DPP pattern), as ONE instruction (v_fmac_f32_dpp).  hipcc's
// DPP combiner folds a lane move into v_mul / v_add but not into the accumulating v_fmac (tied operand), so
// each contraction term cost a v_mov_b32_dpp plus a v_fmac; the fused form is written in asm.  The asm is
// opaque to hipcc's hazard padding (cdna_hip_programming.md 5.7 item 2): a VGPR written by a VALU needs 2
// wait states before a DPP read, hence the s_nop 1 inside every statement (it replaces the s_nop hipcc
// put in front of the v_mov_b32_dpp anyway).
#define ILQR_FMAC_DPP_(NOP, acc, w, coef, CTRL)                                                        \
    asm volatile(NOP "v_fmac_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"        \
                 : "+v"(acc)                                                                           \
                 : "v"(w), "v"(coef))
// first use of a freshly written operand carries the 2 wait states; the next statements on the SAME operand
// follow it in program order (volatile asm statements are not reordered among themselves) and need none
#define ILQR_FMAC_DPP_FIRST(acc, w, coef, CTRL) ILQR_FMAC_DPP_("s_nop 1\n\t", acc, w, coef, CTRL)
#define ILQR_FMAC_DPP_NEXT(acc, w, coef, CTRL) ILQR_FMAC_DPP_("", acc, w, coef, CTRL)

// contraction along the matrix row: sum_d skj[d] * w[(j + d) % 4]
ILQR_DEV float contract_row(const float* skj, float w, float init) {
    float acc = fmaf(skj[0], w, init);
    ILQR_FMAC_DPP_FIRST(acc, w, skj[1], "quad_perm:[1,2,3,0]");
    ILQR_FMAC_DPP_NEXT(acc, w, skj[2], "quad_perm:[2,3,0,1]");
    ILQR_FMAC_DPP_NEXT(acc, w, skj[3], "quad_perm:[3,0,1,2]");
    return acc;
}
ILQR_DEV double contract_row(const double* skj, double w, double init) {
    double acc = fma(skj[0], w, init);
    acc += skj[1] * dpp<kRight1>(w);
    acc += skj[2] * dpp<kRight2>(w);
    acc += skj[3] * dpp<kRight3>(w);
    return acc;
}
// contraction down the matrix column: sum_d ski[d] * w[(i + d) % 4]
ILQR_DEV float contract_col(const float* ski, float w) {
    float acc = ski[0] * w;
    ILQR_FMAC_DPP_FIRST(acc, w, ski[1], "row_ror:12");
    ILQR_FMAC_DPP_NEXT(acc, w, ski[2], "row_ror:8");
    ILQR_FMAC_DPP_NEXT(acc, w, ski[3], "row_ror:4");
    return acc;
}
ILQR_DEV double contract_col(const double* ski, double w) {
    double acc = ski[0] * w;
    acc += ski[1] * dpp<kDown1>(w);
    acc += ski[2] * dpp<kDown2>(w);
    acc += ski[3] * dpp<kDown3>(w);
    return acc;
}
// sum over the 4 lanes of a quad, result in every lane of the quad
template <typename T> ILQR_DEV T quad_sum(T v) {
    v += dpp<kSwap1>(v);
    v += dpp<kRight2>(v);
    return v;
}

// lane (i, j) <- lane (j, i) of the same 16-lane row: the 4x4 transpose is neither quad-local nor a
// row rotation, so it goes through the LDS crossbar (ds_bpermute: one instruction, no LDS memory).
ILQR_DEV float lane_transpose(float v, int src_byte) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_byte, __float_as_int(v)));
}
ILQR_DEV double lane_transpose(double v, int src_byte) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_byte, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_byte, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the two row contractions that share coefficients and rotation pattern (Q_ux from pu, Q_x from V_x).
// (A packed v_pk_fma_f32 form of this pair was tried: hipcc assembled the register pairs through scratch
// and the sweep ran 2.7x slower, so the two streams stay scalar.)
ILQR_DEV void contract_row2(const float* skj, float w0, float w1, float c0, float c1, float& o0, float& o1) {
    float a0 = fmaf(skj[0], w0, c0), a1 = fmaf(skj[0], w1, c1);
    ILQR_FMAC_DPP_FIRST(a0, w0, skj[1], "quad_perm:[1,2,3,0]");
    ILQR_FMAC_DPP_NEXT(a1, w1, skj[1], "quad_perm:[1,2,3,0]");
    ILQR_FMAC_DPP_NEXT(a0, w0, skj[2], "quad_perm:[2,3,0,1]");
    ILQR_FMAC_DPP_NEXT(a1, w1, skj[2], "quad_perm:[2,3,0,1]");
    ILQR_FMAC_DPP_NEXT(a0, w0, skj[3], "quad_perm:[3,0,1,2]");
    ILQR_FMAC_DPP_NEXT(a1, w1, skj[3], "quad_perm:[3,0,1,2]");
    o0 = a0; o1 = a1;
}
ILQR_DEV void contract_row2(const double* skj, double w0, double w1, double c0, double c1, double& o0, double& o1) {
    o0 = contract_row(skj, w0, c0);
    o1 = contract_row(skj, w1, c1);
}

// per-lane constants of the sweep
template <typename T> struct LaneConst {
    T m0, m1;      // 1 where j == 0 / j == 1, else 0: fold l_u / l_uu (tile slots e_0, e_1) into the quad sums
    int tr_byte;   // 4 * (lane of the transposed element)
};

// One Riccati step (iLQR_class.py:100-114) on the lane-distributed state:
//   V  = V_xx[i][j] at lane (i, j);  vx = V_x[j] ("column form": replicated down the rows).
// Returns K[j] (column form) and k (replicated); pd = Q_uu (+mu) > 0.
template <typename T, bool REG>
ILQR_DEV void tile16_step(const Tile16<T>& c, const LaneConst<T>& lc, T mu, T& V, T& vx, T& Kj, T& kff, bool& pd) {
    // P = f_x' V_xx :  P[i][j] = sum_d A[(i+d)%4][i] * V[(i+d)%4][j]
    const T P = contract_col(c.ski, V);
    // pu = f_u' V_xx :  pu[j] = sum_i b[i] V[i][j]   (sum down the rows, result in every row)
    T pu = c.bi * V;
    pu += dpp<kDown2>(pu);
    pu += dpp<kDown1>(pu);
    // Q_ux = l_ux + pu f_x ; Q_x = l_x + f_x' V_x  (same coefficients, same rotations: one packed stream)
    T Qux, Qx;
    contract_row2(c.skj, pu, vx, c.vj[2], c.vj[1], Qux, Qx);
    // Q_uu = l_uu + pu f_u ; Q_u = l_u + f_u' V_x   (l_uu, l_u enter the quad sums on lanes j = 1, j = 0)
    const T Quu = quad_sum(lc.m1 * c.vj[3] + pu * c.vj[0]);
    const T Qu = quad_sum(lc.m0 * c.vj[3] + c.vj[0] * vx);
    const T Qr = REG ? Quu + mu : Quu;
    pd = Qr > T(0);
    const T inv = fast_rcp(Qr);
    Kj = -(Qux * inv);   // K = -Q_uu^-1 Q_ux   (:109)
    kff = -(Qu * inv);   // k = -Q_uu^-1 Q_u    (:110)
    // Q_ux in "row form": lane (i, j) <- Q_ux[i] = the column form's lane i of quad i, broadcast inside each quad by four
    // masked DPP moves (bank_mask: one quad of every 16-lane row per move).  Until round 3 a ds_bpermute from lane (j, i)
    // -- the 4 x 4 transpose through the LDS crossbar -- whose ~68 cycles sat on the step's serial path.
    T Quxi = Qux;
    Quxi = dpp_row<dpp_quad(0, 0, 0, 0), 0x1>(Quxi, Qux);
    Quxi = dpp_row<dpp_quad(1, 1, 1, 1), 0x2>(Quxi, Qux);
    Quxi = dpp_row<dpp_quad(2, 2, 2, 2), 0x4>(Quxi, Qux);
    Quxi = dpp_row<dpp_quad(3, 3, 3, 3), 0x8>(Quxi, Qux);
    // Q_xx = l_xx + P f_x
    const T Qxx = contract_row(c.skj, P, c.lxx);
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

// ---- fp32 step with a hand-ordered instruction stream ---------------------------------------------------
// The chains of a step -- (A) P -> Q_xx, (B) pu -> Q_ux, (C) Q_x, (C2) Q_u,
// (Q) Q_uu -> 1/Q_uu -- are independent until the gain.  A lone wave (the batch puts one wave on each SIMD) issues
// an independent VALU instruction every ~5.4 cycles but a dependent one only every ~8.8, a dependent DPP read
// needs two more wait states, and v_rcp_f32 costs 9 (12.5 dependent) -- tools/micro/issue_rate.hip.  hipcc's
// order left the sweep at ~6.5 cycles/instruction; here the 33 instructions up to the reciprocal are volatile asm
// statements (never reordered among themselves) written round-robin over the chains, so that every operand was
// produced >= 3 instructions earlier: no s_nop, no dependent-issue bubble.  Q_uu is finished EARLY (27) and its
// reciprocal issued at 31, so the transcendental's latency hides under the last four contraction terms instead of
// heading the serial tail; the tail left to hipcc is K, k, V_xx, V_x: two dependent instructions deep.
//
// Q_ux in ROW form (lane (i, j) holds Q_ux[i], for the value update) is taken from the column form -- lane i of quad i --
// by four masked DPP quad broadcasts (bank_mask: one quad of every 16-lane row per move) that sit in the slots where the
// tail used to wait for 1 / Q_uu.  (Rounds 1-2 computed it a second time as a contraction chain through V_xx's symmetry,
// 8 instructions instead of 4 and an ulp-sized deviation from the transposed value; before that a ds_bpermute + lgkmcnt
// wait sat on the critical path: ~68 cycles.)
//
// 1/Q_uu keeps its Newton step (without it the fp32 solve parts from the fp32 oracle's alpha sequence at iterations
// where the cost still moves by 2e-3, against 2e-4 with it -- tests/test_gpu_fullshape.py), but as instructions 34
// and 37 of the block, not as two dependent instructions of the tail.
//
// The statement is cut in five, and with REFILL one load of the PREVIOUS slot's refill sits in each cut (that slot
// was consumed one step earlier: its registers are dead).  A vector-memory instruction of a lone wave is not free
// -- the wave issues in order and a 16-byte-per-lane load holds it ~16 cycles whether or not ALU work follows
// (issue_rate: 8 loads + 56 FMAs = 8 x 54 cycles, the sum of both) -- so spreading the loads does not hide their
// issue; it lets each load's address phase start earlier and measured 0.3-0.8 us (1-2 %) better than issuing the
// five back to back behind the step.
#define ILQR_QP1 "quad_perm:[1,2,3,0]"
#define ILQR_QP2 "quad_perm:[2,3,0,1]"
#define ILQR_QP3 "quad_perm:[3,0,1,2]"
#define ILQR_QSW "quad_perm:[1,0,3,2]"
#define ILQR_QB0 "quad_perm:[0,0,0,0]"
#define ILQR_QB1 "quad_perm:[1,1,1,1]"
#define ILQR_QB2 "quad_perm:[2,2,2,2]"
#define ILQR_QB3 "quad_perm:[3,3,3,3]"
template <bool REFILL>
ILQR_DEV void tile16_step_f32(const TileQ& c, const LaneConst<float>& lc, float& V, float& vx, float& Kj, float& kff,
                              bool& pd, RawTileQ& prev, const i32x4& srd, const TileOffsets& off, int soff) {
    float P, pu, qx, qu, quu, Qxx, Qux, Quxi, t, t2, inv, si1, si2, si3;
#define DPPT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPPE " row_mask:0xf bank_mask:0xf bound_ctrl:1"
    // SK[i][0] and f_u[i] reach their (single-move) consumers 1, 2 and 17 as DPP broadcasts of a / c; SK[i][1..3] and
    // l_ux[i] are broadcast into registers first -- four instructions that depend on the tile only, not on V_xx / V_x,
    // and so issue while the previous step's tail is still in flight.
    asm volatile(
        "v_mov_b32_dpp %[si1], %[a] " ILQR_QB1 DPPT                   //  a  SK[i][1]
        "v_mov_b32_dpp %[si2], %[a] " ILQR_QB2 DPPT                   //  b  SK[i][2]
        "v_mov_b32_dpp %[si3], %[a] " ILQR_QB3 DPPT                   //  c  SK[i][3]
        "v_mul_f32_dpp %[P], %[a], %[V] " ILQR_QB0 DPPT               //  1 A   SK[i][0] * V
        "v_mul_f32_dpp %[pu], %[c], %[V] " ILQR_QB0 DPPT              //  2 B   f_u[i] * V
        "v_fma_f32 %[qx], %[sj0], %[vx], %[vj1]\n\t"                //  4 C   Q_x = l_x + ...
        "v_fmac_f32_dpp %[P], %[V], %[si1] row_ror:12" DPPT           //  5 A
        "v_add_f32_dpp %[t], %[pu], %[pu] row_ror:8" DPPE             //  6 B   pu[i] + pu[i+2]
        : [P] "=&v"(P), [pu] "=&v"(pu), [qx] "=&v"(qx), [t] "=&v"(t), [si1] "=&v"(si1), [si2] "=&v"(si2), [si3] "=&v"(si3)
        : [V] "v"(V), [vx] "v"(vx), [a] "v"(c.a), [c] "v"(c.c), [sj0] "v"(c.skj[0]), [vj1] "v"(c.vj[1]));
    if constexpr (REFILL) prev.template issue_part<0>(srd, off, soff);
    asm volatile(
        "v_fmac_f32_dpp %[qx], %[vx], %[sj1] " ILQR_QP1 DPPT          //  8 C
        "v_fmac_f32_dpp %[P], %[V], %[si2] row_ror:8" DPPT            //  9 A
        "v_add_f32_dpp %[pu], %[t], %[t] row_ror:12" DPPT             // 10 B   pu done (down the rows; column form)
        "v_mul_f32 %[qu], %[vj0], %[vx]\n\t"                        // 12 C2  f_u[j] * V_x[j]
        "v_fmac_f32_dpp %[qx], %[vx], %[sj2] " ILQR_QP2 DPPT          // 13 C
        "v_fmac_f32_dpp %[P], %[V], %[si3] row_ror:4" DPPE            // 14 A   P done
        : [qx] "+v"(qx), [P] "+v"(P), [qu] "=&v"(qu), [pu] "=&v"(pu)
        : [V] "v"(V), [vx] "v"(vx), [si2] "v"(si2), [si3] "v"(si3), [sj1] "v"(c.skj[1]), [sj2] "v"(c.skj[2]),
          [vj0] "v"(c.vj[0]), [t] "v"(t));
    if constexpr (REFILL) prev.template issue_part<1>(srd, off, soff);
    asm volatile(
        "v_mul_f32 %[quu], %[pu], %[vj0]\n\t"                       // 15 Q   pu[j] * f_u[j]
        "v_fma_f32 %[Qux], %[sj0], %[pu], %[vj2]\n\t"               // 16 B   Q_ux[j] = l_ux[j] + ...
        "v_fmac_f32_dpp %[qx], %[vx], %[sj3] " ILQR_QP3 DPPT          // 18 C   Q_x done
        "v_fmac_f32 %[quu], %[m1], %[vj3]\n\t"                      // 19 Q   + l_uu on lane j = 1
        "v_fmac_f32 %[qu], %[m0], %[vj3]\n\t"                       // 20 C2  + l_u on lane j = 0
        "v_fma_f32 %[Qxx], %[sj0], %[P], %[lxx]"                      // 21 A   Q_xx = l_xx + ...
        : [quu] "=&v"(quu), [Qux] "=&v"(Qux), [Qxx] "=&v"(Qxx), [qx] "+v"(qx), [qu] "+v"(qu)
        : [vx] "v"(vx), [sj0] "v"(c.skj[0]), [sj3] "v"(c.skj[3]), [vj0] "v"(c.vj[0]), [vj2] "v"(c.vj[2]),
          [vj3] "v"(c.vj[3]), [lxx] "v"(c.lxx), [m0] "v"(lc.m0), [m1] "v"(lc.m1), [pu] "v"(pu), [P] "v"(P));
    if constexpr (REFILL) prev.template issue_part<2>(srd, off, soff);
    asm volatile(
        "v_fmac_f32_dpp %[Qux], %[pu], %[sj1] " ILQR_QP1 DPPT         // 22 B
        "v_add_f32_dpp %[t2], %[quu], %[quu] " ILQR_QSW DPPT          // 23 Q
        "v_add_f32_dpp %[t], %[qu], %[qu] " ILQR_QSW DPPT             // 25 C2
        "v_fmac_f32_dpp %[Qxx], %[P], %[sj1] " ILQR_QP1 DPPT          // 26 A
        "v_add_f32_dpp %[quu], %[t2], %[t2] " ILQR_QP2 DPPT           // 27 Q   Q_uu done
        "v_fmac_f32_dpp %[Qux], %[pu], %[sj2] " ILQR_QP2 DPPE         // 28 B
        : [Qux] "+v"(Qux), [Qxx] "+v"(Qxx), [quu] "+v"(quu), [t] "=&v"(t), [t2] "=&v"(t2)
        : [sj1] "v"(c.skj[1]), [sj2] "v"(c.skj[2]), [pu] "v"(pu), [P] "v"(P), [qu] "v"(qu));
    if constexpr (REFILL) prev.template issue_part<3>(srd, off, soff);
    asm volatile(
        "v_add_f32_dpp %[qu], %[t], %[t] " ILQR_QP2 DPPT              // 30 C2  Q_u done
        "v_rcp_f32 %[inv], %[quu]\n\t"                              // 31 Q   r ~ 1 / Q_uu
        "v_fmac_f32_dpp %[Qxx], %[P], %[sj2] " ILQR_QP2 DPPT          // 32 A
        "v_fmac_f32_dpp %[Qux], %[pu], %[sj3] " ILQR_QP3 DPPT         // 33 B   Q_ux (column form) done
        "v_fma_f32 %[e], -%[quu], %[inv], 1.0\n\t"                  // 34 Q   e = 1 - Q_uu r
        "v_fmac_f32_dpp %[Qxx], %[P], %[sj3] " ILQR_QP3 DPPT          // 36 A   Q_xx done
        // R: Q_ux in ROW form (lane (i, j) <- Q_ux[i]) = lane i of quad i of the column form, broadcast inside each quad by
        // four masked moves (bank_mask: one quad of every row per move); they sit where the tail used to wait for 1/Q_uu
        "v_mov_b32_dpp %[Quxi], %[Qux] " ILQR_QB0 " row_mask:0xf bank_mask:0x1\n\t"
        "v_fmac_f32 %[inv], %[e], %[inv]\n\t"                         // 37 Q   r += e r (one Newton step)
        "v_mov_b32_dpp %[Quxi], %[Qux] " ILQR_QB1 " row_mask:0xf bank_mask:0x2\n\t"
        "v_mov_b32_dpp %[Quxi], %[Qux] " ILQR_QB2 " row_mask:0xf bank_mask:0x4\n\t"
        "v_mov_b32_dpp %[Quxi], %[Qux] " ILQR_QB3 " row_mask:0xf bank_mask:0x8"
        : [Quxi] "=&v"(Quxi), [Qxx] "+v"(Qxx), [Qux] "+v"(Qux), [qu] "=&v"(qu), [inv] "=&v"(inv), [e] "=&v"(t2)
        : [sj2] "v"(c.skj[2]), [sj3] "v"(c.skj[3]), [pu] "v"(pu), [P] "v"(P), [t] "v"(t), [quu] "v"(quu));
    if constexpr (REFILL) prev.template issue_part<4>(srd, off, soff);
#undef DPPT
#undef DPPE
    pd = quu > 0.0f;
    Kj = -(Qux * inv);
    kff = -(qu * inv);
    V = fmaf(Quxi, Kj, Qxx);
    vx = fmaf(Kj, qu, qx);
}

// Workgroup = 4 waves = 16 trajectories, launched with > 80 KiB of (unused) dynamic LDS so that a CU
// never hosts two workgroups: the 256 workgroups of a 4096-trajectory batch then sit one per CU and
// their four waves one per SIMD.  (With 64-thread workgroups the dispatcher doubled waves up on ~8 % of
// the SIMDs and left as many empty; an issue-bound wave that shares its SIMD runs ~1.35x longer and
// the slowest wave is the kernel's duration.)
constexpr int kTile16PinLds = 84 * 1024;

// NXA = the system's real n_x (2..4): smaller systems ride the same 4 x 4 tile, zero-padded by linearize_kernel
// (padding states have no dynamics and no cost: their gains come out exactly zero and nothing else changes);
// only the gain record (gain_record(NXA, 1) scalars: K[0..NXA), k) knows the difference.
template <typename T, bool REG, int NXA>
__global__ void __launch_bounds__(256) backward_tile16_kernel(KArgs<T> a) {
    // tiles in flight per lane: (D-1) * (loads per tile + 1 store) must stay <= 63 (the vmcnt field)
    // SPREAD (the fp32 step without regularisation): the refill of a slot is issued one load at a time inside the NEXT
    // step's arithmetic (tile16_step_f32<true>) instead of back to back behind its own step
#ifndef ILQR_TILE16_D32
#define ILQR_TILE16_D32 10
#endif
    constexpr bool SPREAD = sizeof(T) == 4 && !REG;
    constexpr int D = SPREAD ? ILQR_TILE16_D32 : sizeof(T) == 4 ? 10 : 7;
    constexpr int R = gain_record(NXA, 1);     // 8 for n_x = 4
    const int lane = threadIdx.x & 63;
    const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3;
    const int gidx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool valid = gidx < a.B;
    const int b = valid ? gidx : a.B - 1;  // out-of-range groups shadow the last trajectory, never store
    const int st = a.status[b];
    const bool act = valid && traj_active(st);
    if (__ballot(act) == 0ull) return;
    if (a.reset_slots && act && l16 == 0) a.cur_slot[b] = 0;   // see linearize_kernel: the trajectory now lives in slot 0
    ClockProbe cp;
    cp.start();
    const size_t B = a.B;
    const int N = a.N;
    T V = a.term[(size_t)(4 + l16) * B + b];
    T vx = a.term[(size_t)j * B + b];
    // buffer descriptors over the whole expansion / gain tensors (host guarantees both are < 2 GiB,
    // otherwise it launches the flat-addressed LDS-ring variant)
    const unsigned lin_bytes = (unsigned)((size_t)N * B * kTile16 * sizeof(T));
    const unsigned gain_bytes = (unsigned)((size_t)N * B * R * sizeof(T));
    const __amdgpu_buffer_rsrc_t rlin = make_rsrc(a.lin, lin_bytes);
    const __amdgpu_buffer_rsrc_t rgain = make_rsrc(a.gains, gain_bytes);
    const int tstride = (int)(B * kTile16 * sizeof(T));  // bytes between consecutive time steps
    const int rstride = (int)(B * R * sizeof(T));
    TileOffsets off;
    off.vi = (int)((b * kTile16 + 4 * i) * sizeof(T));
    off.vj = (int)((b * kTile16 + 4 * j) * sizeof(T));
    off.vl = (int)((b * kTile16 + l16) * sizeof(T));
    // lanes (0, j) store K[j], lane (1, 0) stores k.  fp32: every other lane's offset lies beyond the descriptor's
    // range, where the hardware drops the store -- no exec-mask juggling around the one counted store of a step.
    // (Measured rule, tools/micro/range_probe.hip: the check is per dword against num_records, for 32-, 64- and 128-bit
    // accesses alike.  The wrong fp64 gains of round 1 were not the store: hipcc had copied a ring register whose load
    // was still in flight -- see RawTile.  fp64 keeps the exec predicate because it measured 8 % faster there.)
    constexpr bool DROP = sizeof(T) == 4 || ILQR_DROP_ALL;
    const bool storer = act && ((i == 0 && j < NXA) || l16 == 4);
    const int rec_off = (storer || !DROP) ? (int)((b * R + (i == 0 ? j : NXA)) * sizeof(T)) : 0x7ffffff0;
    bool all_pd = true;
    LaneConst<T> lc;
    lc.m0 = T(j == 0);
    lc.m1 = T(j == 1);
    lc.tr_byte = 4 * ((lane & 48) | (j << 2) | i);

    int t = N - 1;
    if constexpr (SPREAD) {
        const i32x4 srd = make_srd(a.lin, lin_bytes);
        // running byte offsets (one s_sub each per step): gain record of the step, tile of its refill
        int goff = t * rstride, roff = 0;
        auto step = [&](auto refill, const TileQ& c, RawTileQ& prev) {
            T Kj, kff;
            bool pd;
#ifdef ILQR_T16_NOLOAD   // timing experiments only (wrong results): what the step costs without its refill / its store
            tile16_step_f32<false>(c, lc, V, vx, Kj, kff, pd, prev, srd, off, uniform(roff > 0 ? roff : 0));
#else
            tile16_step_f32<decltype(refill)::value>(c, lc, V, vx, Kj, kff, pd, prev, srd, off,
                                                     uniform(roff > 0 ? roff : 0));
#endif
            all_pd = all_pd && pd;
#ifndef ILQR_T16_NOSTORE
            buf_store1(rgain, rec_off, uniform(goff), (i == 0) ? Kj : kff);
#else
            if (a.N == 123457) buf_store1(rgain, rec_off, uniform(goff), (i == 0) ? Kj : kff);
#endif
            goff -= rstride;
            roff -= tstride;
        };
        RawTileQ ring[D];
#ifdef ILQR_T16_STAMPS
        long long stamp_cal = 0, stamp_wait = 0;
        const long long stamp_t0 = __builtin_readcyclecounter();
#endif
        // remainder steps first (no ring), so that the pipelined loop runs whole rings only
        for (int r = N % D; r > 0; --r, --t) {
            TileQ c;
            tileq_load_buf(c, rlin, off, uniform(t * tstride));
            step(std::false_type{}, c, ring[0]);
        }
        if (t >= 0) {
            // D tiles per lane in flight, loaded by asm and waited for with self-counted vmcnt (see RawTile)
            constexpr int NL = RawTileQ::NLOAD;
#pragma unroll
            for (int u = 0; u < D; ++u) ring[u].issue_first(srd, off, uniform((t - u) * tstride));
            // Step u consumes slot u and, between its own instructions, refills slot u-1 (consumed one step earlier)
            // with the tile D steps below that one.  Memory operations in issue order: prologue D*NL loads; first
            // pass: step 0 = 1 store, step u >= 1 = NL loads + 1 store; later passes: every step NL loads + 1 store.
            // "Slot u has landed" therefore allows
            //   first pass:  u == 0: (D-1)*NL              (the younger prologue loads)
            //                u >= 1: (D-1-u)*NL + u + (u-1)*NL = (D-2)*NL + u
            //   afterwards:  1 + (D-2)*(NL+1)              (its refill sat in step u+1 of the previous pass: that
            //                                               step's store, then D-2 whole steps)
            // younger operations outstanding.
            static_assert(1 + (D - 2) * (NL + 1) <= 63 && (D - 1) * NL <= 63, "vmcnt field");
            roff = (t + 1 - D) * tstride;   // step u's refill is tile t-u+1-D
            static_for<D>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                ring[u].template wait<(u == 0) ? (D - 1) * NL : (D - 2) * NL + u>();
                TileQ c;
                ring[u].unpack(c);
                if constexpr (u == 0) step(std::false_type{}, c, ring[0]);   // (roff moves on: step 1 refills tile t-D)
                else step(std::true_type{}, c, ring[u - 1]);
            });
            for (t -= D; t >= 0; t -= D) {
                static_for<D>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
#ifdef ILQR_T16_STAMPS   // diagnostic build (tools/t16_stamps.py): cycles spent in the ring's wait, and an empty pair
                    const long long w0 = __builtin_readcyclecounter();
                    const long long w1 = __builtin_readcyclecounter();
                    ring[u].template wait<1 + (D - 2) * (NL + 1)>();
                    const long long w2 = __builtin_readcyclecounter();
                    stamp_cal += w1 - w0;
                    stamp_wait += w2 - w1;
#else
                    ring[u].template wait<1 + (D - 2) * (NL + 1)>();
#endif
                    TileQ c;
                    ring[u].unpack(c);
                    // slot u-1 (slot D-1 of the previous pass for u == 0) was consumed at step t-u+1; clamped to
                    // tile 0 at the end of the sweep, surplus loads drained below
                    step(std::true_type{}, c, ring[(u + D - 1) % D]);
                });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef ILQR_T16_STAMPS
        if (a.probe && blockIdx.x == 7 && threadIdx.x == 0) {
            a.probe[2] = stamp_wait; a.probe[3] = stamp_cal; a.probe[4] = __builtin_readcyclecounter() - stamp_t0;   // (slots no ClockProbe uses for the sweep)
        }
#endif
    } else {
        auto do_step = [&](const Tile16<T>& c, int t) {
            T Kj, kff;
            bool pd;
            tile16_step<T, REG>(c, lc, a.mu, V, vx, Kj, kff, pd);
            all_pd = all_pd && pd;
            if (DROP || storer) buf_store1(rgain, rec_off, uniform(t * rstride), (i == 0) ? Kj : kff);
        };
        // remainder steps first (no ring), so that the pipelined loop runs whole rings only
        for (int r = N % D; r > 0; --r, --t) {
            Tile16<T> c;
            tile16_load_buf(c, rlin, off, uniform(t * tstride));
            do_step(c, t);
        }
        if (t >= 0) {
            // D tiles per lane in flight, loaded by asm and waited for with self-counted vmcnt (see RawTile)
            constexpr int NL = RawTile<T>::NLOAD;
            const i32x4 srd = make_srd(a.lin, lin_bytes);
            RawTile<T> ring[D];
#pragma unroll
            for (int u = 0; u < D; ++u) ring[u].template issue<true>(srd, off, uniform((t - u) * tstride));
            // first ring pass: only the prologue's loads (plus this pass's own stores / refills) are in flight
#pragma unroll
            for (int u = 0; u < D; ++u) {
                ring[u].template wait<(D - 1) * NL>();
                Tile16<T> c;
                ring[u].unpack(c);
                do_step(c, t - u);
                const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                ring[u].template issue<false>(srd, off, uniform(tn * tstride));
            }
            for (t -= D; t >= 0; t -= D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    // slot u was refilled D steps ago; since then (D-1) steps issued 1 store + NL loads each
                    ring[u].template wait<(D - 1) * (NL + 1)>();
                    Tile16<T> c;
                    ring[u].unpack(c);
                    do_step(c, t - u);
                    // refill the SAME registers with the tile D steps ahead (clamped to tile 0 at the end of the
                    // sweep: the body stays branch-free; the surplus loads are drained before the kernel ends)
                    const int tn = (t - u - D) > 0 ? (t - u - D) : 0;
                    ring[u].template issue<false>(srd, off, uniform(tn * tstride));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    cp.stop(a.probe, 0);
    if (act && l16 == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

// ---------------------------------------------------------------------------------------------
// LDS-ring variant.  The register ring above leaves the s_waitcnt placement to hipcc, which drains
// the whole ring at every loop edge (one exposed HBM latency per D steps).  Here the wave's four
// tiles of one step -- 4 x 192 B (f32) contiguous in HBM because the four trajectories of a wave are
// neighbours in b -- are copied HBM -> LDS by ONE LDS-DMA instruction (global_load_lds_dwordx4,
// 48 lanes x 16 B; two for f64), RING steps ahead, with no VGPR destination.  The DMA and its wait
// are inline asm, so the kernel counts vmcnt itself: each step issues exactly one store and NDMA
// DMAs, hence "tile t has landed" == at most (RING-1)*(NDMA+1) younger operations outstanding
// (loads, stores and LDS-DMA retire in issue order).  Tiles are then read LDS -> VGPR with
// ds_read_b128/b32 (compiler-counted lgkmcnt), one step ahead of their use.
// ---------------------------------------------------------------------------------------------
template <typename T> struct LdsRing;
template <> struct LdsRing<float> { static constexpr int RING = 16, NDMA = 1; };
template <> struct LdsRing<double> { static constexpr int RING = 12, NDMA = 2; };

// one LDS-DMA piece: every active lane copies 16 B from its own global address to
// LDS[m0_base + lane*16].  M0 is compiler-reserved: written and restored inside the statement.
ILQR_DEV void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
template <int N> ILQR_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <typename T> ILQR_DEV void tile16_load_lds(Tile16<T>& tl, const T* tp, int i, int j, int l16) {
    // tp: this trajectory's tile inside the LDS slot (generic pointer to __shared__)
    using V4 = typename Vec4<T>::type;
    const V4 a = *reinterpret_cast<const V4*>(tp + 4 * i);
    const V4 c = *reinterpret_cast<const V4*>(tp + 4 * j);
    const V4 v = *reinterpret_cast<const V4*>(tp + 32 + 4 * j);
    tl.ski[0] = a.x; tl.ski[1] = a.y; tl.ski[2] = a.z; tl.ski[3] = a.w;
    tl.skj[0] = c.x; tl.skj[1] = c.y; tl.skj[2] = c.z; tl.skj[3] = c.w;
    tl.vj[0] = v.x; tl.vj[1] = v.y; tl.vj[2] = v.z; tl.vj[3] = v.w;
    tl.lxx = tp[16 + l16];
    tl.bi = tp[32 + 4 * i];
    tl.luxi = tp[32 + 4 * i + 2];
}

template <typename T, bool REG, int NXA>
__global__ void __launch_bounds__(64) backward_tile16_lds_kernel(KArgs<T> a) {
    constexpr int RING = LdsRing<T>::RING, NDMA = LdsRing<T>::NDMA;
    constexpr int R = gain_record(NXA, 1);
    constexpr int SLOT = 4 * kTile16;                    // scalars per ring slot (4 trajectories)
    constexpr int SLOT_BYTES = SLOT * (int)sizeof(T);    // 768 / 1536
    constexpr int PIECES = SLOT_BYTES / 16;              // 16-B pieces per slot: 48 / 96
    __shared__ __attribute__((aligned(16))) T ring[RING * SLOT];
    const int lane = threadIdx.x;
    const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3, grp = lane >> 4;
    const int gidx = blockIdx.x * 4 + grp;
    const bool valid = gidx < a.B;
    const int b = valid ? gidx : a.B - 1;
    const int st = a.status[b];
    const bool act = valid && traj_active(st);
    if (__ballot(act) == 0ull) return;
    if (a.reset_slots && act && l16 == 0) a.cur_slot[b] = 0;   // see linearize_kernel: the trajectory now lives in slot 0
    ClockProbe cp;
    cp.start();
    const size_t B = a.B;
    const int N = a.N;
    T V = a.term[(size_t)(4 + l16) * B + b];
    T vx = a.term[(size_t)j * B + b];
    T* __restrict__ rec = a.gains + (size_t)b * R + (i == 0 ? j : NXA);
    const bool storer = act && ((i == 0 && j < NXA) || l16 == 4);
    const size_t rstride = B * R;
    bool all_pd = true;

    LaneConst<T> lc;
    lc.m0 = T(j == 0);
    lc.m1 = T(j == 1);
    lc.tr_byte = 4 * ((lane & 48) | (j << 2) | i);

    // DMA source of this lane: piece q of the slot = 16 B number (q % per_tile) of trajectory (q / per_tile)
    constexpr int PER_TILE = PIECES / 4;  // 12 / 24
    const char* src[NDMA];
    bool dma_lane[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int q = lane + 64 * d;
        dma_lane[d] = q < PIECES;
        const int qq = dma_lane[d] ? q : 0;
        int bb = blockIdx.x * 4 + qq / PER_TILE;
        bb = bb < a.B ? bb : a.B - 1;  // a tail wave re-reads the last trajectory, never out of bounds
        src[d] = reinterpret_cast<const char*>(a.lin + (size_t)bb * kTile16) + (qq % PER_TILE) * 16;
    }
    const size_t tbytes = B * kTile16 * sizeof(T);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) T*)ring;

    auto issue = [&](int t_tile, int slot) {
        const int tt = t_tile > 0 ? t_tile : 0;  // past the end of the sweep: harmless re-read of tile 0
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            if (dma_lane[d])
                lds_dma16(src[d] + (size_t)tt * tbytes,
                          (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + slot * SLOT_BYTES + d * 1024)));
    };

    // prologue: RING tiles in flight
    for (int u = 0; u < RING; ++u) issue(N - 1 - u, u);

    const T* mine = ring + grp * kTile16;
    Tile16<T> cur, nxt;
    wait_vmcnt<(RING - 1) * NDMA>();  // only DMAs so far: tile N-1 has landed
    tile16_load_lds(cur, mine, i, j, l16);
    int slot = 0;
    // One step.  Before tile t-1 (next slot) is read into registers one step ahead of its use, its DMA
    // must have retired.  Operations younger than that DMA: in steady state (RING-2) whole steps of
    // {1 store, NDMA DMAs}; during the first RING-2 steps fewer stores have been issued yet, so the
    // count there is bounded below by (RING-2)*NDMA -- the tight bound is used for those steps.
    auto body = [&](int t, auto wtag) {
        const int nslot = (slot + 1 == RING) ? 0 : slot + 1;
        wait_vmcnt<decltype(wtag)::value>();
        tile16_load_lds(nxt, mine + nslot * SLOT, i, j, l16);
        T Kj, kff;
        bool pd;
        tile16_step<T, REG>(cur, lc, a.mu, V, vx, Kj, kff, pd);
        all_pd = all_pd && pd;
        if (storer) rec[(size_t)t * rstride] = (i == 0) ? Kj : kff;
        // slot `slot` has been consumed (its values are in `cur`, already used): refill it RING steps ahead
        issue(t - RING, slot);
        cur = nxt;
        slot = nslot;
    };
    int t = N - 1;
    for (int s = 0; s < RING - 2 && t >= 0; ++s, --t) body(t, std::integral_constant<int, (RING - 2) * NDMA>());
    for (; t >= 0; --t) body(t, std::integral_constant<int, (RING - 2) * (NDMA + 1)>());
    wait_vmcnt<0>();  // drain the DMAs that ran past the end before the LDS allocation is released
    cp.stop(a.probe, 0);
    if (act && l16 == 0 && !all_pd) a.status[b] = st | ILQR_TRAJ_FLAG_NON_PD;
}

// position in the 48-scalar tile of entry e of the dense ILQR_LIN record of an (n, 1) system, n <= 4:
// record = [f_x n*n | f_u n | l_x n | l_u | l_xx n*n | l_ux n | l_uu]
ILQR_DEV int tile16_index_of(int e, int n) {
    const int nn = n * n;
    if (e < nn) { const int i = e / n, j = e % n; return 4 * j + ((i - j + 4) & 3); }   // f_x[i][j]
    e -= nn;
    if (e < n) return 32 + 4 * e;                    // f_u[i]
    e -= n;
    if (e < n) return 32 + 4 * e + 1;                // l_x[i]
    e -= n;
    if (e == 0) return 35;                           // l_u
    e -= 1;
    if (e < nn) return 16 + 4 * (e / n) + (e % n);   // l_xx[i][j]
    e -= nn;
    if (e < n) return 32 + 4 * e + 2;                // l_ux[j]
    return 39;                                       // l_uu
}

template <typename T>
__global__ void tile16_gather_dense_kernel(T* dense, const T* lin, int B, int N, int n) {
    const int E = 2 * n * n + 3 * n + 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * E) return;
    const int e = (int)(idx % E);
    const int t = (int)((idx / E) % N);
    const int b = (int)(idx / ((size_t)E * N));
    dense[idx] = lin[((size_t)t * B + b) * kTile16 + tile16_index_of(e, n)];
}

// inverse: dense ILQR_LIN records -> tiles (pad scalars and the padding rows / columns are zeroed by the caller's memset)
template <typename T>
__global__ void tile16_scatter_dense_kernel(const T* dense, T* lin, int B, int N, int n) {
    const int E = 2 * n * n + 3 * n + 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * E) return;
    const int e = (int)(idx % E);
    const int t = (int)((idx / E) % N);
    const int b = (int)(idx / ((size_t)E * N));
    lin[((size_t)t * B + b) * kTile16 + tile16_index_of(e, n)] = dense[idx];
}

}  // namespace ilqr

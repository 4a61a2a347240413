// dynamics.hpp -- device-side systems, integrators and quadratic costs for gfx950.
//
// What the reference builds with JAX closures + autodiff in
// python/class_files/systems/system_base.py:25-251 is written out here as plain
// per-lane device functions templated on the scalar type: the continuous
// dynamics f_c and its Jacobians, the four integrators (+ their exact discrete
// Jacobians by the chain rule through the stages = what jacfwd evaluates,
// system_base.py:203-205; implicit-function theorem for backward Euler,
// :146-188) and the quadratic stage / terminal costs with their derivatives
// (:212-219).  One lane evaluates one (trajectory, timestep) point; all loops
// have compile-time bounds so every array lives in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ilqr_hip.h"

#define ILQR_DEV __device__ __forceinline__

namespace ilqr {

// Lean sin/cos pair, branch-free, both values from one reduction (the libm entry points carry a Payne-Hanek
// slow path and a branch per call, which dominated the rollout's instruction stream).
// Both precisions reduce by pi (n from the 1.5 * 2^23 / 1.5 * 2^52 rounding constant, so no rndne / cvt), evaluate
// minimax polynomials on [-pi/2, pi/2], and apply ONE sign to both values (sin(r + n pi) = (-1)^n sin r, same for
// cos) taken straight from the low bit of the rounding constant's sum.  Reducing by pi/2 instead costs a quadrant
// select of ~11 integer / compare / select instructions per angle -- measured at 22 % of the fp32 RK4 rollout's
// instruction stream -- against two to four more FMAs here.  Absolute error (checked on the CPU against libm /
// 50-digit arithmetic with the same constants and operation order): float <= 1.1e-7 (sin), 1.5e-7 (cos); double
// <= 2.1e-16, for |x| < 1e3, graceful beyond -- pendulum angles never leave that range on a rollout whose cost is
// still finite.  The float sincos2 evaluates two angles in packed FP32 (v_pk_fma_f32: two lanes' worth of FMA per
// issue slot) with exactly the arithmetic of sincos, so both give bit-identical results.
template <typename T> struct M;
template <> struct M<float> {
    typedef float f2 __attribute__((ext_vector_type(2)));
    static ILQR_DEV float sqrt(float x) { return sqrtf(x); }
    static ILQR_DEV float abs(float x) { return fabsf(x); }
    static constexpr float kMagic = 12582912.0f;             // 1.5 * 2^23: x/pi + kMagic rounds to nearest int
    static constexpr float kInvPi = 0x1.45f306p-2f;
    // pi = kPiHi + kPiMid - 3.4e-15: the third Cody-Waite term would move r by n * 3.4e-15 <= 1e-12 for |x| < 1e3, five
    // orders below an fp32 ulp of r, so two terms are all the reduction takes
    static constexpr float kPiHi = 0x1.921fb6p+1f, kPiMid = -0x1.777a5cp-24f;
    static constexpr float kS0 = -0x1.555548p-3f, kS1 = 0x1.110e42p-7f, kS2 = -0x1.9f588ap-13f, kS3 = 0x1.5c90a2p-19f;
    static constexpr float kC0 = -0.5f, kC1 = 0x1.555546p-5f, kC2 = -0x1.6c134cp-10f, kC3 = 0x1.9f68bp-16f,
                           kC4 = -0x1.17b08ap-22f;
    static ILQR_DEV void sincos(float x, float* sn, float* cs) {
        const float t = fmaf(x, kInvPi, kMagic);
        const float n = t - kMagic;
        float r = fmaf(-n, kPiHi, x);
        r = fmaf(-n, kPiMid, r);
        const float z = r * r;
        const float ps = fmaf(fmaf(fmaf(kS3, z, kS2), z, kS1), z, kS0);
        const float s = fmaf(r * z, ps, r);
        const float pc = fmaf(fmaf(fmaf(fmaf(kC4, z, kC3), z, kC2), z, kC1), z, kC0);
        const float c = fmaf(z, pc, 1.0f);
        const unsigned sign = __float_as_uint(t) << 31;      // parity of n
        *sn = __uint_as_float(__float_as_uint(s) ^ sign);
        *cs = __uint_as_float(__float_as_uint(c) ^ sign);
    }
    static ILQR_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
    static ILQR_DEV f2 splat(float v) { f2 r; r.x = v; r.y = v; return r; }
    // sin / cos of a pair of angles in packed FP32: exactly the arithmetic of sincos() on each half
    static ILQR_DEV void sincos_pk(f2 x, f2* sn, f2* cs) {
        const f2 t = fma2(x, splat(kInvPi), splat(kMagic));
        const f2 n = t - splat(kMagic);
        f2 r = fma2(-n, splat(kPiHi), x);
        r = fma2(-n, splat(kPiMid), r);
        const f2 z = r * r;
        const f2 ps = fma2(fma2(fma2(splat(kS3), z, splat(kS2)), z, splat(kS1)), z, splat(kS0));
        const f2 s = fma2(r * z, ps, r);
        const f2 pc = fma2(fma2(fma2(fma2(splat(kC4), z, splat(kC3)), z, splat(kC2)), z, splat(kC1)), z, splat(kC0));
        const f2 c = fma2(z, pc, splat(1.0f));
        // (-1)^n as a float pair, applied with two packed multiplies (exact): 4 instructions for the four signs
        // instead of 2 shifts + 4 xors
        f2 sg;
        sg.x = __uint_as_float((__float_as_uint(t.x) << 31) | 0x3f800000u);
        sg.y = __uint_as_float((__float_as_uint(t.y) << 31) | 0x3f800000u);
        *sn = s * sg;
        *cs = c * sg;
    }
    static ILQR_DEV void sincos2(float x0, float x1, float* s0, float* c0, float* s1, float* c1) {
        f2 x; x.x = x0; x.y = x1;
        f2 ss, cc;
        sincos_pk(x, &ss, &cc);
        *s0 = ss.x; *c0 = cc.x; *s1 = ss.y; *c1 = cc.y;
    }
};
template <> struct M<double> {
    static ILQR_DEV double sqrt(double x) { return ::sqrt(x); }
    static ILQR_DEV double abs(double x) { return fabs(x); }
    // same scheme as the float version: reduction by pi through the 1.5 * 2^52 rounding constant, near-minimax
    // polynomials of degree 7 in r^2 on [-pi/2, pi/2] (Chebyshev-node fits computed with 40-digit arithmetic; error
    // of the rounded polynomials 4.2e-17 / 1.4e-17), one sign for both values.  |error| <= ~1 ulp for |x| < 1e3.
    static ILQR_DEV void sincos(double x, double* sn, double* cs) {
        constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52
        const double t = fma(x, 0x1.45f306dc9c883p-2, kMagic);
        const double n = t - kMagic;
        double r = fma(-n, 0x1.921fb54442d18p+1, x);
        r = fma(-n, 0x1.1a62633145c07p-53, r);
        r = fma(-n, -0x1.f1976b7ed8fbcp-109, r);
        const double z = r * r;
        double ps = 0x1.892efd890db97p-49;
        ps = fma(ps, z, -0x1.ae4d771729416p-41);
        ps = fma(ps, z, 0x1.6123f55a16315p-33);
        ps = fma(ps, z, -0x1.ae64557c0cbf9p-26);
        ps = fma(ps, z, 0x1.71de3a5419df5p-19);
        ps = fma(ps, z, -0x1.a01a01a0184f4p-13);
        ps = fma(ps, z, 0x1.1111111111104p-7);
        ps = fma(ps, z, -0x1.5555555555555p-3);
        const double s = fma(r * z, ps, r);
        double pc = -0x1.5e8cb6756109ep-53;
        pc = fma(pc, z, 0x1.ae5759e5592bep-45);
        pc = fma(pc, z, -0x1.9396f4137a4d0p-37);
        pc = fma(pc, z, 0x1.1eed8e979ff66p-29);
        pc = fma(pc, z, -0x1.27e4fb77023cap-22);
        pc = fma(pc, z, 0x1.a01a01a019538p-16);
        pc = fma(pc, z, -0x1.6c16c16c16c11p-10);
        pc = fma(pc, z, 0x1.5555555555555p-5);
        const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
        const int sign = __double2loint(t) << 31;       // parity of n
        *sn = __hiloint2double(__double2hiint(s) ^ sign, __double2loint(s));
        *cs = __hiloint2double(__double2hiint(c) ^ sign, __double2loint(c));
    }
    static ILQR_DEV void sincos2(double x0, double x1, double* s0, double* c0, double* s1, double* c1) {
        sincos(x0, s0, c0);
        sincos(x1, s1, c1);
    }
};

// reciprocal = hardware estimate + Newton steps (1 for float, 2 for double): ~1 ulp, 3-5 instructions
// instead of the ~10-15 of the IEEE division sequence; used where the divisor is a well-scaled positive
// number (det M(q) of the pendulum mass matrix, Q_uu of the backward sweep).
ILQR_DEV float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
ILQR_DEV double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// reciprocal square root = hardware estimate + Newton steps (1 for float, 2 for double): ~1 ulp; the Cholesky of the
// wave sweep takes L_cc = d * rsqrt(d) and 1 / L_cc = rsqrt(d) from it instead of an IEEE sqrt and an IEEE division
// per column on the step's serial chain
ILQR_DEV float fast_rsqrt(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x, r * r, 1.5f);
}
ILQR_DEV double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x, r * r, 1.5);
    return r * fma(-0.5 * x, r * r, 1.5);
}

// ---- two points per lane in packed FP32 ---------------------------------------------------------------------
// The templates below (systems, integrators, costs) are written on a scalar type T; instantiated on a float PAIR
// (clang's ext_vector_type(2): element-wise + - * /, a * b + c contracted to v_pk_fma_f32) every lane evaluates two
// (trajectory, time) points at the issue cost of one: v_pk_fma_f32 issues at the rate of v_fma_f32
// (tools/micro/pk_rate.hip).  Each half runs exactly the scalar code's operations in the scalar code's order, so the
// results are bit-identical to two scalar evaluations.  Used by the producers of backward_fused16_kernel.  The
// parameter block stays scalar (SGPRs): SplatParams presents it as pairs.  Backward Euler (a data-dependent Newton
// loop) has no pair form: is_scalar guards it.
// (a bool converts to an ext vector as 0 / -1, not 0 / 1: the templates write `cond ? T(1) : T(0)`, never T(cond))
typedef float pair_f32 __attribute__((ext_vector_type(2)));
template <typename T> struct is_scalar { static constexpr bool value = true; };
template <> struct is_scalar<pair_f32> { static constexpr bool value = false; };
template <> struct M<pair_f32> {
    static ILQR_DEV pair_f32 sqrt(pair_f32 x) { return x; }   // (backward Euler only: never instantiated for pairs)
    static ILQR_DEV pair_f32 abs(pair_f32 x) { return x; }
    static ILQR_DEV void sincos2(pair_f32 x0, pair_f32 x1, pair_f32* s0, pair_f32* c0, pair_f32* s1, pair_f32* c1) {
        M<float>::sincos_pk(x0, s0, c0);
        M<float>::sincos_pk(x1, s1, c1);
    }
    static ILQR_DEV void sincos(pair_f32 x, pair_f32* sn, pair_f32* cs) { M<float>::sincos_pk(x, sn, cs); }
};
ILQR_DEV pair_f32 fast_rcp(pair_f32 x) {
    pair_f32 r;
    r.x = __builtin_amdgcn_rcpf(x.x);
    r.y = __builtin_amdgcn_rcpf(x.y);
    const pair_f32 one = 1.0f;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-x, r, one), r, r);
}
struct SplatParams {
    const float* q;
    ILQR_DEV pair_f32 operator[](int i) const { pair_f32 r = q[i]; return r; }
};

// ---------------------------------------------------------------------------
// Device parameter block (scalars of type T, built on the host in double by
// build_device_params() in ilqr_abi.hip):
//   [ derived system constants (NSYS) | x_target (n) | Q (n*n) | R (m*m) | Q_f (n*n)
//     | Qs (n*n) | Rs (m*m) | Qfs (n*n) ]      Xs = 0.5*(X + X')
// The pointer is a kernel argument indexed by compile-time constants, so the
// compiler reads it with scalar loads (s_load) -- no VGPRs, no LDS.
// ---------------------------------------------------------------------------
template <int NSYS, int NX, int NU> struct ParamLayout {
    static constexpr int SYS = 0;
    static constexpr int XT = NSYS;
    static constexpr int Q = XT + NX;
    static constexpr int R = Q + NX * NX;
    static constexpr int QF = R + NU * NU;
    static constexpr int QS = QF + NX * NX;
    static constexpr int RS = QS + NX * NX;
    static constexpr int QFS = RS + NU * NU;
    static constexpr int TOTAL = QFS + NX * NX;
};

// ---- pendulum (pendulum_sys.py:60-75): derived constants [g/l, d] ------------
template <typename T> struct Pendulum {
    template <typename S> using rebind = Pendulum<S>;
    static constexpr int NX = 2, NU = 1, NSYS = 2, ID = ILQR_SYS_PENDULUM;
    static constexpr bool SECOND_ORDER = true;
    template <typename P> static ILQR_DEV void f(P p, const T* x, const T* u, T* xd) {
        xd[0] = x[1];
        T s, c;
        M<T>::sincos(x[0], &s, &c);
        xd[1] = u[0] - p[1] * x[1] - p[0] * s;
    }
    template <typename P> static ILQR_DEV void fjac(P p, const T* x, const T* u, T* xd, T (*Jx)[2], T (*Ju)[1]) {
        T s, c;
        M<T>::sincos(x[0], &s, &c);
        xd[0] = x[1];
        xd[1] = u[0] - p[1] * x[1] - p[0] * s;
        Jx[0][0] = T(0); Jx[0][1] = T(1);
        Jx[1][0] = -p[0] * c; Jx[1][1] = -p[1];
        Ju[0][0] = T(0); Ju[1][0] = T(1);
    }
};

// ---- double pendulum (UA_double_pendulum_sys.py:84-208, double_pendulum_sys.py:84-206)
// derived constants: [a = m2 l1 l2, c11 = m1 l1^2/4 + m2 l1^2 + m2 l2^2/4 + th1 + th2,
//                     c12 = m2 l2^2/4 + th2 (= m22), gA = m2 g l2/2, gB = (m2 + m1/2) g l1, d1, d2]
template <typename T, int NU_> struct DoublePendulum {
    template <typename S> using rebind = DoublePendulum<S, NU_>;
    static constexpr int NX = 4, NU = NU_, NSYS = 7;
    static constexpr bool SECOND_ORDER = true;
    static constexpr int ID = (NU_ == 1) ? ILQR_SYS_UA_DOUBLE_PENDULUM : ILQR_SYS_DOUBLE_PENDULUM;

    template <typename P> static ILQR_DEV void f(P p, const T* x, const T* u, T* xd) {
        const T a = p[0], c11 = p[1], c12 = p[2], gA = p[3], gB = p[4], d1 = p[5], d2 = p[6];
        const T q1 = x[0], q2 = x[1], q1d = x[2], q2d = x[3];
        // sin(q1 + q2) by the addition theorem: two reductions per evaluation instead of three
        T s1, c1, s2, c2;
        M<T>::sincos2(q1, q2, &s1, &c1, &s2, &c2);
        const T s12 = s1 * c2 + c1 * s2;
        const T m11 = c11 + a * c2, m12 = c12 + T(0.5) * a * c2, m22 = c12;
        const T as2 = a * s2;
        T h1 = u[0] + T(0.5) * as2 * (T(2) * q1d * q2d + q2d * q2d) - gA * s12 - gB * s1 - d1 * q1d;
        T h2 = -T(0.5) * as2 * q1d * q1d - gA * s12 - d2 * q2d;
        if (NU == 2) h2 += u[NU - 1];
        const T idet = fast_rcp(m11 * m22 - m12 * m12);
        xd[0] = q1d;
        xd[1] = q2d;
        xd[2] = (m22 * h1 - m12 * h2) * idet;
        xd[3] = (m11 * h2 - m12 * h1) * idet;
    }

    // ---- the fp32 rollout's RK4 step on PAIRS: Q = (q1, q2), W = (q1', q2') in packed FP32 (v_pk_fma_f32 issues at the
    // rate of v_fma_f32, tools/micro/pk_rate.hip, and the rollout is bound by its instruction count).  The same formulas
    // as f() above with the two generalised forces, the two rows of the 2 x 2 solve and the RK4 combinations each as
    // ONE packed operation; (m22, m11) is built as a pair so the solve needs no register shuffles.  Differences to f()
    // are association only: w = (2 q1' + q2') q2' for 2 q1' q2' + q2'^2, and (a s2 / 2) q1'^2 for ((a s2 / 2) q1') q1'.
    static constexpr bool RK4_PK = true;
    typedef float pf2 __attribute__((ext_vector_type(2)));
    static ILQR_DEV pf2 pk_fma(pf2 a, pf2 b, pf2 c) { return __builtin_elementwise_fma(a, b, c); }
    static ILQR_DEV pf2 pk_splat(float v) { pf2 r; r.x = v; r.y = v; return r; }
    static ILQR_DEV pf2 acc_pk(const float* __restrict__ p, pf2 Q, pf2 W, pf2 uu) {
        const float a = p[0], c11 = p[1], c12 = p[2], gA = p[3], gB = p[4], d1 = p[5], d2 = p[6];
        pf2 S, C;
        M<float>::sincos_pk(Q, &S, &C);                            // (s1, s2), (c1, c2)
        const float s12 = fmaf(C.x, S.y, S.x * C.y);               // sin(q1 + q2)
        pf2 cm; cm.x = c12; cm.y = c11;
        pf2 am; am.x = 0.0f; am.y = a;
        const pf2 Mx = pk_fma(am, pk_splat(C.y), cm);              // (m22, m11)
        const float m12 = fmaf(0.5f * a, C.y, c12);
        const float as2h = (0.5f * a) * S.y;
        pf2 t; t.x = fmaf(2.0f, W.x, W.y); t.y = -W.x;
        const pf2 wv = t * W.yx;                                   // (w, -q1'^2)
        pf2 H = pk_fma(pk_splat(as2h), wv, uu);                    // (u1 + a s2 w / 2, u2 - a s2 q1'^2 / 2)
        H = pk_fma(pk_splat(-gA), pk_splat(s12), H);
        H.x = fmaf(-gB, S.x, H.x);
        pf2 dd; dd.x = d1; dd.y = d2;
        H = pk_fma(-dd, W, H);                                     // (h1, h2)
        const float idet = fast_rcp(fmaf(Mx.y, c12, -(m12 * m12)));
        const pf2 num = pk_fma(Mx, H, -(pk_splat(m12) * H.yx));    // (m22 h1 - m12 h2, m11 h2 - m12 h1)
        return num * pk_splat(idet);
    }
    static ILQR_DEV void rk4_pk(const float* __restrict__ p, float dt, const float* x, const float* u, float* xn) {
        pf2 Q, W, uu;
        Q.x = x[0]; Q.y = x[1]; W.x = x[2]; W.y = x[3];
        uu.x = u[0]; uu.y = (NU == 2) ? u[NU - 1] : 0.0f;
        const pf2 hh = pk_splat(dt / 2.0f), h = pk_splat(dt), h6 = pk_splat(dt / 6.0f), two = pk_splat(2.0f);
        const pf2 A1 = acc_pk(p, Q, W, uu);
        const pf2 W2 = pk_fma(hh, A1, W);
        const pf2 A2 = acc_pk(p, pk_fma(hh, W, Q), W2, uu);
        const pf2 W3 = pk_fma(hh, A2, W);
        const pf2 A3 = acc_pk(p, pk_fma(hh, W2, Q), W3, uu);
        const pf2 W4 = pk_fma(h, A3, W);
        const pf2 A4 = acc_pk(p, pk_fma(h, W3, Q), W4, uu);
        const pf2 Qn = pk_fma(h6, pk_fma(two, W3, pk_fma(two, W2, W)) + W4, Q);
        const pf2 Wn = pk_fma(h6, pk_fma(two, A3, pk_fma(two, A2, A1)) + A4, W);
        xn[0] = Qn.x; xn[1] = Qn.y; xn[2] = Wn.x; xn[3] = Wn.y;
    }

    // M qdd = h  =>  d qdd = M^-1 (dh - dM qdd)
    template <typename P> static ILQR_DEV void fjac(P p, const T* x, const T* u, T* xd, T (*Jx)[4], T (*Ju)[NU]) {
        const T a = p[0], c11 = p[1], c12 = p[2], gA = p[3], gB = p[4], d1 = p[5], d2 = p[6];
        const T q1 = x[0], q2 = x[1], q1d = x[2], q2d = x[3];
        T s1, c1, s2, c2;
        M<T>::sincos2(q1, q2, &s1, &c1, &s2, &c2);
        const T s12 = s1 * c2 + c1 * s2, c12q = c1 * c2 - s1 * s2;
        const T m11 = c11 + a * c2, m12 = c12 + T(0.5) * a * c2, m22 = c12;
        const T as2 = a * s2, ac2 = a * c2;
        const T w = T(2) * q1d * q2d + q2d * q2d;
        T h1 = u[0] + T(0.5) * as2 * w - gA * s12 - gB * s1 - d1 * q1d;
        T h2 = -T(0.5) * as2 * q1d * q1d - gA * s12 - d2 * q2d;
        if (NU == 2) h2 += u[NU - 1];
        const T idet = fast_rcp(m11 * m22 - m12 * m12);
        const T i11 = m22 * idet, i12 = -m12 * idet, i22 = m11 * idet;
        const T qdd1 = i11 * h1 + i12 * h2, qdd2 = i12 * h1 + i22 * h2;
        xd[0] = q1d; xd[1] = q2d; xd[2] = qdd1; xd[3] = qdd2;
        // dh/dz, z = q1, q2, q1d, q2d
        T dh1[4], dh2[4];
        dh1[0] = -gA * c12q - gB * c1;
        dh2[0] = -gA * c12q;
        // dM/dq2 = [[-a s2, -a s2/2], [-a s2/2, 0]]
        dh1[1] = T(0.5) * ac2 * w - gA * c12q + as2 * qdd1 + T(0.5) * as2 * qdd2;
        dh2[1] = -T(0.5) * ac2 * q1d * q1d - gA * c12q + T(0.5) * as2 * qdd1;
        dh1[2] = as2 * q2d - d1;
        dh2[2] = -as2 * q1d;
        dh1[3] = as2 * (q1d + q2d);
        dh2[3] = -d2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Jx[0][j] = (j == 2 ? T(1) : T(0));
            Jx[1][j] = (j == 3 ? T(1) : T(0));
            Jx[2][j] = i11 * dh1[j] + i12 * dh2[j];
            Jx[3][j] = i12 * dh1[j] + i22 * dh2[j];
        }
        Ju[0][0] = T(0); Ju[1][0] = T(0); Ju[2][0] = i11; Ju[3][0] = i12;
        if (NU == 2) {
            Ju[0][NU - 1] = T(0); Ju[1][NU - 1] = T(0); Ju[2][NU - 1] = i12; Ju[3][NU - 1] = i22;
        }
    }
};

// ---- linear system x_dot = A x + B u (matlab/CLASSES/Linear_iLQR_CLASS.m:56-60)
// derived constants: A (n*n), B (n*m) row-major
template <typename T, int NX_, int NU_> struct Linear {
    template <typename S> using rebind = Linear<S, NX_, NU_>;
    static constexpr int NX = NX_, NU = NU_, NSYS = NX_ * NX_ + NX_ * NU_, ID = ILQR_SYS_LINEAR;
    template <typename P> static ILQR_DEV void f(P p, const T* x, const T* u, T* xd) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) acc += p[i * NX + j] * x[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) acc += p[NX * NX + i * NU + j] * u[j];
            xd[i] = acc;
        }
    }
    template <typename P> static ILQR_DEV void fjac(P p, const T* x, const T* u, T* xd, T (*Jx)[NX_], T (*Ju)[NU_]) {
        f(p, x, u, xd);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) Jx[i][j] = p[i * NX + j];
#pragma unroll
            for (int j = 0; j < NU; ++j) Ju[i][j] = p[NX * NX + i * NU + j];
        }
    }
};

// ---------------------------------------------------------------------------
// small dense helpers (compile-time sizes, static indexing only)
// ---------------------------------------------------------------------------
// In-place LU with partial pivoting realised as predicated row swaps, so no
// dynamically indexed register arrays (= no scratch).  Solves A X = RHS for NR
// right-hand sides held as columns of rhs[N][NR].
template <typename T, int N, int NR>
ILQR_DEV void lu_solve_inplace(T (*A)[N], T (*rhs)[NR]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // pivot search
        T best = M<T>::abs(A[k][k]);
        int piv = k;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            T v = M<T>::abs(A[i][k]);
            if (v > best) { best = v; piv = i; }
        }
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = (piv == i);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T a = A[k][j], b = A[i][j];
                A[k][j] = sw ? b : a;
                A[i][j] = sw ? a : b;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                T a = rhs[k][j], b = rhs[i][j];
                rhs[k][j] = sw ? b : a;
                rhs[i][j] = sw ? a : b;
            }
        }
        const T inv = T(1) / A[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = A[i][k] * inv;
#pragma unroll
            for (int j = k + 1; j < N; ++j) A[i][j] -= l * A[k][j];
#pragma unroll
            for (int j = 0; j < NR; ++j) rhs[i][j] -= l * rhs[k][j];
        }
    }
    // back substitution
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        const T inv = T(1) / A[k][k];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            T acc = rhs[k][j];
#pragma unroll
            for (int i = k + 1; i < N; ++i) acc -= A[k][i] * rhs[i][j];
            rhs[k][j] = acc * inv;
        }
    }
}

// ---------------------------------------------------------------------------
// integrators (system_base.py:50-140) and their discrete Jacobians
// ---------------------------------------------------------------------------
// which systems get the multi-stage / implicit integrators: n_x <= 4 by default (the big linear systems only
// get the closed-form ones: register budget), or any Dyn that asks for them (user-defined plugins)
template <typename Dyn, typename = void> struct all_integrators { static constexpr bool value = (Dyn::NX <= 4); };
template <typename Dyn> struct all_integrators<Dyn, decltype((void)Dyn::ALL_INTEGRATORS)> {
    static constexpr bool value = Dyn::ALL_INTEGRATORS;
};

// mechanical systems x = [q, q_dot] declare SECOND_ORDER: the chain rule through the integrator stages then skips the
// trivial half of the continuous Jacobian (Stepper::stage)
template <typename Dyn, typename = void> struct second_order { static constexpr bool value = false; };
template <typename Dyn> struct second_order<Dyn, decltype((void)Dyn::SECOND_ORDER)> {
    static constexpr bool value = Dyn::SECOND_ORDER && Dyn::NX % 2 == 0;
};

// a Dyn with RK4_PK offers the fp32 rollout a hand-packed RK4 step (Dyn::rk4_pk)
template <typename T, typename Dyn, typename = void> struct rk4_packed { static constexpr bool value = false; };
#ifndef ILQR_NO_RK4_PK
template <typename Dyn> struct rk4_packed<float, Dyn, decltype((void)Dyn::RK4_PK)> { static constexpr bool value = Dyn::RK4_PK; };
#endif

template <typename T, typename Dyn> struct Stepper {
    static constexpr int NX = Dyn::NX, NU = Dyn::NU;
    static constexpr bool SMALL = all_integrators<Dyn>::value;

    // backward Euler quasi-Newton (system_base.py:88-140): explicit-Euler guess, one
    // Jacobian I - dt*J_x at the guess reused, stop at ||F||_2 <= 1e-5 or 20 iterations.
    template <typename P> static ILQR_DEV void backward_euler(P p, T dt, const T* x, const T* u, T* xn) {
        T k[NX], Jx[NX][NX], Ju[NX][NU];
        Dyn::f(p, x, u, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = x[i] + dt * k[i];
        Dyn::fjac(p, xn, u, k, Jx, Ju);
        T J[NX][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) J[i][j] = (i == j ? T(1) : T(0)) - dt * Jx[i][j];
        T F[NX], nrm2 = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { F[i] = xn[i] - x[i] - dt * k[i]; nrm2 += F[i] * F[i]; }
        int it = 0;
        while (M<T>::sqrt(nrm2) > T(1e-5) && it < 20) {
            T Jc[NX][NX], r[NX][1];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                r[i][0] = -F[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) Jc[i][j] = J[i][j];
            }
            lu_solve_inplace<T, NX, 1>(Jc, r);
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] += r[i][0];
            Dyn::f(p, xn, u, k);
            nrm2 = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) { F[i] = xn[i] - x[i] - dt * k[i]; nrm2 += F[i] * F[i]; }
            ++it;
        }
    }

    template <typename P> static ILQR_DEV void step(int integ, P p, T dt, const T* x, const T* u, T* xn) {
        T k1[NX];
        if (integ == ILQR_INT_DISCRETE) { Dyn::f(p, x, u, xn); return; }
        if (integ == ILQR_INT_EULER || !SMALL) {
            Dyn::f(p, x, u, k1);
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = x[i] + k1[i] * dt;
            return;
        }
        if constexpr (SMALL) {
            if (integ == ILQR_INT_MIDPOINT) {
                T xm[NX], k2[NX];
                Dyn::f(p, x, u, k1);
#pragma unroll
                for (int i = 0; i < NX; ++i) xm[i] = x[i] + (dt / T(2)) * k1[i];
                Dyn::f(p, xm, u, k2);
#pragma unroll
                for (int i = 0; i < NX; ++i) xn[i] = x[i] + dt * k2[i];
                return;
            }
            if (integ == ILQR_INT_RK4) {
                if constexpr (rk4_packed<T, Dyn>::value) {
                    Dyn::rk4_pk(p, dt, x, u, xn);
                    return;
                }
                T xs[NX], k2[NX], k3[NX], k4[NX];
                Dyn::f(p, x, u, k1);
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[i] = x[i] + dt / T(2) * k1[i];
                Dyn::f(p, xs, u, k2);
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[i] = x[i] + dt / T(2) * k2[i];
                Dyn::f(p, xs, u, k3);
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[i] = x[i] + dt * k3[i];
                Dyn::f(p, xs, u, k4);
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    xn[i] = x[i] + (dt / T(6)) * (k1[i] + T(2) * k2[i] + T(2) * k3[i] + k4[i]);
                return;
            }
            if constexpr (is_scalar<T>::value) backward_euler(p, dt, x, u, xn);
        }
    }

    // C = A*B helpers on register tiles
    template <int R, int K, int C>
    static ILQR_DEV void matmul(const T (*A)[K], const T (*B)[C], T (*out)[C]) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < K; ++k) acc += A[i][k] * B[k][j];
                out[i][j] = acc;
            }
    }

    // one explicit stage of the chain rule: given the stage point xs = x + c*k_prev with
    // d xs/dx = I + c*Kx_prev, d xs/du = c*Ku_prev, evaluate k = f_c(xs, u) and
    // Kx = J_x(xs) (I + c Kx_prev), Ku = J_x(xs) (c Ku_prev) + J_u(xs).
    template <typename P> static ILQR_DEV void stage(P p, const T* x, const T* u, T c, const T* kprev,
                               const T (*Kxp)[NX], const T (*Kup)[NU], T* k, T (*Kx)[NX], T (*Ku)[NU]) {
        T xs[NX], Jx[NX][NX], Ju[NX][NU], Dx[NX][NX], Du[NX][NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            xs[i] = x[i] + c * kprev[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) Dx[i][j] = (i == j ? T(1) : T(0)) + c * Kxp[i][j];
#pragma unroll
            for (int j = 0; j < NU; ++j) Du[i][j] = c * Kup[i][j];
        }
        Dyn::fjac(p, xs, u, k, Jx, Ju);
        if constexpr (second_order<Dyn>::value) {
            // x = [q, q_dot], x_dot = [q_dot, a(q, q_dot, u)]: the top half of J_x is [0 I] and of J_u is 0, so
            // the top rows of the products are rows of D (same values as multiplying through: 0 * d + 1 * d), and
            // only the acceleration rows cost multiply-adds -- half of the chain rule's arithmetic
            constexpr int NQ = NX / 2;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
#pragma unroll
                for (int j = 0; j < NX; ++j) Kx[i][j] = Dx[NQ + i][j];
#pragma unroll
                for (int j = 0; j < NU; ++j) Ku[i][j] = Du[NQ + i][j];
            }
#pragma unroll
            for (int i = NQ; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    T acc = T(0);
#pragma unroll
                    for (int s = 0; s < NX; ++s) acc += Jx[i][s] * Dx[s][j];
                    Kx[i][j] = acc;
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    T acc = T(0);
#pragma unroll
                    for (int s = 0; s < NX; ++s) acc += Jx[i][s] * Du[s][j];
                    Ku[i][j] = acc + Ju[i][j];
                }
            }
        } else {
            matmul<NX, NX, NX>(Jx, Dx, Kx);
            matmul<NX, NX, NU>(Jx, Du, Ku);
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NU; ++j) Ku[i][j] += Ju[i][j];
        }
    }

    // f, f_x, f_u of the discrete map at (x, u)
    template <typename P> static ILQR_DEV void step_jac(int integ, P p, T dt, const T* x, const T* u, T* xn,
                                  T (*fx)[NX], T (*fu)[NU]) {
        T k1[NX];
        if (integ == ILQR_INT_DISCRETE) { Dyn::fjac(p, x, u, xn, fx, fu); return; }
        if (integ == ILQR_INT_EULER || !SMALL) {
            Dyn::fjac(p, x, u, k1, fx, fu);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                xn[i] = x[i] + k1[i] * dt;
#pragma unroll
                for (int j = 0; j < NX; ++j) fx[i][j] = (i == j ? T(1) : T(0)) + dt * fx[i][j];
#pragma unroll
                for (int j = 0; j < NU; ++j) fu[i][j] = dt * fu[i][j];
            }
            return;
        }
        if constexpr (SMALL) {
            T K1x[NX][NX], K1u[NX][NU];
            if (integ == ILQR_INT_MIDPOINT) {
                T k2[NX], K2x[NX][NX], K2u[NX][NU];
                Dyn::fjac(p, x, u, k1, K1x, K1u);
                stage(p, x, u, dt / T(2), k1, K1x, K1u, k2, K2x, K2u);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    xn[i] = x[i] + dt * k2[i];
#pragma unroll
                    for (int j = 0; j < NX; ++j) fx[i][j] = (i == j ? T(1) : T(0)) + dt * K2x[i][j];
#pragma unroll
                    for (int j = 0; j < NU; ++j) fu[i][j] = dt * K2u[i][j];
                }
                return;
            }
            if (integ == ILQR_INT_RK4) {
                T k[NX], Kx[NX][NX], Ku[NX][NU], kn[NX], Knx[NX][NX], Knu[NX][NU];
                T sk[NX], sx[NX][NX], su[NX][NU];
                Dyn::fjac(p, x, u, k, Kx, Ku);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    sk[i] = k[i];
#pragma unroll
                    for (int j = 0; j < NX; ++j) sx[i][j] = Kx[i][j];
#pragma unroll
                    for (int j = 0; j < NU; ++j) su[i][j] = Ku[i][j];
                }
                const T cs[3] = {dt / T(2), dt / T(2), dt};
                const T ws[3] = {T(2), T(2), T(1)};
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    stage(p, x, u, cs[s], k, Kx, Ku, kn, Knx, Knu);
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        k[i] = kn[i];
                        sk[i] += ws[s] * kn[i];
#pragma unroll
                        for (int j = 0; j < NX; ++j) { Kx[i][j] = Knx[i][j]; sx[i][j] += ws[s] * Knx[i][j]; }
#pragma unroll
                        for (int j = 0; j < NU; ++j) { Ku[i][j] = Knu[i][j]; su[i][j] += ws[s] * Knu[i][j]; }
                    }
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    xn[i] = x[i] + (dt / T(6)) * sk[i];
#pragma unroll
                    for (int j = 0; j < NX; ++j) fx[i][j] = (i == j ? T(1) : T(0)) + (dt / T(6)) * sx[i][j];
#pragma unroll
                    for (int j = 0; j < NU; ++j) fu[i][j] = (dt / T(6)) * su[i][j];
                }
                return;
            }
            // backward Euler: implicit-function theorem at the converged point
            // (system_base.py:146-188): f_x = (I - dt J_x)^-1, f_u = (I - dt J_x)^-1 dt J_u
            if constexpr (is_scalar<T>::value) {
            backward_euler(p, dt, x, u, xn);
            T Jx[NX][NX], Ju[NX][NU], J[NX][NX], rhs[NX][NX + NU];
            Dyn::fjac(p, xn, u, k1, Jx, Ju);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j < NX; ++j) { J[i][j] = (i == j ? T(1) : T(0)) - dt * Jx[i][j]; rhs[i][j] = (i == j ? T(1) : T(0)); }
#pragma unroll
                for (int j = 0; j < NU; ++j) rhs[i][NX + j] = dt * Ju[i][j];
            }
            lu_solve_inplace<T, NX, NX + NU>(J, rhs);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j < NX; ++j) fx[i][j] = rhs[i][j];
#pragma unroll
                for (int j = 0; j < NU; ++j) fu[i][j] = rhs[i][NX + j];
            }
            }
        }
    }
};

// ---------------------------------------------------------------------------
// costs.  Built-in systems: the quadratic forms of pendulum_sys.py:77-98 / UA_double_pendulum_sys.py:114-136,
// read from the parameter block.  A Dyn with CUSTOM_COST (a user system whose _l_fcn / _l_f_fcn were traced,
// system_base.py:262-275) supplies l, lf and their first and second derivatives as generated code instead;
// like the reference's _l_fcn it returns the stage cost as the user wrote it (dt scaling included or not).
// ---------------------------------------------------------------------------
template <typename Dyn, typename = void> struct has_custom_cost { static constexpr bool value = false; };
template <typename Dyn> struct has_custom_cost<Dyn, decltype((void)Dyn::CUSTOM_COST)> {
    static constexpr bool value = Dyn::CUSTOM_COST;
};

template <typename T, typename Dyn> struct Cost {
    static constexpr int NX = Dyn::NX, NU = Dyn::NU;
    static constexpr bool CUSTOM = has_custom_cost<Dyn>::value;
    using L = ParamLayout<Dyn::NSYS, NX, NU>;
#ifdef ILQR_NO_COST_PK
    static constexpr bool stage_cost_packed = false;
#else
    static constexpr bool stage_cost_packed = true;
#endif

    // l(x,u) = (0.5 dx'Q dx + 0.5 u'R u) * dt
    template <typename P> static ILQR_DEV T stage(P p, T dt, const T* x, const T* u) {
        if constexpr (CUSTOM) {
            return Dyn::l(x, u);
        } else if constexpr (sizeof(T) == 4 && NX == 4 && stage_cost_packed) {
            // fp32, n_x = 4 (the rollout's hot case): dx'Q dx as (Q'dx)'dx with the two halves of Q'dx accumulated in
            // packed FP32 -- row i of Q contributes (Q_i0, Q_i1) dx_i and (Q_i2, Q_i3) dx_i, adjacent in the row-major
            // block -- 11 vector instructions instead of 24; the sum is the same up to the order of its 16 terms.
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto pair = [](float a, float b) { f2 r; r.x = a; r.y = b; return r; };
            auto bc = [](float a) { f2 r; r.x = a; r.y = a; return r; };
            const f2 d01 = pair(x[0], x[1]) - pair(p[L::XT + 0], p[L::XT + 1]);
            const f2 d23 = pair(x[2], x[3]) - pair(p[L::XT + 2], p[L::XT + 3]);
            f2 r01 = pair(p[L::Q + 0], p[L::Q + 1]) * bc(d01.x);
            f2 r23 = pair(p[L::Q + 2], p[L::Q + 3]) * bc(d01.x);
            r01 = __builtin_elementwise_fma(pair(p[L::Q + 4], p[L::Q + 5]), bc(d01.y), r01);
            r23 = __builtin_elementwise_fma(pair(p[L::Q + 6], p[L::Q + 7]), bc(d01.y), r23);
            r01 = __builtin_elementwise_fma(pair(p[L::Q + 8], p[L::Q + 9]), bc(d23.x), r01);
            r23 = __builtin_elementwise_fma(pair(p[L::Q + 10], p[L::Q + 11]), bc(d23.x), r23);
            r01 = __builtin_elementwise_fma(pair(p[L::Q + 12], p[L::Q + 13]), bc(d23.y), r01);
            r23 = __builtin_elementwise_fma(pair(p[L::Q + 14], p[L::Q + 15]), bc(d23.y), r23);
            const f2 cc = __builtin_elementwise_fma(r23, d23, r01 * d01);
            const T cx = cc.x + cc.y;
            T cu = T(0);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NU; ++j) r += p[L::R + i * NU + j] * u[j];
                cu += u[i] * r;
            }
            return (T(0.5) * cx + T(0.5) * cu) * dt;
        } else {
            T dx[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) dx[i] = x[i] - p[L::XT + i];
            T cx = T(0), cu = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) r += p[L::Q + i * NX + j] * dx[j];
                cx += dx[i] * r;
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NU; ++j) r += p[L::R + i * NU + j] * u[j];
                cu += u[i] * r;
            }
            return (T(0.5) * cx + T(0.5) * cu) * dt;
        }
    }
    // l_f(x) = 0.5 dx'Q_f dx   (not scaled by dt)
    template <typename P> static ILQR_DEV T terminal(P p, const T* x) {
        if constexpr (CUSTOM) {
            return Dyn::lf(x);
        } else {
            T dx[NX], c = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) dx[i] = x[i] - p[L::XT + i];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) r += p[L::QF + i * NX + j] * dx[j];
                c += dx[i] * r;
            }
            return T(0.5) * c;
        }
    }
    // gradient of the stage cost: l_x (n_x), l_u (n_u)
    template <typename P> static ILQR_DEV void grad(P p, T dt, const T* x, const T* u, T* lx, T* lu) {
        if constexpr (CUSTOM) {
            T lxx[NX][NX], lux[NU][NX], luu[NU][NU];
            Dyn::l_derivs(x, u, lx, lu, lxx, lux, luu);   // inlined: the unused second derivatives fold away
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) r += p[L::QS + i * NX + j] * (x[j] - p[L::XT + j]);
                lx[i] = r * dt;
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NU; ++j) r += p[L::RS + i * NU + j] * u[j];
                lu[i] = r * dt;
            }
        }
    }
    // second derivatives of the stage cost: l_xx (n_x x n_x), l_ux (n_u x n_x, system_base.py:216), l_uu
    template <typename P> static ILQR_DEV void hess(P p, T dt, const T* x, const T* u, T (*lxx)[NX], T (*lux)[NX],
                              T (*luu)[NU]) {
        if constexpr (CUSTOM) {
            T lx[NX], lu[NU];
            Dyn::l_derivs(x, u, lx, lu, lxx, lux, luu);
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NX; ++j) lxx[i][j] = p[L::QS + i * NX + j] * dt;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
#pragma unroll
                for (int j = 0; j < NX; ++j) lux[i][j] = T(0);
#pragma unroll
                for (int j = 0; j < NU; ++j) luu[i][j] = p[L::RS + i * NU + j] * dt;
            }
        }
    }
    template <typename P> static ILQR_DEV void l_f_x(P p, const T* x, T* out) {
        if constexpr (CUSTOM) {
            T H[NX][NX];
            Dyn::lf_derivs(x, out, H);
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                T r = T(0);
#pragma unroll
                for (int j = 0; j < NX; ++j) r += p[L::QFS + i * NX + j] * (x[j] - p[L::XT + j]);
                out[i] = r;
            }
        }
    }
    template <typename P> static ILQR_DEV void l_f_xx(P p, const T* x, T (*H)[NX]) {
        if constexpr (CUSTOM) {
            T g[NX];
            Dyn::lf_derivs(x, g, H);
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NX; ++j) H[i][j] = p[L::QFS + i * NX + j];
        }
    }
};

}  // namespace ilqr

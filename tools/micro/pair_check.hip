// pair_check: the system / integrator / cost templates instantiated on the float pair (two points per lane) against two
// scalar evaluations, element by element (expected: bit-identical).   hipcc -O3 --offload-arch=gfx950 -ffp-contract=on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../iterative-linear-quadratic-regulator_amd/csrc/dynamics.hpp"
using namespace ilqr;
constexpr int NP = 77;
template <int INTEG>
__global__ void k(const float* params, const float* X, float* out_s, float* out_p, float dt) {
    using D1 = DoublePendulum<float, 1>;
    using D2 = DoublePendulum<pair_f32, 1>;
    const int i = threadIdx.x;
    float pl[NP];
    for (int q = 0; q < NP; ++q) pl[q] = params[q];
    float* os = out_s + (size_t)i * 2 * 64;
    float* op = out_p + (size_t)i * 2 * 64;
    pair_f32 x2[4], u2[1];
    for (int h = 0; h < 2; ++h) {
        float x[4], u[1], xn[4], fx[4][4], fu[4][1], gx[4], gu[1], lxx[4][4], lux[1][4], luu[1][1];
        for (int q = 0; q < 4; ++q) x[q] = X[(i * 2 + h) * 8 + q];
        u[0] = X[(i * 2 + h) * 8 + 4];
        for (int q = 0; q < 4; ++q) { if (h == 0) x2[q].x = x[q]; else x2[q].y = x[q]; }
        if (h == 0) u2[0].x = u[0]; else u2[0].y = u[0];
        Stepper<float, D1>::step_jac(INTEG, pl, dt, x, u, xn, fx, fu);
        Cost<float, D1>::grad(pl, dt, x, u, gx, gu);
        Cost<float, D1>::hess(pl, dt, x, u, lxx, lux, luu);
        float* o = os + h * 64;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { o[a * 4 + b] = fx[a][b]; o[16 + a * 4 + b] = lxx[a][b]; }
        for (int a = 0; a < 4; ++a) { o[32 + a] = fu[a][0]; o[36 + a] = gx[a]; o[40 + a] = xn[a]; o[44 + a] = lux[0][a]; }
        o[48] = gu[0]; o[49] = luu[0][0];
    }
    pair_f32 xn[4], fx[4][4], fu[4][1], gx[4], gu[1], lxx[4][4], lux[1][4], luu[1][1];
    const SplatParams p{pl};
    const pair_f32 dt2 = dt;
    Stepper<pair_f32, D2>::step_jac(INTEG, p, dt2, x2, u2, xn, fx, fu);
    Cost<pair_f32, D2>::grad(p, dt2, x2, u2, gx, gu);
    Cost<pair_f32, D2>::hess(p, dt2, x2, u2, lxx, lux, luu);
    for (int h = 0; h < 2; ++h) {
        float* o = op + h * 64;
        auto pk = [&](pair_f32 v) { return h == 0 ? v.x : v.y; };
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { o[a * 4 + b] = pk(fx[a][b]); o[16 + a * 4 + b] = pk(lxx[a][b]); }
        for (int a = 0; a < 4; ++a) { o[32 + a] = pk(fu[a][0]); o[36 + a] = pk(gx[a]); o[40 + a] = pk(xn[a]); o[44 + a] = pk(lux[0][a]); }
        o[48] = pk(gu[0]); o[49] = pk(luu[0][0]);
    }
}
int main() {
    float hp[NP], hx[64 * 2 * 8];
    srand(1);
    for (int q = 0; q < NP; ++q) hp[q] = 0.2f + (rand() % 1000) / 500.0f;
    for (int q = 0; q < 64 * 16; ++q) hx[q] = (rand() % 2000) / 500.0f - 2.0f;
    float *dp, *dx, *ds, *dq;
    hipMalloc(&dp, sizeof hp); hipMalloc(&dx, sizeof hx); hipMalloc(&ds, 64 * 128 * 4); hipMalloc(&dq, 64 * 128 * 4);
    hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice); hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    static float hs[64 * 128], hq[64 * 128];
    const char* names[5] = {"euler", "midpoint", "rk4", "(backward_euler: n/a)", "discrete"};
    int bad_total = 0;
    for (int integ : {0, 1, 2, 4}) {
        hipMemset(ds, 0, sizeof hs); hipMemset(dq, 0, sizeof hq);
        if (integ == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dp, dx, ds, dq, 0.01f);
        if (integ == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dp, dx, ds, dq, 0.01f);
        if (integ == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dp, dx, ds, dq, 0.01f);
        if (integ == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, dp, dx, ds, dq, 0.01f);
        hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost); hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        double worst = 0;
        for (int q = 0; q < 64 * 128; ++q)
            if (memcmp(&hs[q], &hq[q], 4)) { ++bad; if (first < 0) first = q; double d = fabs((double)hs[q] - hq[q]); if (d > worst) worst = d; }
        printf("%-10s: %d of %d values differ (first at lane %d half %d entry %d: %g vs %g; max |d| %g)\n", names[integ], bad, 64 * 100,
               first / 128, (first % 128) / 64, first % 64, first >= 0 ? hs[first] : 0.f, first >= 0 ? hq[first] : 0.f, worst);
        bad_total += bad;
    }
    return bad_total != 0;
}

// ilqr_abi.cpp -- extern "C" entry points of libilqr_hip.so (include/ilqr_hip.h).
// Argument validation lives here so that the errors the reference raises as Python
// exceptions (bad U_init shape, iLQR_class.py:50-52; unknown integrator,
// system_base.py:198) come back as ILQR_ERR_INVALID_ARG before anything touches the GPU.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <string>

#include "solver.hpp"

namespace ilqr {

int n_sys_params_abi(int system, int n_x, int n_u) {
    switch (system) {
        case ILQR_SYS_PENDULUM: return (n_x == 2 && n_u == 1) ? 3 : -1;
        case ILQR_SYS_UA_DOUBLE_PENDULUM: return (n_x == 4 && n_u == 1) ? 9 : -1;
        case ILQR_SYS_DOUBLE_PENDULUM: return (n_x == 4 && n_u == 2) ? 9 : -1;
        case ILQR_SYS_LINEAR: return (n_x >= 1 && n_u >= 1 && n_x <= 64 && n_u <= 64) ? n_x * n_x + n_x * n_u : -1;
        case ILQR_SYS_CUSTOM: return (n_x >= 1 && n_u >= 1 && n_x <= 6 && n_u <= n_x) ? 0 : -1;
        default: return -1;
    }
}

int param_count(int system, int n_x, int n_u) {
    const int ns = n_sys_params_abi(system, n_x, n_u);
    if (ns < 0) return -1;
    return ns + n_x + n_x * n_x + n_u * n_u + n_x * n_x;
}

}  // namespace ilqr

using ilqr::SolverBase;

struct ilqr_solver_s {
    SolverBase* impl;
    void* plugin;  // dlopen handle of a user-system plugin, or nullptr
};

static thread_local std::string g_create_error;

static int fail_create(int code, const std::string& msg) {
    g_create_error = msg;
    return code;
}

extern "C" {

int ilqr_abi_version(void) { return ILQR_ABI_VERSION; }

int ilqr_device_count(int* count) {
    if (!count) return ILQR_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return ILQR_ERR_NO_DEVICE; }
    *count = n;
    return ILQR_OK;
}

int ilqr_param_count(int system, int n_x, int n_u) { return ilqr::param_count(system, n_x, n_u); }

int ilqr_is_supported(int system, int n_x, int n_u, int dtype) {
    if (dtype == ILQR_F32) return ilqr::supported_f32(system, n_x, n_u) ? 1 : 0;
    if (dtype == ILQR_F64) return ilqr::supported_f64(system, n_x, n_u) ? 1 : 0;
    return 0;
}

const char* ilqr_last_error(ilqr_handle h) {
    if (!h || !h->impl) return g_create_error.c_str();
    return h->impl->err.c_str();
}

static int create_impl(ilqr_handle* out, const ilqr_config* cfg, const char* plugin_path) {
    if (!out) return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: out is NULL");
    *out = nullptr;
    if (!cfg) return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: cfg is NULL");
    if (cfg->struct_size != sizeof(ilqr_config))
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: struct_size does not match this library's ilqr_config");
    if (cfg->n_x < 1 || cfg->n_u < 1 || cfg->horizon < 1 || cfg->batch < 1)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: n_x, n_u, horizon and batch must be >= 1");
    if (cfg->n_alpha < 1 || cfg->n_alpha > ilqr::kMaxAlpha)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: n_alpha must be in [1, 16]");
    if (cfg->n_trials < 1 || cfg->n_trials > 64)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: n_trials must be in [1, 64]");
    if (cfg->maxiter < 0) return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: maxiter must be >= 0");
    if (cfg->dtype != ILQR_F32 && cfg->dtype != ILQR_F64)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: unknown dtype");
    if (cfg->integrator < ILQR_INT_EULER || cfg->integrator > ILQR_INT_DISCRETE)
        return fail_create(ILQR_ERR_INVALID_ARG,
                           "Unknown integrator. Supported: 'rk4', 'midpoint', 'euler', 'backward_euler'.");
    if (cfg->plant_integrator < -1 || cfg->plant_integrator > ILQR_INT_DISCRETE)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: unknown plant integrator");
    if (!(cfg->dt > 0.0)) return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: dt must be > 0");
    if (!(cfg->alpha_factor > 0.0 && cfg->alpha_factor < 1.0))
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: alpha_factor must be in (0, 1)");
    if (cfg->mu < 0.0) return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: mu must be >= 0");
    const int want = ilqr::param_count(cfg->system, cfg->n_x, cfg->n_u);
    if (want < 0)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: unknown system or n_x/n_u do not match the system");
    if (!cfg->params || cfg->n_params != want)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: params is NULL or n_params != ilqr_param_count()");
    if ((cfg->system == ILQR_SYS_CUSTOM) != (plugin_path != nullptr))
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: ILQR_SYS_CUSTOM goes through ilqr_create_custom (and only it)");
    if (!plugin_path && !ilqr_is_supported(cfg->system, cfg->n_x, cfg->n_u, cfg->dtype))
        return fail_create(ILQR_ERR_UNSUPPORTED, "ilqr_create: no kernels compiled for this (system, n_x, n_u, dtype)");
    if ((cfg->integrator == ILQR_INT_MIDPOINT || cfg->integrator == ILQR_INT_RK4 ||
         cfg->integrator == ILQR_INT_BACKWARD_EULER || cfg->plant_integrator == ILQR_INT_MIDPOINT ||
         cfg->plant_integrator == ILQR_INT_RK4 || cfg->plant_integrator == ILQR_INT_BACKWARD_EULER) &&
        cfg->n_x > 4 && !plugin_path)
        return fail_create(ILQR_ERR_UNSUPPORTED, "ilqr_create: n_x > 4 supports the 'euler' and 'discrete' integrators only");

    // the product path has no CPU fallback: a gfx950 device is mandatory
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail_create(ILQR_ERR_NO_DEVICE, "ilqr_create: no HIP device visible (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail_create(ILQR_ERR_HIP, "ilqr_create: hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail_create(ILQR_ERR_NO_DEVICE,
                           std::string("ilqr_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    std::string err;
    int status = ILQR_OK;
    SolverBase* impl = nullptr;
    void* plugin = nullptr;
    if (plugin_path) {
        // user-defined system: the plugin holds the kernels instantiated for the generated dynamics
        plugin = dlopen(plugin_path, RTLD_NOW | RTLD_LOCAL);
        if (!plugin) return fail_create(ILQR_ERR_INVALID_ARG, std::string("ilqr_create_custom: dlopen failed: ") + dlerror());
        typedef int (*info_fn)(int*, int*, int*);
        typedef SolverBase* (*make_fn)(const ilqr_config*, char*, int, int*);
        info_fn info = (info_fn)dlsym(plugin, "ilqr_plugin_info");
        make_fn make = (make_fn)dlsym(plugin, "ilqr_plugin_make_solver");
        int abi = 0, pnx = 0, pnu = 0;
        if (!info || !make || info(&abi, &pnx, &pnu) != 0 || abi != ILQR_ABI_VERSION) {
            dlclose(plugin);
            return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create_custom: not an ilqr system plugin of this ABI version");
        }
        if (pnx != cfg->n_x || pnu != cfg->n_u) {
            dlclose(plugin);
            return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create_custom: n_x / n_u do not match the plugin's system");
        }
        char msg[512] = {0};
        impl = make(cfg, msg, (int)sizeof(msg), &status);
        if (!impl) {
            dlclose(plugin);
            return fail_create(status ? status : ILQR_ERR_HIP, msg);
        }
    } else {
        impl = (cfg->dtype == ILQR_F32) ? ilqr::make_solver_f32(*cfg, err, &status) : ilqr::make_solver_f64(*cfg, err, &status);
        if (!impl) return fail_create(status ? status : ILQR_ERR_HIP, err);
    }
    impl->cfg.params = nullptr;  // never retain the caller's host pointer
    ilqr_handle h = new ilqr_solver_s{impl, plugin};
    *out = h;
    return ILQR_OK;
}

int ilqr_create(ilqr_handle* out, const ilqr_config* cfg) { return create_impl(out, cfg, nullptr); }

int ilqr_create_custom(ilqr_handle* out, const ilqr_config* cfg, const char* plugin_path) {
    if (!plugin_path) {
        if (out) *out = nullptr;
        return fail_create(ILQR_ERR_INVALID_ARG, "ilqr_create_custom: plugin_path is NULL");
    }
    return create_impl(out, cfg, plugin_path);
}



int ilqr_destroy(ilqr_handle h) {
    if (!h) return ILQR_OK;
    delete h->impl;  // its code lives in the plugin: destroy before unloading
    if (h->plugin) dlclose(h->plugin);
    delete h;
    return ILQR_OK;
}

#define ILQR_FWD(h, call)                                 \
    do {                                                  \
        if (!(h) || !(h)->impl) return ILQR_ERR_INVALID_ARG; \
        return (h)->impl->call;                           \
    } while (0)

int ilqr_sync(ilqr_handle h) { ILQR_FWD(h, sync()); }
int ilqr_set_problem(ilqr_handle h, const void* x0, const void* U_init) { ILQR_FWD(h, set_problem(x0, U_init)); }
int ilqr_set(ilqr_handle h, int field, const void* src, size_t bytes) { ILQR_FWD(h, set(field, src, bytes)); }
int ilqr_get(ilqr_handle h, int field, void* dst, size_t bytes) { ILQR_FWD(h, get(field, dst, bytes)); }
int ilqr_initial_rollout(ilqr_handle h) { ILQR_FWD(h, initial_rollout()); }
int ilqr_linearize(ilqr_handle h) { ILQR_FWD(h, linearize()); }
int ilqr_backward(ilqr_handle h) { ILQR_FWD(h, backward()); }
int ilqr_forward(ilqr_handle h, const double* alphas, int n) { ILQR_FWD(h, forward(alphas, n)); }
int ilqr_select(ilqr_handle h) { ILQR_FWD(h, select()); }
int ilqr_iterate(ilqr_handle h, int n_iters) { ILQR_FWD(h, iterate(n_iters)); }
int ilqr_flush(ilqr_handle h) { ILQR_FWD(h, flush()); }
int ilqr_solve(ilqr_handle h, int32_t* iters_out, void* cost_out) { ILQR_FWD(h, solve(iters_out, cost_out)); }
int ilqr_backward_pass(ilqr_handle h, const void* X, const void* U, void* U_ff_out, void* K_out) {
    ILQR_FWD(h, backward_pass(X, U, U_ff_out, K_out));
}

int ilqr_backward_tensors(ilqr_handle h, const void* lin, const void* term, void* U_ff_out, void* K_out) {
    ILQR_FWD(h, backward_tensors(lin, term, U_ff_out, K_out));
}
int ilqr_forward_pass(ilqr_handle h, const void* x0, double alpha, const void* X_old, const void* U_old,
                      const void* U_ff, const void* K, void* X_new, void* U_new, void* cost) {
    ILQR_FWD(h, forward_pass(x0, alpha, X_old, U_old, U_ff, K, X_new, U_new, cost));
}
int ilqr_eval_points(ilqr_handle h, int integrator, int npts, const void* x, const void* u, void* f, void* f_x,
                     void* f_u, void* l, void* l_x, void* l_u, void* l_xx, void* l_ux, void* l_uu, void* l_f,
                     void* l_f_x, void* l_f_xx) {
    if (!h || !h->impl) return ILQR_ERR_INVALID_ARG;
    if (integrator > ILQR_INT_DISCRETE) { h->impl->err = "eval_points: unknown integrator"; return ILQR_ERR_INVALID_ARG; }
    if (integrator >= 0 && integrator != ILQR_INT_EULER && integrator != ILQR_INT_DISCRETE && h->impl->cfg.n_x > 4) {
        h->impl->err = "eval_points: n_x > 4 supports the 'euler' and 'discrete' integrators only";
        return ILQR_ERR_UNSUPPORTED;
    }
    void* outs[12] = {f, f_x, f_u, l, l_x, l_u, l_xx, l_ux, l_uu, l_f, l_f_x, l_f_xx};
    return h->impl->eval_points(integrator, npts, x, u, outs);
}
int ilqr_mpc_reset(ilqr_handle h, const void* x0, const void* U_init) { ILQR_FWD(h, mpc_reset(x0, U_init)); }
int ilqr_mpc_rearm(ilqr_handle h, const void* x0, const void* U_init) { ILQR_FWD(h, mpc_rearm(x0, U_init)); }
int ilqr_mpc_run(ilqr_handle h, int n_steps, void* u_out, void* x_out, void* cost_out) {
    ILQR_FWD(h, mpc_run(n_steps, u_out, x_out, cost_out));
}
int ilqr_status_reduce(ilqr_handle h, void* dev_out4) { ILQR_FWD(h, status_reduce(dev_out4)); }
/* diagnostic (not in the public header): raw clock-probe buffer, valid when ILQR_CLOCK_PROBE was set */
int ilqr_debug_probe_dump(ilqr_handle h, long long* dst, size_t n) {
    if (!h || !h->impl) return ILQR_ERR_INVALID_ARG;
    return h->impl->probe_dump(dst, n);
}
/* diagnostic (not in the public header): retarget the handle's launches to another stream (tools/cumask_probe.py) */
int ilqr_debug_set_stream(ilqr_handle h, void* stream) {
    if (!h || !h->impl) return ILQR_ERR_INVALID_ARG;
    return h->impl->debug_set_stream(stream);
}
int ilqr_timing_enable(ilqr_handle h, int on) { ILQR_FWD(h, timing_enable(on)); }
int ilqr_timing_reset(ilqr_handle h) { ILQR_FWD(h, timing_reset()); }
int ilqr_timing_get(ilqr_handle h, double ms[ILQR_N_PHASES], int64_t launches[ILQR_N_PHASES]) {
    ILQR_FWD(h, timing_get(ms, launches));
}
int ilqr_algorithmic_bytes(ilqr_handle h, double bytes[ILQR_N_PHASES]) {
    if (!h || !h->impl || !bytes) return ILQR_ERR_INVALID_ARG;
    return h->impl->algorithmic_bytes(bytes);
}

}  // extern "C"

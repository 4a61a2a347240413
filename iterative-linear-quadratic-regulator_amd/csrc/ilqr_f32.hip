// float instantiation of the solver and its kernels (one translation unit per dtype so the
// two compile in parallel).
#include "solver.hpp"
#define ILQR_T float
#include "instantiate.inc"
namespace ilqr {
SolverBase* make_solver_f32(const ilqr_config& cfg, std::string& err, int* status) {
    auto* s = new SolverT<float>();
    const int rc = s->init(cfg);
    if (rc) { err = s->err; *status = rc; delete s; return nullptr; }
    return s;
}
bool supported_f32(int system, int n_x, int n_u) { Ops<float> o; return find_ops<float>(system, n_x, n_u, &o); }
}  // namespace ilqr

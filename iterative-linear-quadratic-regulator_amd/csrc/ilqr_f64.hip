// double instantiation of the solver and its kernels.
#include "solver.hpp"
#define ILQR_T double
#include "instantiate.inc"
namespace ilqr {
SolverBase* make_solver_f64(const ilqr_config& cfg, std::string& err, int* status) {
    auto* s = new SolverT<double>();
    const int rc = s->init(cfg);
    if (rc) { err = s->err; *status = rc; delete s; return nullptr; }
    return s;
}
bool supported_f64(int system, int n_x, int n_u) { Ops<double> o; return find_ops<double>(system, n_x, n_u, &o); }
}  // namespace ilqr

// robotoc::OCPSolver (robotoc_amd/host/robotoc_hip_solver.hpp) over the device-side linearisation
// (robotoc_hip_device_source.hpp) on a contact sequence with lifts and touch-downs and the Constraints object of
// examples/anymal/trot.cpp:131-146 (six joint-limit components + FrictionCone): OCPSolver::solve, nothing of the
// iteration on the host.   usage: ocp_solver_trot_test <problem.bin> <out.bin>
//   problem.bin (tests/test_cpp_solver.py writes it): rtoc_robot_model, rtoc_configuration_cost, int n, rtoc_grid[n],
//   unsigned mask[n], double positions[n][ncontacts][3], double q0[nq], v0[nv], double f_init[n][max_dimf],
//   double q_max, v_max, u_max, mu, barrier
#include <cstdio>
#include <vector>

#include "../../robotoc_amd/host/robotoc_hip_device_source.hpp"

using namespace robotoc;

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  rtoc_robot_model model;
  rtoc_configuration_cost cost;
  int n = 0;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  if (!rd(f, &model, 1) || !rd(f, &cost, 1) || !rd(f, &n, 1) || n < 2) return 4;
  const int nv = model.nv, nq = model.nq, nc = model.ncontacts, nu = nv - 6, dimf = 3 * nc;
  std::vector<rtoc_grid> g(n);
  std::vector<unsigned> mask(n);
  std::vector<double> pos((size_t)n * nc * 3), q0(nq), v0(nv), finit((size_t)n * dimf);
  double lim[5];
  const bool ok = rd(f, g.data(), n) && rd(f, mask.data(), n) && rd(f, pos.data(), pos.size()) && rd(f, q0.data(), nq) && rd(f, v0.data(), nv) &&
                  rd(f, finit.data(), finit.size()) && rd(f, lim, 5);
  std::fclose(f);
  if (!ok) return 4;
  try {
    std::vector<GridInfo> grid(n);
    for (int i = 0; i < n; ++i) {
      grid[i].type = static_cast<GridType>(g[i].type);
      grid[i].dt = g[i].dt;
      grid[i].switching_constraint = g[i].switching_constraint != 0;
      grid[i].dimf = g[i].dimf, grid[i].dims = g[i].dims;
      grid[i].num_grids_in_phase = g[i].num_grids_in_phase;
      grid[i].stage = g[i].time_stage < 0 ? 0 : g[i].time_stage;
    }
    RobotDims dims;
    dims.dimv = nv, dims.dimu = nu, dims.dim_passive = 6, dims.max_dimf = dimf;
    Solution s0(n, SplitSolution(dims));
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < nq; ++k) s0[i].q(k) = q0[k];
      for (int k = 0; k < dimf; ++k) s0[i].f_full(k) = finit[(size_t)i * dimf + k];
    }
    auto source = std::make_shared<ConfigurationCostSource>(model, cost, grid, mask, pos, s0);
    const std::vector<double> qmax(nu, lim[0]), qmin(nu, -lim[0]), vmax(nu, lim[1]), umax(nu, lim[2]);
    source->setJointLimits(qmin, qmax, vmax, umax);
    source->setFrictionCone(std::vector<double>(nc, lim[3]), false);
    source->setBarrierParam(lim[4], 0.995);
    SolverOCP ocp(source);
    SolverOptions opt;
    opt.max_iter = 150;
    opt.kkt_tol = 1.0e-8;
    OCPSolver solver(ocp, opt);
    Vec q(nq), v(nv);
    for (int k = 0; k < nq; ++k) q(k) = q0[k];
    for (int k = 0; k < nv; ++k) v(k) = v0[k];
    solver.solve(0.0, q, v, true);
    const SolverStatistics& st = solver.getSolverStatistics();
    std::printf("OCPSolver::solve, constrained trot on the device: KKT error %.3e -> %.3e in %d iterations, converged %d\n",
                std::sqrt(st.performance_index.front()), solver.KKTError(), st.iter, (int)st.convergence);
    if (solver.status() != 0) return 5;
    const Solution& s = solver.getSolution();
    std::vector<double> out;
    out.push_back(st.iter), out.push_back(st.convergence ? 1.0 : 0.0), out.push_back(solver.KKTError()), out.push_back(std::sqrt(st.performance_index.front()));
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < nq; ++k) out.push_back(s[i].q(k));
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < nu; ++k) out.push_back(s[i].u(k));
    f = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), f);
    std::fclose(f);
    return st.convergence ? 0 : 6;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 7;
  }
}

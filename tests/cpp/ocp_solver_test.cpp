// C++ host-side test of the robotoc::OCPSolver / DirectMultipleShooting shells
// (robotoc_amd/host/robotoc_hip_solver.hpp) on the GPU:  ocp_solver_test <stage dump> <output file>
// runs OCPSolver::updateSolution (the reference's ocp_solver.cpp:111-145 call sequence) on the recorded stage data and
// writes what it computed; tests/test_cpp_solver.py compares that with the CPU oracle's sequence of the same
// iteration.  Also checked here: value semantics (a copy of the solver computes the same iteration on its own device
// context), solve() with the convergence test, argument validation like the reference's solver layer.
#include <cmath>
#include <cstdio>
#include <string>

#include "../../robotoc_amd/host/robotoc_hip_solver.hpp"

using namespace robotoc;

static int fail(const char* what) {
  std::fprintf(stderr, "FAILED: %s\n", what);
  return 1;
}

static void put(FILE* f, const Vec& v) { std::fwrite(v.data(), sizeof(double), v.size(), f); }

int main(int argc, char** argv) {
  if (argc < 3) return fail("usage: ocp_solver_test <stage dump> <output file>");
  if (rtoc_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  auto source = std::make_shared<StageDumpSource>(argv[1]);
  SolverOCP ocp(source);
  SolverOptions options;
  options.max_iter = 3;
  options.kkt_tol = 1.0e-07;
  OCPSolver solver(ocp, options);
  const RobotDims robot = source->robot();
  Vec q(robot.dimv + (robot.dim_passive > 0 ? 1 : 0)), v(robot.dimv);
  const double t = 0.0;
  solver.discretize(t);
  solver.initConstraints();
  OCPSolver copy(solver);  // deep copy of the device context before the iteration
  const Solution s_before = solver.getSolution();
  solver.updateSolution(t, q, v);
  if (solver.status() != 0) return fail("numerical status bits set");
  const double kkt_error = solver.KKTError();
  const SolverStatistics& st = solver.getSolverStatistics();
  if (st.primal_step_size.size() != 1 || st.dual_step_size.size() != 1) return fail("step sizes not recorded");
  const Direction& d = solver.getDirection();
  const Solution& s = solver.getSolution();
  const std::vector<LQRPolicy>& lqr = solver.getLQRPolicy();
  const RiccatiFactorization& ric = solver.getRiccatiFactorization();
  const int n = solver.getTimeDiscretization().size();
  FILE* f = std::fopen(argv[2], "wb");
  if (!f) return fail("cannot open the output file");
  const double head[4] = {kkt_error, st.primal_step_size[0], st.dual_step_size[0], static_cast<double>(n)};
  std::fwrite(head, sizeof(double), 4, f);
  for (int i = 0; i < n; ++i) {
    put(f, d[i].dx);
    put(f, d[i].du);
    put(f, d[i].dlmdgmm);
    put(f, d[i].daf_full);
    put(f, d[i].dbetamu_full);
    put(f, s[i].q);
    put(f, s[i].v);
    put(f, s[i].a);
    put(f, s[i].u);
    put(f, s[i].lmd);
    put(f, s[i].gmm);
    put(f, ric[i].s);
    put(f, lqr[i].k);
  }
  std::fclose(f);
  // the iterate moved
  double moved = 0.0;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < robot.dimv; ++k) moved += std::fabs(s[i].v(k) - s_before[i].v(k));
  if (!(moved > 0.0)) return fail("updateSolution left the iterate unchanged");
  // value semantics: the copy runs the same iteration on its own context and gets the same numbers
  copy.updateSolution(t, q, v);
  const Direction& dc = copy.getDirection();
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 2 * robot.dimv; ++k)
      if (dc[i].dx(k) != d[i].dx(k)) return fail("copy of the solver computed a different direction");
  if (copy.context() == solver.context()) return fail("copy shares the device context");
  // solve(): max_iter iterations on the replayed linearisation, statistics filled, no convergence claim on it
  OCPSolver assigned;
  assigned = copy;
  assigned.solve(t, q, v, true);
  if (assigned.getSolverStatistics().iter != options.max_iter) return fail("solve() did not run max_iter iterations");
  if (assigned.getSolverStatistics().performance_index.size() != static_cast<size_t>(options.max_iter)) return fail("statistics");
  // a tolerance above the KKT error converges in one iteration
  SolverOptions loose = options;
  loose.kkt_tol = 2.0 * kkt_error;
  assigned.setSolverOptions(loose);
  assigned.solve(t, q, v, true);
  if (!assigned.getSolverStatistics().convergence || assigned.getSolverStatistics().iter != 1) return fail("convergence test");
  // argument validation like ocp_solver.cpp:150-155
  bool threw = false;
  try {
    Vec bad(3);
    assigned.solve(t, bad, v);
  } catch (const std::out_of_range&) {
    threw = true;
  }
  if (!threw) return fail("solve() accepted a q of the wrong size");
  threw = false;
  try {
    SolverOptions ls;
    ls.enable_line_search = true;
    assigned.setSolverOptions(ls);
  } catch (const std::logic_error&) {
    threw = true;
  }
  if (!threw) return fail("line search request was not rejected");
  // RiccatiRecursion: default-constructible and copyable like the reference's (riccati_recursion.hpp:40-60)
  RiccatiRecursion empty;
  threw = false;
  try {
    empty.status();
  } catch (const std::logic_error&) {
    threw = true;
  }
  if (!threw) return fail("default-constructed RiccatiRecursion must refuse work");
  std::printf("ocp_solver_test passed: KKT error %.6e, steps %.6f / %.6f, %d grid points\n", kkt_error, st.primal_step_size[0],
              st.dual_step_size[0], n);
  return 0;
}

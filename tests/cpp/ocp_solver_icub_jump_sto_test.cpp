// robotoc::OCPSolver (robotoc_amd/host/robotoc_hip_solver.hpp) on BASELINE configs[3] as the reference poses it
// (examples/icub/python/jump_sto.py): iCub on its two soles -- SURFACE contacts, contact placements with rotations --, two jumps, four
// STO-enabled events, ConfigurationSpaceCost, the joint limits of the URDF, FrictionCone, STOConstraints, T = 2.6 s, N = 130,
// kkt_tol_mesh = 0.1, max_dt_mesh = T / N, initial_sto_reg_iter = 10, max_iter = 350 -- described the reference's way (ContactSequence
// with init / push_back(..., sto = true), OCP{T, N}) and solved by OCPSolver::solve with nothing of the iteration on the host.
//   usage: ocp_solver_icub_jump_sto_test <problem.bin> <out.bin>
//   problem.bin (tests/test_cpp_solver.py writes it): rtoc_robot_model, rtoc_configuration_cost, int N, int nev, double T,
//   event times [nev], placements: positions [nev + 1][nc][3], rotations [nc][9], q0[nq], v0[nv], min_dwell[nev + 1],
//   q_min[nu], q_max[nu], v_max[nu], u_max[nu], mu, max_dt_mesh
//   out.bin: iterations, converged, KKT error, number of mesh refinements, first refinement iteration, event times [nev], KKT history
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
  int N = 0, nev = 0;
  double T = 0.0;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  if (!rd(f, &model, 1) || !rd(f, &cost, 1) || !rd(f, &N, 1) || !rd(f, &nev, 1) || !rd(f, &T, 1)) return 4;
  const int nv = model.nv, nq = model.nq, nc = model.ncontacts, nu = nv - 6, dimf = 6 * nc;
  std::vector<double> times(nev), pos((nev + 1) * nc * 3), rot(nc * 9), q0(nq), v0(nv), min_dwell(nev + 1), qmin(nu), qmax(nu), vmax(nu), umax(nu);
  double mu = 0.0, max_dt_mesh = 0.0;
  const bool ok = rd(f, times.data(), nev) && rd(f, pos.data(), pos.size()) && rd(f, rot.data(), rot.size()) && rd(f, q0.data(), nq) &&
                  rd(f, v0.data(), nv) && rd(f, min_dwell.data(), nev + 1) && rd(f, qmin.data(), nu) && rd(f, qmax.data(), nu) &&
                  rd(f, vmax.data(), nu) && rd(f, umax.data(), nu) && rd(f, &mu, 1) && rd(f, &max_dt_mesh, 1);
  std::fclose(f);
  if (!ok) return 4;
  try {
    const unsigned all = (1u << nc) - 1u;
    auto phase_pos = [&](int p) { return std::vector<double>(pos.begin() + (size_t)p * nc * 3, pos.begin() + (size_t)(p + 1) * nc * 3); };
    ContactSequence cs(std::vector<int>(nc, 6));
    cs.init(all, phase_pos(0), rot);                                     // contact_sequence.init(contact_status_standing)
    for (int e = 0; e < nev; ++e)                                        // push_back(contact_status_flying | _standing, t, sto=True)
      cs.push_back(e % 2 == 0 ? 0u : all, phase_pos(e + 1), times[e], true, rot);
    if (cs.numLiftEvents() != nev / 2 || cs.numImpactEvents() != nev / 2) return 8;
    const TimeDiscretization td0 = discretize(cs, T, N, 0.0, true);
    const int n = td0.size();
    RobotDims dims;
    dims.dimv = nv, dims.dimu = nu, dims.dim_passive = 6, dims.max_dimf = dimf;
    Solution s0(n, SplitSolution(dims));
    for (int i = 0; i < n; ++i) {   // the example's initial guess: the standing pose on every grid point
      for (int k = 0; k < nq; ++k) s0[i].q(k) = q0[k];
      for (int k = 0; k < nv; ++k) s0[i].v(k) = v0[k];
    }
    auto sto = std::make_shared<STOConstraints>(min_dwell, 1.0e-3, 0.995);
    auto source = std::make_shared<ConfigurationCostSource>(model, cost, cs, T, N, s0, sto);
    source->setJointLimits(qmin, qmax, vmax, umax);
    source->setFrictionCone(std::vector<double>(nc, mu), false);
    source->setBarrierParam(1.0e-3, 0.995);
    SolverOCP ocp(source);
    SolverOptions opt;
    opt.max_iter = 350;
    opt.kkt_tol = 1.0e-7;
    opt.kkt_tol_mesh = 0.1;
    opt.max_dt_mesh = max_dt_mesh;
    opt.initial_sto_reg_iter = 10;
    OCPSolver solver(ocp, opt);
    Vec q(nq), v(nv);
    for (int k = 0; k < nq; ++k) q(k) = q0[k];
    for (int k = 0; k < nv; ++k) v(k) = v0[k];
    solver.solve(0.0, q, v, true);
    const SolverStatistics& st = solver.getSolverStatistics();
    std::printf("OCPSolver::solve, iCub jump_sto example on the device: KKT error %.3e -> %.3e in %d iterations, converged %d, mesh refinements %d, "
                "event times", std::sqrt(st.performance_index.front()), solver.KKTError(), st.iter, (int)st.convergence, (int)st.mesh_refinement_iter.size());
    for (int e = 0; e < nev; ++e) std::printf(" %.6f", solver.eventTimes()[e]);
    std::printf("\n");
    if (solver.status() != 0) return 5;
    std::vector<double> out;
    out.push_back(st.iter), out.push_back(st.convergence ? 1.0 : 0.0), out.push_back(solver.KKTError());
    out.push_back((double)st.mesh_refinement_iter.size()), out.push_back(st.mesh_refinement_iter.empty() ? -1.0 : st.mesh_refinement_iter[0]);
    for (int e = 0; e < nev; ++e) out.push_back(solver.eventTimes()[e]);
    for (double e : st.performance_index) out.push_back(std::sqrt(e));
    f = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), f);
    std::fclose(f);
    return st.convergence ? 0 : 6;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 7;
  }
}

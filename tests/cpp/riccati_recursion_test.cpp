// C++ host-side test of robotoc::RiccatiRecursion (robotoc_amd/host/robotoc_hip.hpp) on the GPU,
// written in the style of the reference's own tests
// (test/riccati/riccati_factorizer_test.cpp:36-71, unconstr_riccati_recursion_test.cpp:61-106):
// random but structurally valid inputs (kkt_factory.cpp:7-36), optimised routine vs naive dense
// re-derivation, relative tolerance 1e-9.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

#include "../../robotoc_amd/host/robotoc_hip.hpp"

using namespace robotoc;

static std::mt19937_64 rng(20260925);
static double rnd() { return std::uniform_real_distribution<double>(-1.0, 1.0)(rng); }

static double relerr(const double* a, const double* b, int n) {
  double num = 0, den = 0;
  for (int i = 0; i < n; ++i) {
    num += (a[i] - b[i]) * (a[i] - b[i]);
    den += b[i] * b[i] > a[i] * a[i] ? b[i] * b[i] : a[i] * a[i];
  }
  return std::sqrt(num / (den > 1e-300 ? den : 1e-300));
}

int main(int argc, char** argv) {
  // `riccati_recursion_test scan`: the same closed-form checks with the horizon scan (setHorizonScan) at the
  // scan's tolerance (1e-8: P+ of a stage comes from the scan, P of the next record from its own stage)
  const bool scan = argc > 1 && std::string(argv[1]) == "scan";
  if (rtoc_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  const RobotDims robot = {18, 12, 6, 12};  // ANYmal
  OCP ocp;
  ocp.robot = robot;
  ocp.N = 10;
  ocp.reserved_num_discrete_events = 1;
  const int nv = robot.dimv, nu = robot.dimu, nx = 2 * nv, N = ocp.N;
  const double dt = 0.02;
  std::vector<GridInfo> grid(N + 1);
  for (int i = 0; i <= N; ++i) {
    grid[i].type = (i == N) ? GridType::Terminal : GridType::Intermediate;
    grid[i].dt = (i == N) ? 0.0 : dt;
    grid[i].stage = i;
    grid[i].num_grids_in_phase = N;
    grid[i].dimf = 12;
  }
  TimeDiscretization td(grid);
  KKTMatrix kkt_matrix(N + 1, SplitKKTMatrix(robot));
  KKTResidual kkt_residual(N + 1, SplitKKTResidual(robot));
  RiccatiFactorization factorization(N + 1, SplitRiccatiFactorization(robot));
  Direction d(N + 1, SplitDirection(robot));
  for (int i = 0; i <= N; ++i) {
    SplitKKTMatrix& m = kkt_matrix[i];
    SplitKKTResidual& r = kkt_residual[i];
    std::vector<double> seed((nx + nu) * (nx + nu));
    for (auto& v : seed) v = rnd();
    Mat H(nx + nu, nx + nu);
    for (int a = 0; a < nx + nu; ++a)
      for (int b = 0; b < nx + nu; ++b) {
        double acc = 0;
        for (int k = 0; k < nx + nu; ++k) acc += seed[a + k * (nx + nu)] * seed[b + k * (nx + nu)];
        H(a, b) = acc;
      }
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) m.Qxx(a, b) = H(a, b);
    for (int a = 0; a < nx; ++a) r.lx(a) = rnd();
    if (i == N) continue;
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nu; ++b) m.Qxu(a, b) = H(a, nx + b);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nu; ++b) m.Quu(a, b) = H(nx + a, nx + b);
    for (int a = 0; a < nv; ++a) {
      m.Fxx(a, a) = 1.0;
      m.Fxx(a, nv + a) = dt;
    }
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        m.Fxx(a, b) = rnd();
        m.Fxx(a, nv + b) = rnd();
      }
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nx; ++b) m.Fxx(nv + a, b) = rnd();
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nu; ++b) m.Fvu(a, b) = rnd();
    for (int a = 0; a < nx; ++a) r.Fx(a) = rnd();
    for (int a = 0; a < nu; ++a) r.lu(a) = rnd();
  }
  const KKTMatrix kkt_matrix_ref = kkt_matrix;
  const KKTResidual kkt_residual_ref = kkt_residual;

  RiccatiRecursion riccati_recursion(ocp);
  riccati_recursion.setHorizonScan(scan);
  riccati_recursion.backwardRiccatiRecursion(td, kkt_matrix, kkt_residual, factorization);
  if (riccati_recursion.status() != 0) {
    std::fprintf(stderr, "status %u\n", riccati_recursion.status());
    return 1;
  }
  for (int a = 0; a < nx; ++a) d[0].dx(a) = 0.1 * rnd();
  riccati_recursion.forwardRiccatiRecursion(td, kkt_matrix, kkt_residual, factorization, d);
  const std::vector<LQRPolicy>& lqr = riccati_recursion.getLQRPolicy();

  double worst = 0;
  // terminal: P_N = Qxx_N, s_N = -lx_N
  worst = std::fmax(worst, relerr(factorization[N].P.data(), kkt_matrix_ref[N].Qxx.data(), nx * nx));
  for (int i = N - 1; i >= 0; --i) {
    const SplitKKTMatrix& m0 = kkt_matrix_ref[i];
    const Mat& Pn = factorization[i + 1].P;
    const Vec& sn = factorization[i + 1].s;
    // naive dense F, H, G, lu with B = [0; Fvu]
    Mat A = m0.Fxx, B(nx, nu);
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nu; ++b) B(nv + a, b) = m0.Fvu(a, b);
    auto mul = [](const Mat& X, bool tx, const Mat& Y, bool ty) {
      const int M = tx ? X.cols() : X.rows(), K = tx ? X.rows() : X.cols(), Nn = ty ? Y.rows() : Y.cols();
      Mat Z(M, Nn);
      for (int a = 0; a < M; ++a)
        for (int b = 0; b < Nn; ++b) {
          double acc = 0;
          for (int k = 0; k < K; ++k) acc += (tx ? X(k, a) : X(a, k)) * (ty ? Y(b, k) : Y(k, b));
          Z(a, b) = acc;
        }
      return Z;
    };
    Mat PA = mul(Pn, false, A, false), PB = mul(Pn, false, B, false);
    Mat F = mul(A, true, PA, false), Hm = mul(A, true, PB, false), G = mul(B, true, PB, false);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) F(a, b) += m0.Qxx(a, b);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nu; ++b) Hm(a, b) += m0.Qxu(a, b);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nu; ++b) G(a, b) += m0.Quu(a, b);
    // the recursion mutates Qxu, Quu in place (brrf.cpp:39-41)
    worst = std::fmax(worst, relerr(kkt_matrix[i].Qxu.data(), Hm.data(), nx * nu));
    worst = std::fmax(worst, relerr(kkt_matrix[i].Quu.data(), G.data(), nu * nu));
    // G K = -H^T   (riccati_factorizer.cpp:55)
    Mat GK(nu, nx);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nx; ++b) {
        double acc = 0;
        for (int k = 0; k < nu; ++k) acc += G(a, k) * lqr[i].K(k, b);
        GK(a, b) = acc;
      }
    Mat mHt(nu, nx);
    for (int a = 0; a < nu; ++a)
      for (int b = 0; b < nx; ++b) mHt(a, b) = -Hm(b, a);
    worst = std::fmax(worst, relerr(GK.data(), mHt.data(), nu * nx));
    // P = sym(F - K^T G K)  (brrf.cpp:82-85)
    Mat Pref(nx, nx);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) {
        double acc = 0;
        for (int k = 0; k < nu; ++k) acc += lqr[i].K(k, a) * GK(k, b);
        Pref(a, b) = F(a, b) - acc;
      }
    Mat Ps(nx, nx);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) Ps(a, b) = 0.5 * (Pref(a, b) + Pref(b, a));
    worst = std::fmax(worst, relerr(factorization[i].P.data(), Ps.data(), nx * nx));
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < a; ++b)
        if (factorization[i].P(a, b) != factorization[i].P(b, a)) return 3;  // exactly symmetric
    // forward: du = K dx + k ; dx+ = Fx + A dx + B du ; dlmdgmm = P dx - s (riccati_factorizer.cpp:200-253)
    Vec du(nu), dxn(nx), lam(nx);
    for (int a = 0; a < nu; ++a) {
      double acc = lqr[i].k(a);
      for (int k = 0; k < nx; ++k) acc += lqr[i].K(a, k) * d[i].dx(k);
      du(a) = acc;
    }
    for (int a = 0; a < nx; ++a) {
      double acc = kkt_residual_ref[i].Fx(a);
      for (int k = 0; k < nx; ++k) acc += A(a, k) * d[i].dx(k);
      for (int k = 0; k < nu; ++k) acc += B(a, k) * du(k);
      dxn(a) = acc;
      double l = -factorization[i].s(a);
      for (int k = 0; k < nx; ++k) l += factorization[i].P(a, k) * d[i].dx(k);
      lam(a) = l;
    }
    worst = std::fmax(worst, relerr(d[i].du.data(), du.data(), nu));
    worst = std::fmax(worst, relerr(d[i + 1].dx.data(), dxn.data(), nx));
    worst = std::fmax(worst, relerr(d[i].dlmdgmm.data(), lam.data(), nx));
    (void)sn;
  }
  std::printf("robotoc::RiccatiRecursion (C++ host over the C ABI%s): worst rel err %.3e\n", scan ? ", horizon scan" : "", worst);
  // argument validation mirrors the reference's exceptions
  bool threw = false;
  try {
    riccati_recursion.setRegularization(-1.0);
  } catch (const std::out_of_range&) {
    threw = true;
  }
  // forwardRiccatiRecursion honours the containers it is handed (riccati_recursion.cpp:83-131): an edit of
  // kkt_residual between the two passes changes the result exactly as the recursion says, and undoing it restores it
  {
    Direction d1 = d;
    const double old = kkt_residual[2].Fx(0);
    kkt_residual[2].Fx(0) = old + 1.0;
    riccati_recursion.forwardRiccatiRecursion(td, kkt_matrix, kkt_residual, factorization, d);
    if (std::fabs((d[3].dx(0) - d1[3].dx(0)) - 1.0) > 1e-9) return 4;  // dx_3 = Fx_2 + ...
    kkt_residual[2].Fx(0) = old;
    riccati_recursion.forwardRiccatiRecursion(td, kkt_matrix, kkt_residual, factorization, d);
    for (int i = 0; i <= N; ++i)
      for (int k = 0; k < nx; ++k)
        if (d[i].dx(k) != d1[i].dx(k)) return 5;
  }
  // value semantics (riccati_recursion.hpp:50-60): a copy runs on its own device context
  {
    RiccatiRecursion copy(riccati_recursion);
    if (copy.context() == riccati_recursion.context()) return 6;
    Direction d2 = d;
    copy.forwardRiccatiRecursion(td, kkt_matrix, kkt_residual, factorization, d2);
    for (int k = 0; k < nx; ++k)
      if (d2[N].dx(k) != d[N].dx(k)) return 7;
    RiccatiRecursion assigned;
    assigned = copy;
    if (assigned.status() != 0) return 8;
  }
  return (worst < (scan ? 1e-8 : 1e-9) && threw) ? 0 : 1;
}

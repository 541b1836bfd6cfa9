// C++ host-side test of robotoc::UnconstrRiccatiRecursion (robotoc_amd/host/robotoc_hip.hpp) on the GPU,
// after the reference's test/riccati/unconstr_riccati_recursion_test.cpp:37-106: iiwa14 sizes (nv = 7,
// N = 20, T = 1), SPD [Qxx Qxu; . Qaa], random Fx, lx, la; the recursion is checked against a naive dense
// re-derivation with A = [[I, dt I],[0, I]], B = [0; dt I].  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

#include "../../robotoc_amd/host/robotoc_hip.hpp"

using namespace robotoc;

static std::mt19937_64 rng(20260926);
static double rnd() { return std::uniform_real_distribution<double>(-1.0, 1.0)(rng); }
static double relerr(const double* a, const double* b, int n) {
  double num = 0, den = 0;
  for (int i = 0; i < n; ++i) {
    num += (a[i] - b[i]) * (a[i] - b[i]);
    den += b[i] * b[i] > a[i] * a[i] ? b[i] * b[i] : a[i] * a[i];
  }
  return std::sqrt(num / (den > 1e-300 ? den : 1e-300));
}
static Mat mul(const Mat& X, bool tx, const Mat& Y, bool ty) {
  const int M = tx ? X.cols() : X.rows(), K = tx ? X.rows() : X.cols(), N = ty ? Y.rows() : Y.cols();
  Mat Z(M, N);
  for (int a = 0; a < M; ++a)
    for (int b = 0; b < N; ++b) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (tx ? X(k, a) : X(a, k)) * (ty ? Y(b, k) : Y(k, b));
      Z(a, b) = acc;
    }
  return Z;
}

int main(int argc, char** argv) {
  const bool scan = argc > 1 && std::string(argv[1]) == "scan";  // same checks with setHorizonScan, at 1e-8
  if (rtoc_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  const RobotDims robot = {7, 7, 0, 0};  // iiwa14
  OCP ocp;
  ocp.robot = robot;
  ocp.N = 20;
  ocp.T = 1.0;
  const int nv = 7, nx = 14, N = ocp.N;
  const double dt = ocp.T / N;
  KKTMatrix kkt_matrix(N + 1, SplitKKTMatrix(robot));
  KKTResidual kkt_residual(N + 1, SplitKKTResidual(robot));
  UnconstrRiccatiFactorization factorization(N + 1, SplitRiccatiFactorization(robot));
  Direction d(N + 1, SplitDirection(robot));
  for (int i = 0; i <= N; ++i) {
    const int n = nx + nv;
    Mat S(n, n), H(n, n);
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) S(a, b) = rnd();
    H = mul(S, false, S, true);
    for (int a = 0; a < n; ++a) H(a, a) += 1.0;
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) kkt_matrix[i].Qxx(a, b) = H(a, b);
    for (int a = 0; a < nx; ++a) kkt_residual[i].lx(a) = rnd();
    if (i == N) continue;
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nv; ++b) kkt_matrix[i].Qxu(a, b) = H(a, nx + b);  // [Qqa; Qva]
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) kkt_matrix[i].Quu(a, b) = H(nx + a, nx + b);  // Qaa
    for (int a = 0; a < nx; ++a) kkt_residual[i].Fx(a) = rnd();
    for (int a = 0; a < nv; ++a) kkt_residual[i].lu(a) = rnd();  // la
  }
  UnconstrRiccatiRecursion rr(ocp);
  rr.setHorizonScan(scan);
  rr.backwardRiccatiRecursion(kkt_matrix, kkt_residual, factorization);
  if (rr.status() != 0) return 1;
  for (int a = 0; a < nx; ++a) d[0].dx(a) = 0.1 * rnd();
  rr.forwardRiccatiRecursion(kkt_residual, factorization, d);
  const std::vector<LQRPolicy>& lqr = rr.getLQRPolicy();

  Mat A(nx, nx), B(nx, nv);
  for (int a = 0; a < nx; ++a) A(a, a) = 1.0;
  for (int a = 0; a < nv; ++a) {
    A(a, nv + a) = dt;
    B(nv + a, a) = dt;
  }
  double worst = relerr(factorization[N].P.data(), kkt_matrix[N].Qxx.data(), nx * nx);
  for (int i = N - 1; i >= 0; --i) {
    const Mat& Pn = factorization[i + 1].P;
    Mat PA = mul(Pn, false, A, false), PB = mul(Pn, false, B, false);
    Mat F = mul(A, true, PA, false), Hm = mul(A, true, PB, false), G = mul(B, true, PB, false);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) F(a, b) += kkt_matrix[i].Qxx(a, b);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nv; ++b) Hm(a, b) += kkt_matrix[i].Qxu(a, b);
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) G(a, b) += kkt_matrix[i].Quu(a, b);
    // G K = -H^T (unconstr_riccati_factorizer.cpp:32-33)
    Mat GK(nv, nx), mHt(nv, nx);
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nx; ++b) {
        double acc = 0;
        for (int k = 0; k < nv; ++k) acc += G(a, k) * lqr[i].K(k, b);
        GK(a, b) = acc;
        mHt(a, b) = -Hm(b, a);
      }
    worst = std::fmax(worst, relerr(GK.data(), mHt.data(), nv * nx));
    // P = sym(F - K^T G K) (unconstr_backward_riccati_recursion_factorizer.cpp:58-64)
    Mat Pref(nx, nx);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) {
        double acc = 0;
        for (int k = 0; k < nv; ++k) acc += lqr[i].K(k, a) * GK(k, b);
        Pref(a, b) = F(a, b) - acc;
      }
    Mat Ps(nx, nx);
    for (int a = 0; a < nx; ++a)
      for (int b = 0; b < nx; ++b) Ps(a, b) = 0.5 * (Pref(a, b) + Pref(b, a));
    worst = std::fmax(worst, relerr(factorization[i].P.data(), Ps.data(), nx * nx));
    // forward (unconstr_riccati_factorizer.cpp:40-59): da = K dx + k ; dx+ = Fx + A dx + B da
    Vec da(nv), dxn(nx), lam(nx);
    for (int a = 0; a < nv; ++a) {
      double acc = lqr[i].k(a);
      for (int k = 0; k < nx; ++k) acc += lqr[i].K(a, k) * d[i].dx(k);
      da(a) = acc;
    }
    for (int a = 0; a < nx; ++a) {
      double acc = kkt_residual[i].Fx(a);
      for (int k = 0; k < nx; ++k) acc += A(a, k) * d[i].dx(k);
      for (int k = 0; k < nv; ++k) acc += B(a, k) * da(k);
      dxn(a) = acc;
      double l = -factorization[i].s(a);
      for (int k = 0; k < nx; ++k) l += factorization[i].P(a, k) * d[i].dx(k);
      lam(a) = l;
    }
    worst = std::fmax(worst, relerr(d[i].du.data(), da.data(), nv));
    worst = std::fmax(worst, relerr(d[i + 1].dx.data(), dxn.data(), nx));
    worst = std::fmax(worst, relerr(d[i].dlmdgmm.data(), lam.data(), nx));
  }
  std::printf("robotoc::UnconstrRiccatiRecursion (C++ host over the C ABI%s): worst rel err %.3e\n", scan ? ", horizon scan" : "", worst);
  bool threw = false;
  try {
    OCP bad = ocp;
    bad.robot = RobotDims{18, 12, 6, 12};
    UnconstrRiccatiRecursion x(bad);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  return (worst < (scan ? 1e-8 : 1e-9) && threw) ? 0 : 1;
}

// C++ host-side test of robotoc::condenseContactDynamics / expandContactDynamicsPrimal / expandContactDynamicsDual
// and condenseImpactDynamics / expandImpactDynamicsPrimal / expandImpactDynamicsDual
// (robotoc_amd/host/robotoc_hip_dynamics.hpp) on the GPU, after the reference's
// test/dynamics/contact_dynamics_test.cpp:86-229 and test/dynamics/impact_dynamics_test.cpp:73-129: a quadruped (dimv 18, dimu 12, 4 point contacts) with an
// empty, a half and a full contact status; random linearisation data; every condensed member is checked
// against the closed form written with naive dense algebra (the saddle-matrix inverse by Gauss-Jordan, i.e.
// the defining identity of Robot::computeMJtJinv).  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../robotoc_amd/host/robotoc_hip_dynamics.hpp"

using namespace robotoc;

static std::mt19937_64 rng(20260927);
static double rnd() { return std::uniform_real_distribution<double>(-1.0, 1.0)(rng); }
static int failures = 0;

static void expect_approx(const char* what, const Mat& a, const Mat& b, int rows, int cols, double tol = 1e-10) {
  double num = 0, den = 0;
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) {
      num += (a(i, j) - b(i, j)) * (a(i, j) - b(i, j));
      den += b(i, j) * b(i, j);
    }
  const double e = std::sqrt(num / (den > 1e-300 ? den : 1.0));
  if (!(e < tol)) {
    std::printf("FAIL %-24s rel err %.3e\n", what, e);
    ++failures;
  }
}
static void expect_approx(const char* what, const Vec& a, const Vec& b, int n, double tol = 1e-10) {
  double num = 0, den = 0;
  for (int i = 0; i < n; ++i) {
    num += (a(i) - b(i)) * (a(i) - b(i));
    den += b(i) * b(i);
  }
  const double e = std::sqrt(num / (den > 1e-300 ? den : 1.0));
  if (!(e < tol)) {
    std::printf("FAIL %-24s rel err %.3e\n", what, e);
    ++failures;
  }
}
static Mat block(const Mat& X, int r0, int c0, int nr, int nc) {
  Mat Z(nr, nc);
  for (int j = 0; j < nc; ++j)
    for (int i = 0; i < nr; ++i) Z(i, j) = X(r0 + i, c0 + j);
  return Z;
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
static Vec mulv(const Mat& X, bool tx, const Vec& y) {
  const int M = tx ? X.cols() : X.rows(), K = tx ? X.rows() : X.cols();
  Vec z(M);
  for (int a = 0; a < M; ++a) {
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (tx ? X(k, a) : X(a, k)) * y(k);
    z(a) = acc;
  }
  return z;
}
static Mat inverse(Mat A) {  // Gauss-Jordan with partial pivoting
  const int n = A.rows();
  Mat I(n, n);
  for (int i = 0; i < n; ++i) I(i, i) = 1.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(A(r, c)) > std::fabs(A(p, c))) p = r;
    for (int j = 0; j < n; ++j) {
      std::swap(A(c, j), A(p, j));
      std::swap(I(c, j), I(p, j));
    }
    const double inv = 1.0 / A(c, c);
    for (int j = 0; j < n; ++j) {
      A(c, j) *= inv;
      I(c, j) *= inv;
    }
    for (int r = 0; r < n; ++r)
      if (r != c) {
        const double f = A(r, c);
        for (int j = 0; j < n; ++j) {
          A(r, j) -= f * A(c, j);
          I(r, j) -= f * I(c, j);
        }
      }
  }
  return I;
}

static void run(Robot& robot, const int dimf) {
  const int dimv = robot.dimv(), dimu = robot.dimu(), dimp = robot.dim_passive(), dimx = 2 * dimv, dimvf = dimv + dimf;
  const double dt = 0.01;
  const ContactStatus contact_status(dimf);
  ContactDynamicsData data(robot);
  data.setContactDimension(contact_status.dimf());
  // what linearizeContactDynamics leaves (Pinocchio side): M = dIDda SPD, J = dCda, dIDCdqv, IDC, lu_passive
  {
    Mat Lo(dimv, dimv);
    for (int j = 0; j < dimv; ++j)
      for (int i = j; i < dimv; ++i) Lo(i, j) = rnd();
    data.dIDda = mul(Lo, false, Lo, true);
    for (int i = 0; i < dimv; ++i) data.dIDda(i, i) += 1.0;
  }
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimf; ++i) data.dCda_full(i, j) = rnd();
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimvf; ++i) data.dIDCdqv_full(i, j) = rnd();
  for (int i = 0; i < dimvf; ++i) data.IDC_full(i) = rnd();
  for (int i = 0; i < dimp; ++i) data.lu_passive(i) = rnd();
  // SplitKKTMatrix::Random / SplitKKTResidual::Random with the state-equation blocks of the reference test (:110-118)
  SplitKKTMatrix kkt_matrix(robot.dims());
  SplitKKTResidual kkt_residual(robot.dims());
  {
    Mat S(dimx, dimx);
    for (int j = 0; j < dimx; ++j)
      for (int i = 0; i < dimx; ++i) S(i, j) = rnd();
    kkt_matrix.Qxx = mul(S, false, S, true);
    Mat U(dimu, dimu);
    for (int j = 0; j < dimu; ++j)
      for (int i = 0; i < dimu; ++i) U(i, j) = rnd();
    kkt_matrix.Quu = mul(U, false, U, true);
    Mat F(dimf > 0 ? dimf : 1, dimf > 0 ? dimf : 1);
    for (int j = 0; j < dimf; ++j)
      for (int i = 0; i < dimf; ++i) F(i, j) = rnd();
    const Mat FF = mul(F, false, F, true);
    for (int j = 0; j < dimf; ++j)
      for (int i = 0; i < dimf; ++i) kkt_matrix.Qff_full(i, j) = FF(i, j);
  }
  for (int j = 0; j < dimu; ++j)
    for (int i = 0; i < dimx; ++i) kkt_matrix.Qxu(i, j) = rnd();
  for (int i = 0; i < dimv; ++i) kkt_matrix.Qaa(i, i) = rnd();  // Qaa.setZero(); Qaa.diagonal().setRandom() (:107-108)
  for (int j = 0; j < dimf; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix.Qqf_full(i, j) = rnd();
  for (int i = 0; i < dimv; ++i) kkt_matrix.Fxx(i, i) = 1.0;  // Fqq = I (+ random 6x6 corner), Fqv = dt I
  if (robot.hasFloatingBase())
    for (int j = 0; j < 6; ++j)
      for (int i = 0; i < 6; ++i) kkt_matrix.Fxx(i, j) = rnd();
  for (int i = 0; i < dimv; ++i) kkt_matrix.Fxx(i, dimv + i) = dt;
  for (int i = 0; i < dimx; ++i) {
    kkt_residual.Fx(i) = rnd();
    kkt_residual.lx(i) = rnd();
    kkt_matrix.hx(i) = rnd();
  }
  for (int i = 0; i < dimu; ++i) {
    kkt_residual.lu(i) = rnd();
    kkt_matrix.hu(i) = rnd();
  }
  for (int i = 0; i < dimv; ++i) {
    kkt_residual.la(i) = rnd();
    kkt_matrix.ha(i) = rnd();
  }
  for (int i = 0; i < dimf; ++i) {
    kkt_residual.lf_full(i) = rnd();
    kkt_matrix.hf_full(i) = rnd();
  }
  kkt_residual.h = rnd();
  SplitKKTMatrix kkt_matrix_ref = kkt_matrix;
  SplitKKTResidual kkt_residual_ref = kkt_residual;
  const Vec lu_passive_in = data.lu_passive;

  condenseContactDynamics(robot, contact_status, dt, data, kkt_matrix, kkt_residual);

  // ---- closed form (reference test :121-177) ----
  Mat saddle(dimvf, dimvf);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimv; ++i) saddle(i, j) = data.dIDda(i, j);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimf; ++i) {
      saddle(dimv + i, j) = data.dCda_full(i, j);
      saddle(j, dimv + i) = data.dCda_full(i, j);
    }
  const Mat MJtJinv = inverse(saddle);  // robot.computeMJtJinv (:120)
  const Mat dIDCdqv = block(data.dIDCdqv_full, 0, 0, dimvf, dimx);
  Vec IDC(dimvf);
  for (int i = 0; i < dimvf; ++i) IDC(i) = data.IDC_full(i);
  const Mat MJtJinv_dIDCdqv = mul(MJtJinv, false, dIDCdqv, false);
  const Vec MJtJinv_IDC = mulv(MJtJinv, false, IDC);
  Mat Qaaff(dimvf, dimvf);
  for (int i = 0; i < dimv; ++i) Qaaff(i, i) = kkt_matrix_ref.Qaa(i, i);
  for (int j = 0; j < dimf; ++j)
    for (int i = 0; i < dimf; ++i) Qaaff(dimv + i, dimv + j) = kkt_matrix_ref.Qff_full(i, j);
  Mat Qafqv = mul(Qaaff, false, MJtJinv_dIDCdqv, false);
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimvf; ++i) Qafqv(i, j) = -Qafqv(i, j);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimf; ++i) Qafqv(dimv + i, j) -= kkt_matrix_ref.Qqf_full(j, i);
  Mat IO_mat(dimvf, dimv);
  for (int i = 0; i < dimv; ++i) IO_mat(i, i) = 1.0;
  const Mat Qafu_full = mul(mul(Qaaff, false, MJtJinv, false), false, IO_mat, false);
  Vec laf(dimvf), haf(dimvf);
  for (int i = 0; i < dimv; ++i) {
    laf(i) = kkt_residual_ref.la(i);
    haf(i) = kkt_matrix_ref.ha(i);
  }
  for (int i = 0; i < dimf; ++i) {
    laf(dimv + i) = -kkt_residual_ref.lf_full(i);
    haf(dimv + i) = -kkt_matrix_ref.hf_full(i);
  }
  {
    const Vec t = mulv(mul(Qaaff, false, MJtJinv, false), false, IDC);
    for (int i = 0; i < dimvf; ++i) laf(i) -= t(i);
  }
  const Mat Qqf = block(kkt_matrix_ref.Qqf_full, 0, 0, dimv, dimf > 0 ? dimf : 0);
  {
    const Mat t = mul(MJtJinv_dIDCdqv, true, Qafqv, false);
    for (int j = 0; j < dimx; ++j)
      for (int i = 0; i < dimx; ++i) kkt_matrix_ref.Qxx(i, j) -= t(i, j);
    if (dimf > 0) {
      const Mat t2 = mul(Qqf, false, block(MJtJinv_dIDCdqv, dimv, 0, dimf, dimx), false);
      for (int j = 0; j < dimx; ++j)
        for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Qxx(i, j) += t2(i, j);
    }
  }
  Mat Qxu_full(dimx, dimv);
  for (int j = 0; j < dimu; ++j)
    for (int i = 0; i < dimx; ++i) Qxu_full(i, dimp + j) = kkt_matrix_ref.Qxu(i, j);
  {
    const Mat t = mul(MJtJinv_dIDCdqv, true, Qafu_full, false);
    for (int j = 0; j < dimv; ++j)
      for (int i = 0; i < dimx; ++i) Qxu_full(i, j) -= t(i, j);
    if (dimf > 0) {
      const Mat t2 = mul(Qqf, false, block(MJtJinv, dimv, 0, dimf, dimv), false);
      for (int j = 0; j < dimv; ++j)
        for (int i = 0; i < dimv; ++i) Qxu_full(i, j) -= t2(i, j);
    }
  }
  const Mat Qxu_passive_ref = block(Qxu_full, 0, 0, dimx, dimp);
  kkt_matrix_ref.Qxu = block(Qxu_full, 0, dimp, dimx, dimu);
  const Mat Quu_full = mul(mul(IO_mat, true, MJtJinv, false), false, Qafu_full, false);
  const Mat Quu_passive_topRight_ref = block(Quu_full, 0, dimp, dimp, dimu);
  for (int j = 0; j < dimu; ++j)
    for (int i = 0; i < dimu; ++i) kkt_matrix_ref.Quu(i, j) += Quu_full(dimp + i, dimp + j);
  {
    const Vec t = mulv(MJtJinv_dIDCdqv, true, laf);
    for (int i = 0; i < dimx; ++i) kkt_residual_ref.lx(i) -= t(i);
    if (dimf > 0) {
      Vec tail(dimf);
      for (int i = 0; i < dimf; ++i) tail(i) = MJtJinv_IDC(dimv + i);
      const Vec t2 = mulv(Qqf, false, tail);
      for (int i = 0; i < dimv; ++i) kkt_residual_ref.lx(i) += t2(i);
    }
  }
  Vec lu_full(dimv), hu_full(dimv);
  for (int i = 0; i < dimp; ++i) lu_full(i) = lu_passive_in(i);
  for (int i = 0; i < dimu; ++i) {
    lu_full(dimp + i) = kkt_residual_ref.lu(i);
    hu_full(dimp + i) = kkt_matrix_ref.hu(i);
  }
  {
    const Mat IOtM = mul(IO_mat, true, MJtJinv, false);
    const Vec t = mulv(IOtM, false, laf), th = mulv(IOtM, false, haf);
    for (int i = 0; i < dimv; ++i) {
      lu_full(i) += t(i);
      hu_full(i) += th(i);
    }
  }
  Vec lu_passive_ref(dimp);
  for (int i = 0; i < dimp; ++i) lu_passive_ref(i) = lu_full(i);
  for (int i = 0; i < dimu; ++i) {
    kkt_residual_ref.lu(i) = lu_full(dimp + i);
    kkt_matrix_ref.hu(i) = hu_full(dimp + i);
  }
  // Fvv = I; Fxx -= OOIO MJtJinv_dIDCdqv; Fvu; Fx (:158-163)
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Fxx(dimv + i, dimv + j) = (i == j) ? 1.0 : 0.0;
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Fxx(dimv + i, j) -= dt * MJtJinv_dIDCdqv(i, j);
  for (int j = 0; j < dimu; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Fvu(i, j) = dt * MJtJinv(i, dimp + j);
  for (int i = 0; i < dimv; ++i) kkt_residual_ref.Fx(dimv + i) -= dt * MJtJinv_IDC(i);
  // STO sensitivities (:165-175)
  {
    double dot = 0;
    for (int i = 0; i < dimvf; ++i) dot += MJtJinv_IDC(i) * haf(i);
    kkt_residual_ref.h -= dot;
    const Vec t = mulv(MJtJinv_dIDCdqv, true, haf);
    for (int i = 0; i < dimx; ++i) kkt_matrix_ref.hx(i) -= t(i);
    if (dimf > 0) {
      Vec tail(dimf);
      for (int i = 0; i < dimf; ++i) tail(i) = MJtJinv_IDC(dimv + i);
      const Vec t2 = mulv(Qqf, false, tail);
      for (int i = 0; i < dimv; ++i) kkt_matrix_ref.hx(i) += (1.0 / dt) * t2(i);
    }
  }

  expect_approx("MJtJinv", data.MJtJinv_full, MJtJinv, dimvf, dimvf);
  expect_approx("MJtJinv_dIDCdqv", data.MJtJinv_dIDCdqv_full, MJtJinv_dIDCdqv, dimvf, dimx);
  expect_approx("MJtJinv_IDC", data.MJtJinv_IDC_full, MJtJinv_IDC, dimvf);
  expect_approx("Qafqv", data.Qafqv_full, Qafqv, dimvf, dimx);
  expect_approx("Qafu_full", data.Qafu_full_full, Qafu_full, dimvf, dimv);
  expect_approx("laf", data.laf_full, laf, dimvf);
  expect_approx("haf", data.haf_full, haf, dimvf);
  expect_approx("Qxu_passive", data.Qxu_passive, Qxu_passive_ref, dimx, dimp);
  expect_approx("Quu_passive_topRight", data.Quu_passive_topRight, Quu_passive_topRight_ref, dimp, dimu);
  expect_approx("lu_passive", data.lu_passive, lu_passive_ref, dimp);
  expect_approx("Qxx", kkt_matrix.Qxx, kkt_matrix_ref.Qxx, dimx, dimx);
  expect_approx("Qxu", kkt_matrix.Qxu, kkt_matrix_ref.Qxu, dimx, dimu);
  expect_approx("Quu", kkt_matrix.Quu, kkt_matrix_ref.Quu, dimu, dimu);
  expect_approx("Fxx", kkt_matrix.Fxx, kkt_matrix_ref.Fxx, dimx, dimx);
  expect_approx("Fvu", kkt_matrix.Fvu, kkt_matrix_ref.Fvu, dimv, dimu);
  expect_approx("hx", kkt_matrix.hx, kkt_matrix_ref.hx, dimx);
  expect_approx("hu", kkt_matrix.hu, kkt_matrix_ref.hu, dimu);
  expect_approx("lx", kkt_residual.lx, kkt_residual_ref.lx, dimx);
  expect_approx("lu", kkt_residual.lu, kkt_residual_ref.lu, dimu);
  expect_approx("Fx", kkt_residual.Fx, kkt_residual_ref.Fx, dimx);
  if (!(std::fabs(kkt_residual.h - kkt_residual_ref.h) < 1e-10 * (1.0 + std::fabs(kkt_residual_ref.h)))) {
    std::printf("FAIL h %.12e vs %.12e\n", kkt_residual.h, kkt_residual_ref.h);
    ++failures;
  }
  // EXPECT_TRUE(kkt_matrix.Qxx.isApprox(kkt_matrix.Qxx.transpose())) / Quu (:179-180)
  Mat Qxxt(dimx, dimx), Quut(dimu, dimu);
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimx; ++i) Qxxt(i, j) = kkt_matrix.Qxx(j, i);
  for (int j = 0; j < dimu; ++j)
    for (int i = 0; i < dimu; ++i) Quut(i, j) = kkt_matrix.Quu(j, i);
  expect_approx("Qxx symmetric", kkt_matrix.Qxx, Qxxt, dimx, dimx);
  expect_approx("Quu symmetric", kkt_matrix.Quu, Quut, dimu, dimu);

  // ---- expansion (:182-207) ----
  SplitDirection d(robot.dims()), d_next(robot.dims());
  for (int i = 0; i < dimx; ++i) {
    d.dx(i) = rnd();
    d_next.dlmdgmm(i) = rnd();
  }
  for (int i = 0; i < dimu; ++i) d.du(i) = rnd();
  expandContactDynamicsPrimal(data, d);
  Vec du_full(dimv);
  for (int i = 0; i < dimu; ++i) du_full(dimp + i) = d.du(i);
  Vec daf_ref(dimvf);
  {
    Vec t = mulv(dIDCdqv, false, d.dx);
    const Vec iu = mulv(IO_mat, false, du_full);
    for (int i = 0; i < dimvf; ++i) t(i) = t(i) - iu(i) + IDC(i);
    const Vec r = mulv(MJtJinv, false, t);
    for (int i = 0; i < dimvf; ++i) daf_ref(i) = (i < dimv ? -r(i) : r(i));  // df *= -1
  }
  expect_approx("daf", d.daf_full, daf_ref, dimvf);
  expect_approx("laf untouched by the primal expansion", data.laf_full, laf, dimvf);

  const double dts = rnd();
  expandContactDynamicsDual(dt, dts, data, d_next, d);
  Vec rhs(dimvf);
  {
    const Vec a = mulv(Qafqv, false, d.dx), b = mulv(Qafu_full, false, du_full);
    for (int i = 0; i < dimvf; ++i) rhs(i) = a(i) + b(i) + laf(i) + dts * haf(i);
    for (int i = 0; i < dimv; ++i) rhs(i) += dt * d_next.dlmdgmm(dimv + i);  // OOIO^T dlmdgmm
  }
  Vec dbetamu_ref = mulv(MJtJinv, false, rhs);
  for (int i = 0; i < dimvf; ++i) dbetamu_ref(i) = -dbetamu_ref(i);
  expect_approx("dbetamu", d.dbetamu_full, dbetamu_ref, dimvf);
  expect_approx("laf after the dual expansion", data.laf_full, rhs, dimvf);
  if (robot.hasFloatingBase()) {
    Vec dnu_ref(dimp);
    const Vec a = mulv(Qxu_passive_ref, true, d.dx), b = mulv(Quu_passive_topRight_ref, false, d.du);
    for (int i = 0; i < dimp; ++i) {
      double g = 0;
      for (int k = 0; k < dimv; ++k) g += MJtJinv(i, k) * dt * d_next.dlmdgmm(dimv + k);
      dnu_ref(i) = -(lu_passive_ref(i) + a(i) + b(i) + g);
    }
    expect_approx("dnu_passive", d.dnu_passive, dnu_ref, dimp);
  }
}

// after test/dynamics/impact_dynamics_test.cpp:73-129
static void run_impact(Robot& robot, const int dimf) {
  const int dimv = robot.dimv(), dimx = 2 * dimv, dimvf = dimv + dimf;
  const ImpactStatus impact_status(dimf);
  ContactDynamicsData data(robot);
  data.setContactDimension(impact_status.dimf());
  {
    Mat Lo(dimv, dimv);
    for (int j = 0; j < dimv; ++j)
      for (int i = j; i < dimv; ++i) Lo(i, j) = rnd();
    data.dIDddv = mul(Lo, false, Lo, true);
    for (int i = 0; i < dimv; ++i) data.dIDddv(i, i) += 1.0;
  }
  // dIDCdqv = [dIDdq 0; dCdq dCdv] (RNEAImpactDerivatives has no velocity block)
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimvf; ++i) data.dIDCdqv_full(i, j) = (i < dimv && j >= dimv) ? 0.0 : rnd();
  for (int i = 0; i < dimvf; ++i) data.IDC_full(i) = rnd();
  SplitKKTMatrix kkt_matrix(robot.dims());
  SplitKKTResidual kkt_residual(robot.dims());
  {
    Mat S(dimx, dimx);
    for (int j = 0; j < dimx; ++j)
      for (int i = 0; i < dimx; ++i) S(i, j) = rnd();
    kkt_matrix.Qxx = mul(S, false, S, true);
    Mat F(dimf, dimf);
    for (int j = 0; j < dimf; ++j)
      for (int i = 0; i < dimf; ++i) F(i, j) = rnd();
    const Mat FF = mul(F, false, F, true);
    for (int j = 0; j < dimf; ++j)
      for (int i = 0; i < dimf; ++i) kkt_matrix.Qff_full(i, j) = FF(i, j);
  }
  for (int i = 0; i < dimv; ++i) kkt_matrix.Qdvdv(i, i) = rnd();  // Qdvdv.setZero(); diagonal().setRandom() (:95-96)
  for (int j = 0; j < dimf; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix.Qqf_full(i, j) = rnd();
  // kkt_matrix.Fxx.setZero() (:94)
  for (int i = 0; i < dimx; ++i) {
    kkt_residual.Fx(i) = rnd();
    kkt_residual.lx(i) = rnd();
  }
  for (int i = 0; i < dimv; ++i) kkt_residual.ldv(i) = rnd();
  for (int i = 0; i < dimf; ++i) kkt_residual.lf_full(i) = rnd();
  SplitKKTMatrix kkt_matrix_ref = kkt_matrix;
  SplitKKTResidual kkt_residual_ref = kkt_residual;

  condenseImpactDynamics(robot, impact_status, data, kkt_matrix, kkt_residual);

  Mat saddle(dimvf, dimvf);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimv; ++i) saddle(i, j) = data.dIDddv(i, j);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimf; ++i) {
      saddle(dimv + i, j) = data.dIDCdqv_full(dimv + i, dimv + j);  // dCdv
      saddle(j, dimv + i) = data.dIDCdqv_full(dimv + i, dimv + j);
    }
  const Mat MJtJinv = inverse(saddle);  // robot.computeMJtJinv(dIDddv, dCdv) (:99)
  const Mat dIDCdqv = block(data.dIDCdqv_full, 0, 0, dimvf, dimx);
  Vec IDC(dimvf);
  for (int i = 0; i < dimvf; ++i) IDC(i) = data.IDC_full(i);
  const Mat MJtJinv_dIDCdqv = mul(MJtJinv, false, dIDCdqv, false);
  const Vec MJtJinv_IDC = mulv(MJtJinv, false, IDC);
  Mat Qdvdvff(dimvf, dimvf);
  for (int i = 0; i < dimv; ++i) Qdvdvff(i, i) = kkt_matrix_ref.Qdvdv(i, i);
  for (int j = 0; j < dimf; ++j)
    for (int i = 0; i < dimf; ++i) Qdvdvff(dimv + i, dimv + j) = kkt_matrix_ref.Qff_full(i, j);
  Mat Qdvfqv = mul(Qdvdvff, false, MJtJinv_dIDCdqv, false);
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimvf; ++i) Qdvfqv(i, j) = -Qdvfqv(i, j);
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimf; ++i) Qdvfqv(dimv + i, j) -= kkt_matrix_ref.Qqf_full(j, i);
  Vec ldvf(dimvf);
  for (int i = 0; i < dimv; ++i) ldvf(i) = kkt_residual_ref.ldv(i);
  for (int i = 0; i < dimf; ++i) ldvf(dimv + i) = -kkt_residual_ref.lf_full(i);
  {
    const Vec t = mulv(Qdvdvff, false, MJtJinv_IDC);
    for (int i = 0; i < dimvf; ++i) ldvf(i) -= t(i);
  }
  const Mat Qqf = block(kkt_matrix_ref.Qqf_full, 0, 0, dimv, dimf);
  {
    const Mat t = mul(MJtJinv_dIDCdqv, true, Qdvfqv, false);
    const Mat t2 = mul(Qqf, false, block(MJtJinv_dIDCdqv, dimv, 0, dimf, dimx), false);
    for (int j = 0; j < dimx; ++j)
      for (int i = 0; i < dimx; ++i) kkt_matrix_ref.Qxx(i, j) -= t(i, j);
    for (int j = 0; j < dimx; ++j)
      for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Qxx(i, j) += t2(i, j);
    const Vec tv = mulv(MJtJinv_dIDCdqv, true, ldvf);
    for (int i = 0; i < dimx; ++i) kkt_residual_ref.lx(i) -= tv(i);
    Vec tail(dimf);
    for (int i = 0; i < dimf; ++i) tail(i) = MJtJinv_IDC(dimv + i);
    const Vec t3 = mulv(Qqf, false, tail);
    for (int i = 0; i < dimv; ++i) kkt_residual_ref.lx(i) += t3(i);
  }
  for (int j = 0; j < dimv; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Fxx(dimv + i, dimv + j) = (i == j) ? 1.0 : 0.0;  // Fvv().setIdentity()
  for (int j = 0; j < dimx; ++j)
    for (int i = 0; i < dimv; ++i) kkt_matrix_ref.Fxx(dimv + i, j) -= MJtJinv_dIDCdqv(i, j);
  for (int i = 0; i < dimv; ++i) kkt_residual_ref.Fx(dimv + i) -= MJtJinv_IDC(i);

  expect_approx("impact MJtJinv", data.MJtJinv_full, MJtJinv, dimvf, dimvf);
  expect_approx("impact MJtJinv_dIDCdqv", data.MJtJinv_dIDCdqv_full, MJtJinv_dIDCdqv, dimvf, dimx);
  expect_approx("impact MJtJinv_IDC", data.MJtJinv_IDC_full, MJtJinv_IDC, dimvf);
  expect_approx("impact Qdvfqv", data.Qafqv_full, Qdvfqv, dimvf, dimx);
  expect_approx("impact ldvf", data.laf_full, ldvf, dimvf);
  expect_approx("impact Qxx", kkt_matrix.Qxx, kkt_matrix_ref.Qxx, dimx, dimx);
  expect_approx("impact Fxx", kkt_matrix.Fxx, kkt_matrix_ref.Fxx, dimx, dimx);
  expect_approx("impact lx", kkt_residual.lx, kkt_residual_ref.lx, dimx);
  expect_approx("impact Fx", kkt_residual.Fx, kkt_residual_ref.Fx, dimx);

  SplitDirection d(robot.dims()), d_next(robot.dims());
  for (int i = 0; i < dimx; ++i) {
    d.dx(i) = rnd();
    d_next.dlmdgmm(i) = rnd();
  }
  expandImpactDynamicsPrimal(data, d);
  Vec ddvf_ref(dimvf);
  {
    Vec t = mulv(dIDCdqv, false, d.dx);
    for (int i = 0; i < dimvf; ++i) t(i) += IDC(i);
    const Vec r = mulv(MJtJinv, false, t);
    for (int i = 0; i < dimvf; ++i) ddvf_ref(i) = (i < dimv ? -r(i) : r(i));
  }
  expect_approx("impact ddvf", d.daf_full, ddvf_ref, dimvf);
  expandImpactDynamicsDual(data, d_next, d);
  Vec rhs = mulv(Qdvfqv, false, d.dx);
  for (int i = 0; i < dimvf; ++i) rhs(i) += ldvf(i);
  for (int i = 0; i < dimv; ++i) rhs(i) += d_next.dlmdgmm(dimv + i);
  Vec dbetamu_ref = mulv(MJtJinv, false, rhs);
  for (int i = 0; i < dimvf; ++i) dbetamu_ref(i) = -dbetamu_ref(i);
  expect_approx("impact dbetamu", d.dbetamu_full, dbetamu_ref, dimvf);
}

int main() {
  if (rtoc_device_count() < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 2;
  }
  Robot robot(18, 12, 6, 12);  // quadruped: floating base, four point contacts
  for (int rep = 0; rep < 3; ++rep)
    for (int dimf : {0, 6, 12}) run(robot, dimf);
  for (int rep = 0; rep < 2; ++rep)
    for (int dimf : {6, 12}) run_impact(robot, dimf);
  // argument checks in the reference's style
  bool threw = false;
  try {
    ContactDynamicsData data(robot);
    data.setContactDimension(6);
    SplitKKTMatrix m(robot.dims());
    SplitKKTResidual r(robot.dims());
    condenseContactDynamics(robot, ContactStatus(12), 0.01, data, m, r);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  if (!threw) {
    std::printf("FAIL mismatching contact dimension must throw\n");
    ++failures;
  }
  threw = false;
  try {
    ContactDynamicsData data(robot);
    SplitDirection d(robot.dims());
    expandContactDynamicsPrimal(data, d);
  } catch (const std::logic_error&) {
    threw = true;
  }
  if (!threw) {
    std::printf("FAIL expansion before condensation must throw\n");
    ++failures;
  }
  std::printf(failures ? "%d FAILURES\n" : "contact_dynamics_test: all checks passed\n", failures);
  return failures ? 1 : 0;
}

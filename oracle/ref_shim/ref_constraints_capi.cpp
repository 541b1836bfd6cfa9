// ref_constraints_capi.cpp -- one grid point of the REFERENCE'S OWN Constraints object on plain arrays.
// TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref.so, built by oracle/Makefile.ref from the sources under /root/reference).
//
// Reference code that runs here: Constraints (src/constraints/constraints.cpp) with the six joint-limit components
// (joint_{position,velocity,torques}_{lower,upper}_limit.cpp; ref_accel_limits_stage: joint_acceleration_{lower,upper}_limit.cpp), FrictionCone / ImpactFrictionCone (friction_cone.cpp,
// impact_friction_cone.cpp) or ContactWrenchCone / ImpactWrenchCone (contact_wrench_cone.cpp, impact_wrench_cone.cpp), ConstraintsData's stage mask
// (constraints_data.cpp), pdipm.hxx.  What does not: Eigen (oracle/ref_shim/mini_eigen.hpp) and Pinocchio -- the frame
// kinematics the cones read (world rotation and LOCAL Jacobian of the contact frames) are INJECTED by the caller, so what is
// pinned is the reference's composition of them (the rigid-body kinematics themselves stay unpinned).
#include <memory>
#include <vector>

#include "robotoc/constraints/constraints.hpp"
#include "robotoc/constraints/contact_wrench_cone.hpp"
#include "robotoc/constraints/friction_cone.hpp"
#include "robotoc/constraints/impact_friction_cone.hpp"
#include "robotoc/constraints/impact_wrench_cone.hpp"
#include "robotoc/constraints/joint_acceleration_lower_limit.hpp"
#include "robotoc/constraints/joint_acceleration_upper_limit.hpp"
#include "robotoc/constraints/joint_position_lower_limit.hpp"
#include "robotoc/constraints/joint_position_upper_limit.hpp"
#include "robotoc/constraints/joint_torques_lower_limit.hpp"
#include "robotoc/constraints/joint_torques_upper_limit.hpp"
#include "robotoc/constraints/joint_velocity_lower_limit.hpp"
#include "robotoc/constraints/joint_velocity_upper_limit.hpp"

using namespace robotoc;

namespace {
void load(ConstraintComponentData& d, const double* slack, const double* dual, const double* residual, const double* cmpl, int n) {
  for (int i = 0; i < n; ++i) d.slack(i) = slack[i], d.dual(i) = dual[i], d.residual(i) = residual[i], d.cmpl(i) = cmpl[i];
}
void store(const ConstraintComponentData& d, double* slack, double* dual, double* residual, double* cmpl, double* cond, double* dslack,
           double* ddual, int n) {
  for (int i = 0; i < n; ++i)
    slack[i] = d.slack(i), dual[i] = d.dual(i), residual[i] = d.residual(i), cmpl[i] = d.cmpl(i), cond[i] = d.cond(i),
    dslack[i] = d.dslack(i), ddual[i] = d.ddual(i);
}
}  // namespace

extern "C" {

// phase_mask: 1 setSlackAndDual, 2 linearizeConstraints (evalConstraint + evalDerivatives), 4 condenseSlackAndDual,
// 8 expandSlackAndDual + maxSlackStepSize / maxDualStepSize.  Row order of slack ... ddual: the components in the order they are
// added (q lower, q upper, v lower, v upper, u lower, u upper: nu rows each; then the cone rows by contact INDEX, 5 or 17 each).
// Blocks are dense column-major: Qxx (2nv x 2nv), Quu (nu x nu), Qqf (nv x dimf), Qff (dimf x dimf), dimf = rows of the
// ACTIVE contacts; f: [ncontacts][6] (the first contact_dim entries used).  dg_dq / dg_df out: per contact index, 5 x nv / 5 x 3.
int ref_constraints_stage(int nv, int nu, int ncontacts, int contact_dim, int time_stage, int impact, unsigned active, const double* mu,
                          const double* surf_rot, const double* frame_R, const double* frame_Jlocal, const double* limits, int cone_kind,
                          double X, double Y, double barrier, double tau, const double* q, const double* v, const double* u, const double* f,
                          int phase_mask, double* slack, double* dual, double* residual, double* cmpl, double* cond, double* dslack,
                          double* ddual, double* lx, double* lu, double* lf, double* Qxx, double* Quu, double* Qqf, double* Qff,
                          const double* dx, const double* du, const double* df, double* steps, double* dg_dq_out, double* dg_df_out) {
  const bool floating = nv != nu;
  const int nq = floating ? nv + 1 : nv;
  std::vector<ContactType> types(ncontacts, contact_dim == 3 ? ContactType::PointContact : ContactType::SurfaceContact);
  Robot robot(nv, nu, types);
  Eigen::VectorXd qmin(nu), qmax(nu), vmax(nu), umax(nu);
  for (int i = 0; i < nu; ++i) qmin(i) = limits[i], qmax(i) = limits[nu + i], vmax(i) = limits[2 * nu + i], umax(i) = limits[3 * nu + i];
  robot.setJointLimits(qmin, qmax, vmax, umax);
  for (int c = 0; c < ncontacts; ++c) {
    Eigen::Matrix3d R;
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) R(r, k) = frame_R[9 * c + 3 * r + k];
    Eigen::MatrixXd J(6, nv);
    for (int j = 0; j < nv; ++j)
      for (int r = 0; r < 6; ++r) J(r, j) = frame_Jlocal[(size_t)c * 6 * nv + r + 6 * j];
    robot.setFrameKinematics(c, R, J);
  }
  auto constraints = std::make_shared<Constraints>(barrier, tau);
  constraints->add("joint_position_lower", std::make_shared<JointPositionLowerLimit>(robot));
  constraints->add("joint_position_upper", std::make_shared<JointPositionUpperLimit>(robot));
  constraints->add("joint_velocity_lower", std::make_shared<JointVelocityLowerLimit>(robot));
  constraints->add("joint_velocity_upper", std::make_shared<JointVelocityUpperLimit>(robot));
  constraints->add("joint_torques_lower", std::make_shared<JointTorquesLowerLimit>(robot));
  constraints->add("joint_torques_upper", std::make_shared<JointTorquesUpperLimit>(robot));
  const int cone_rows = cone_kind == 1 ? 5 : (cone_kind == 2 ? 17 : 0);
  if (cone_kind == 1) {
    constraints->add("friction_cone", std::make_shared<FrictionCone>(robot));
    constraints->add("impact_friction_cone", std::make_shared<ImpactFrictionCone>(robot));
  } else if (cone_kind == 2) {
    constraints->add("contact_wrench_cone", std::make_shared<ContactWrenchCone>(robot, X, Y));
    constraints->add("impact_wrench_cone", std::make_shared<ImpactWrenchCone>(robot, X, Y));   // the impact-level twin (impact grids)
  }
  // contact status of the grid point
  ContactStatus cs = robot.createContactStatus();
  ImpactStatus is = robot.createImpactStatus();
  int dimf = 0;
  for (int c = 0; c < ncontacts; ++c) {
    const bool on = (active >> c) & 1u;
    if (on) dimf += contact_dim;
    if (impact) {
      if (on) is.activateImpact(c);
      is.setFrictionCoefficient(c, mu[c]);
    } else {
      if (on) cs.activateContact(c);
      cs.setFrictionCoefficient(c, mu[c]);
    }
    if (surf_rot) {
      Eigen::Matrix3d R;
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) R(r, k) = surf_rot[9 * c + 3 * r + k];
      if (impact) is.setContactPlacement(c, Eigen::Vector3d::Zero(), R);
      else cs.setContactPlacement(c, Eigen::Vector3d::Zero(), R);
    }
  }
  SplitSolution s(robot);
  if (impact) s.setContactStatus(is);
  else s.setContactStatus(cs);
  for (int i = 0; i < nq; ++i) s.q(i) = q[i];
  for (int i = 0; i < nv; ++i) s.v(i) = v[i];
  for (int i = 0; i < nu; ++i) s.u(i) = u[i];
  for (int c = 0; c < ncontacts; ++c)
    for (int k = 0; k < 6; ++k) s.f[c](k) = k < contact_dim ? f[6 * c + k] : 0.0;
  s.set_f_stack();
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  SplitDirection d(robot);
  km.setContactDimension(dimf), kr.setContactDimension(dimf), d.setContactDimension(dimf);
  km.setZero(), kr.setZero();
  const int nx = 2 * nv;
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nx; ++i) km.Qxx(i, j) = Qxx[i + (size_t)j * nx];
  for (int j = 0; j < nu; ++j)
    for (int i = 0; i < nu; ++i) km.Quu(i, j) = Quu[i + (size_t)j * nu];
  for (int j = 0; j < dimf; ++j) {
    for (int i = 0; i < nv; ++i) km.Qqf()(i, j) = Qqf[i + (size_t)j * nv];
    for (int i = 0; i < dimf; ++i) km.Qff()(i, j) = Qff[i + (size_t)j * dimf];
  }
  for (int i = 0; i < nx; ++i) kr.lx(i) = lx[i];
  for (int i = 0; i < nu; ++i) kr.lu(i) = lu[i];
  for (int i = 0; i < dimf; ++i) kr.lf()(i) = lf[i];
  ConstraintsData data = constraints->createConstraintsData(robot, impact ? -1 : time_stage);
  // component data <-> the flat row arrays
  std::vector<ConstraintComponentData*> comp;
  std::vector<int> rows;
  for (auto& c : data.position_level_data) comp.push_back(&c), rows.push_back(nu);
  for (auto& c : data.velocity_level_data) comp.push_back(&c), rows.push_back(nu);
  int k = 0;
  for (auto& c : data.acceleration_level_data) comp.push_back(&c), rows.push_back(k++ < 2 ? nu : cone_rows * ncontacts);
  for (auto& c : data.impact_level_data) comp.push_back(&c), rows.push_back(cone_rows * ncontacts);
  // the impact-level cone shares the cone rows of the flat arrays (a grid point is either an impact grid or not)
  auto offset_of = [&](size_t ci) {
    int o = 0;
    for (size_t t = 0; t < ci && t < 6 + (cone_kind ? 1u : 0u); ++t) o += rows[t];
    return ci < 6 + (cone_kind ? 1u : 0u) ? o : 6 * nu;
  };
  for (size_t ci = 0; ci < comp.size(); ++ci) {
    const int o = offset_of(ci);
    load(*comp[ci], slack + o, dual + o, residual + o, cmpl + o, rows[ci]);
  }
  if (phase_mask & 1) {
    if (impact) constraints->setSlackAndDual(robot, is, data, s);
    else constraints->setSlackAndDual(robot, cs, data, s);
  }
  if (phase_mask & 2) {
    if (impact) constraints->linearizeConstraints(robot, is, data, s, kr);
    else constraints->linearizeConstraints(robot, cs, data, s, kr);
  }
  if (phase_mask & 4) {
    if (impact) constraints->condenseSlackAndDual(is, data, km, kr);
    else constraints->condenseSlackAndDual(cs, data, km, kr);
  }
  if (phase_mask & 8) {
    for (int i = 0; i < nx; ++i) d.dx(i) = dx[i];
    for (int i = 0; i < nu; ++i) d.du(i) = du[i];
    for (int i = 0; i < dimf; ++i) d.df()(i) = df[i];
    if (impact) constraints->expandSlackAndDual(is, data, d);
    else constraints->expandSlackAndDual(cs, data, d);
    steps[0] = constraints->maxSlackStepSize(data);
    steps[1] = constraints->maxDualStepSize(data);
  }
  const bool active_level[4] = {data.isPositionLevelValid(), data.isVelocityLevelValid(), data.isAccelerationLevelValid(), data.isImpactLevelValid()};
  for (size_t ci = 0; ci < comp.size(); ++ci) {
    const int level = ci < 2 ? 0 : (ci < 4 ? 1 : (ci < 6 + (cone_kind ? 1u : 0u) ? 2 : 3));
    if (!active_level[level]) continue;   // rows the stage mask switches off keep what the caller passed
    const int o = offset_of(ci);
    store(*comp[ci], slack + o, dual + o, residual + o, cmpl + o, cond + o, dslack + o, ddual + o, rows[ci]);
    if (cone_kind == 1 && ci >= 6 && (phase_mask & 2)) {   // FrictionCone: J[i] = dg_dq (5 x nv), J[n + i] = dg_df (5 x 3)
      for (int c = 0; c < ncontacts; ++c) {
        for (int j = 0; j < nv; ++j)
          for (int r = 0; r < 5; ++r) dg_dq_out[(size_t)c * 5 * nv + r + 5 * j] = comp[ci]->J[c](r, j);
        for (int j = 0; j < 3; ++j)
          for (int r = 0; r < 5; ++r) dg_df_out[(size_t)c * 15 + r + 5 * j] = comp[ci]->J[ncontacts + c](r, j);
      }
    }
  }
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nx; ++i) Qxx[i + (size_t)j * nx] = km.Qxx(i, j);
  for (int j = 0; j < nu; ++j)
    for (int i = 0; i < nu; ++i) Quu[i + (size_t)j * nu] = km.Quu(i, j);
  for (int j = 0; j < dimf; ++j) {
    for (int i = 0; i < nv; ++i) Qqf[i + (size_t)j * nv] = km.Qqf()(i, j);
    for (int i = 0; i < dimf; ++i) Qff[i + (size_t)j * dimf] = km.Qff()(i, j);
  }
  for (int i = 0; i < nx; ++i) lx[i] = kr.lx(i);
  for (int i = 0; i < nu; ++i) lu[i] = kr.lu(i);
  for (int i = 0; i < dimf; ++i) lf[i] = kr.lf()(i);
  return 0;
}

// JointAccelerationLowerLimit / JointAccelerationUpperLimit (src/constraints/joint_acceleration_{lower,upper}_limit.cpp) inside the
// reference's Constraints object, one grid point.  Rows: lower limits (nu), then upper limits (nu).  phase_mask as above.
// Qaa_diag / la: SplitKKTMatrix::Qaa.diagonal() / SplitKKTResidual::la (nv each, in / out); a, da: nv each.
int ref_accel_limits_stage(int nv, int nu, int time_stage, int impact, const double* amin, const double* amax, double barrier, double tau,
                           const double* a, int phase_mask, double* slack, double* dual, double* residual, double* cmpl, double* cond,
                           double* dslack, double* ddual, double* Qaa_diag, double* la, const double* da, double* steps) {
  Robot robot(nv, nu, std::vector<ContactType>());
  Eigen::VectorXd lo(nu), hi(nu);
  for (int i = 0; i < nu; ++i) lo(i) = amin[i], hi(i) = amax[i];
  auto constraints = std::make_shared<Constraints>(barrier, tau);
  constraints->add("joint_acceleration_lower", std::make_shared<JointAccelerationLowerLimit>(robot, lo));
  constraints->add("joint_acceleration_upper", std::make_shared<JointAccelerationUpperLimit>(robot, hi));
  ContactStatus cs = robot.createContactStatus();
  ImpactStatus is = robot.createImpactStatus();
  SplitSolution s(robot);
  if (impact) s.setContactStatus(is);
  else s.setContactStatus(cs);
  for (int i = 0; i < nv; ++i) s.a(i) = a[i];
  SplitKKTMatrix km(robot);
  SplitKKTResidual kr(robot);
  SplitDirection d(robot);
  km.setZero(), kr.setZero();
  for (int i = 0; i < nv; ++i) km.Qaa(i, i) = Qaa_diag[i], kr.la(i) = la[i];
  ConstraintsData data = constraints->createConstraintsData(robot, impact ? -1 : time_stage);
  if (!data.isAccelerationLevelValid()) return 1;   // impact grids carry no acceleration-level rows (constraints_data.cpp:20-45)
  if (data.acceleration_level_data.size() != 2) return -1;
  for (int c = 0; c < 2; ++c) load(data.acceleration_level_data[c], slack + c * nu, dual + c * nu, residual + c * nu, cmpl + c * nu, nu);
  if (phase_mask & 1) constraints->setSlackAndDual(robot, cs, data, s);
  if (phase_mask & 2) constraints->linearizeConstraints(robot, cs, data, s, kr);
  if (phase_mask & 4) constraints->condenseSlackAndDual(cs, data, km, kr);
  if (phase_mask & 8) {
    for (int i = 0; i < nv; ++i) d.da()(i) = da[i];
    constraints->expandSlackAndDual(cs, data, d);
    steps[0] = constraints->maxSlackStepSize(data);
    steps[1] = constraints->maxDualStepSize(data);
  }
  for (int c = 0; c < 2; ++c)
    store(data.acceleration_level_data[c], slack + c * nu, dual + c * nu, residual + c * nu, cmpl + c * nu, cond + c * nu, dslack + c * nu,
          ddual + c * nu, nu);
  for (int i = 0; i < nv; ++i) Qaa_diag[i] = km.Qaa(i, i), la[i] = kr.la(i);
  return 0;
}

}  // extern "C"

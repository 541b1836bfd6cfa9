/*
 * rtoc_oracle_aba_cs.c -- COMPLEX-STEP derivatives of the second formulation (rtoc_oracle_aba.c; TEST INFRASTRUCTURE ONLY).
 *
 * dFD/dq and dFD/dv of the articulated-body algorithm, by compiling rtoc_oracle_aba.c once more with complex scalars
 * (d r / d x_j = Im r(x + i h e_j) / h, h = 1e-30: exact to rounding, Squire & Trapp 1998).  Configuration perturbations are
 * taken on the manifold, q (+) i h e_j, written out here to first order in h (second-order terms are 1e-60: they do not exist in
 * double precision), independently of rtoc_oracle_rbd.c's retraction: a revolute angle moves by i h; a free-flyer translates by
 * R (i h e_k) or turns by the unit quaternion [i h e_k / 2, 1] applied on the right (body-frame increments, as
 * pinocchio::integrate defines them).
 * With these, the RNEA derivatives the device computes (Robot::RNEADerivatives, include/robotoc/robot/robot.hxx:548-575) follow
 * from the OTHER side of the dynamics:  ID(q, v, FD(q, v, tau)) = tau  =>  dID/dq = -M dFD/dq,  dID/dv = -M dFD/dv,  dID/da = M.
 */
#include <complex.h>
#include <math.h>
#include <string.h>
#include <tgmath.h>
#undef I

#include "../include/rtoc_robot.h"

typedef double _Complex cplx;

typedef struct orc_acs_model {
  int njoints, nq, nv, ncontacts;
  int parent[RTOC_MAX_JOINTS], type[RTOC_MAX_JOINTS], idx_q[RTOC_MAX_JOINTS], idx_v[RTOC_MAX_JOINTS];
  cplx placement_R[RTOC_MAX_JOINTS][9], placement_p[RTOC_MAX_JOINTS][3], axis[RTOC_MAX_JOINTS][3], mass[RTOC_MAX_JOINTS];
  cplx com[RTOC_MAX_JOINTS][3], inertia[RTOC_MAX_JOINTS][9];
  int contact_type[RTOC_MAX_CONTACTS], contact_parent[RTOC_MAX_CONTACTS];
  cplx contact_R[RTOC_MAX_CONTACTS][9], contact_p[RTOC_MAX_CONTACTS][3], contact_kp[RTOC_MAX_CONTACTS], contact_kd[RTOC_MAX_CONTACTS];
  cplx gravity[3];
} orc_acs_model;

#define ABA_RE(x) creal(x)
#define orc_aba_forward_dynamics orc_acs_forward_dynamics
#define orc_aba_crba orc_acs_crba
#define rtoc_robot_model orc_acs_model
#define double double _Complex
#include "rtoc_oracle_aba.c"
#undef double
#undef rtoc_robot_model

static void to_c(const double* x, int n, cplx* out) {
  for (int i = 0; i < n; ++i) out[i] = x ? x[i] : 0.0;
}
static void model_to_c(const rtoc_robot_model* m, orc_acs_model* c) {
  memset(c, 0, sizeof *c);
  c->njoints = m->njoints, c->nq = m->nq, c->nv = m->nv, c->ncontacts = m->ncontacts;
  memcpy(c->parent, m->parent, sizeof m->parent), memcpy(c->type, m->type, sizeof m->type);
  memcpy(c->idx_q, m->idx_q, sizeof m->idx_q), memcpy(c->idx_v, m->idx_v, sizeof m->idx_v);
  memcpy(c->contact_type, m->contact_type, sizeof m->contact_type), memcpy(c->contact_parent, m->contact_parent, sizeof m->contact_parent);
  to_c(&m->placement_R[0][0], RTOC_MAX_JOINTS * 9, &c->placement_R[0][0]), to_c(&m->placement_p[0][0], RTOC_MAX_JOINTS * 3, &c->placement_p[0][0]);
  to_c(&m->axis[0][0], RTOC_MAX_JOINTS * 3, &c->axis[0][0]), to_c(m->mass, RTOC_MAX_JOINTS, c->mass);
  to_c(&m->com[0][0], RTOC_MAX_JOINTS * 3, &c->com[0][0]), to_c(&m->inertia[0][0], RTOC_MAX_JOINTS * 9, &c->inertia[0][0]);
  to_c(&m->contact_R[0][0], RTOC_MAX_CONTACTS * 9, &c->contact_R[0][0]), to_c(&m->contact_p[0][0], RTOC_MAX_CONTACTS * 3, &c->contact_p[0][0]);
  to_c(m->contact_kp, RTOC_MAX_CONTACTS, c->contact_kp), to_c(m->contact_kd, RTOC_MAX_CONTACTS, c->contact_kd), to_c(m->gravity, 3, c->gravity);
}

/* q (+) i h e_j to first order in h */
static void perturb_q(const rtoc_robot_model* m, const double* q, int j, double h, cplx* qp) {
  to_c(q, m->nq, qp);
  for (int i = 0; i < m->njoints; ++i) {
    const int iv = m->idx_v[i], iq = m->idx_q[i];
    if (m->type[i] != RTOC_JOINT_FREE_FLYER) {
      if (iv == j) qp[iq] += h * _Complex_I;
      continue;
    }
    if (j < iv || j >= iv + 6) continue;
    const int k = j - iv;
    const double x = q[iq + 3], y = q[iq + 4], z = q[iq + 5], w = q[iq + 6];
    if (k < 3) {
      /* translation by R e_k: column k of the rotation of the unit quaternion */
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                           2 * (y * z - x * w),     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
      for (int r = 0; r < 3; ++r) qp[iq + r] += h * _Complex_I * R[3 * r + k];
    } else {
      /* q (x) [e/2 * i h, 1]: (v, w)(v', w') = (w v' + w' v + v x v', w w' - v . v') */
      double e[3] = {0, 0, 0};
      e[k - 3] = 0.5;
      const double vx[3] = {y * e[2] - z * e[1], z * e[0] - x * e[2], x * e[1] - y * e[0]};
      qp[iq + 3] += h * _Complex_I * (w * e[0] + vx[0]);
      qp[iq + 4] += h * _Complex_I * (w * e[1] + vx[1]);
      qp[iq + 5] += h * _Complex_I * (w * e[2] + vx[2]);
      qp[iq + 6] += h * _Complex_I * (-(x * e[0] + y * e[1] + z * e[2]));
    }
  }
}

/* dadq, dadv: nv x nv column-major (leading dimension nv): d FD / d q (tangent), d FD / d v at fixed tau and contact forces */
void orc_aba_linearize_cs(const rtoc_robot_model* mr, const double* q, const double* v, const double* tau, const double* fstack, int nf,
                          unsigned active, double* dadq, double* dadv) {
  const double h = 1e-30;
  orc_acs_model mc;
  model_to_c(mr, &mc);
  const int nv = mr->nv;
  cplx qc[RTOC_MAX_JOINTS + 8], vc[RTOC_MAX_JOINTS + 6], tc[RTOC_MAX_JOINTS + 6], fc[6 * RTOC_MAX_CONTACTS], res[RTOC_MAX_JOINTS + 6];
  memset(fc, 0, sizeof fc);
  to_c(tau, nv, tc), to_c(fstack, nf, fc);
  for (int j = 0; j < nv; ++j) {
    perturb_q(mr, q, j, h, qc);
    to_c(v, nv, vc);
    orc_acs_forward_dynamics(&mc, qc, vc, tc, fc, active, res);
    for (int i = 0; i < nv; ++i) dadq[i + (size_t)j * nv] = cimag(res[i]) / h;
    to_c(q, mr->nq, qc);
    vc[j] += h * _Complex_I;
    orc_acs_forward_dynamics(&mc, qc, vc, tc, fc, active, res);
    for (int i = 0; i < nv; ++i) dadv[i + (size_t)j * nv] = cimag(res[i]) / h;
  }
}

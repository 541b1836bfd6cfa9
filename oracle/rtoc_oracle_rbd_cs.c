/*
 * rtoc_oracle_rbd_cs.c -- COMPLEX-STEP derivatives of the rigid-body restatement (TEST INFRASTRUCTURE ONLY, see rtoc_oracle.c).
 *
 * The second witness of rtoc_linearize_contact_dynamics' derivative blocks.  The device computes dID/d(q, v, a) and
 * dC/d(q, v, a) by an analytical tangent walk (robotoc_amd/csrc/rigid_body.hpp); rtoc_oracle_rbd.c only evaluates ID and C, and its
 * central differences bound the derivative at ~1e-9.  Here the SAME evaluation is compiled once more with complex scalars
 * (every `double` of rtoc_oracle_rbd.c becomes `double _Complex`; <tgmath.h> maps sqrt / sin / cos / acos onto their analytic
 * continuations; branches read the real part or the modulus through ORC_RE / ORC_MAG), and
 *     d r / d x_j = Im r(x + i h e_j) / h,   h = 1e-30
 * is exact to rounding: no subtraction, no step-size trade-off (Squire & Trapp 1998).  Configuration perturbations go through the
 * manifold retraction q (+) i h e_j (orc_rbd_integrate: SE(3) exponential on the free-flyer root), as pinocchio's derivatives are
 * defined.  What this does NOT remove: evaluation and witness still share the restated recursion -- PARITY UNPINNED as stated in
 * rtoc_oracle_rbd.c (Pinocchio is not in the image); tools/record_pinocchio_fixture.py is the recorder for a machine that has it.
 */
#include <complex.h>
#include <math.h>
#include <string.h>
#include <tgmath.h>
#undef I /* rtoc_oracle_rbd.c names an inertia tensor I; the imaginary unit is _Complex_I below */

#include "../include/rtoc_robot.h" /* the model tables stay real */

typedef double _Complex cplx;

/* rtoc_robot_model with complex reals: what rtoc_oracle_rbd.c sees as its model in this build */
typedef struct orc_cs_model {
  int njoints, nq, nv, ncontacts;
  int parent[RTOC_MAX_JOINTS], type[RTOC_MAX_JOINTS], idx_q[RTOC_MAX_JOINTS], idx_v[RTOC_MAX_JOINTS];
  cplx placement_R[RTOC_MAX_JOINTS][9], placement_p[RTOC_MAX_JOINTS][3], axis[RTOC_MAX_JOINTS][3], mass[RTOC_MAX_JOINTS];
  cplx com[RTOC_MAX_JOINTS][3], inertia[RTOC_MAX_JOINTS][9];
  int contact_type[RTOC_MAX_CONTACTS], contact_parent[RTOC_MAX_CONTACTS];
  cplx contact_R[RTOC_MAX_CONTACTS][9], contact_p[RTOC_MAX_CONTACTS][3], contact_kp[RTOC_MAX_CONTACTS], contact_kd[RTOC_MAX_CONTACTS];
  cplx gravity[3];
} orc_cs_model;

#define ORC_RE(x) creal(x)
#define ORC_MAG(x) cabs(x)
#define orc_rbd_log6 orc_cs_rbd_log6
#define orc_rbd_exp6 orc_cs_rbd_exp6
#define orc_se3_integrate orc_cs_se3_integrate
#define orc_se3_difference orc_cs_se3_difference
#define orc_rbd_integrate orc_cs_rbd_integrate
#define orc_rbd_eval_ex orc_cs_rbd_eval_ex
#define orc_rbd_eval orc_cs_rbd_eval
#define orc_rbd_linearize_fd_ex orc_cs_rbd_linearize_fd_ex
#define orc_rbd_linearize_fd orc_cs_rbd_linearize_fd
#define orc_rbd_mass_matrix_world orc_cs_rbd_mass_matrix_world
#define orc_rbd_energy orc_cs_rbd_energy
#define orc_rbd_momentum_world orc_cs_rbd_momentum_world
#define orc_rbd_contact_placement orc_cs_rbd_contact_placement
#define orc_rbd_contact_position orc_cs_rbd_contact_position
#define rtoc_robot_model orc_cs_model
#define double double _Complex
#include "rtoc_oracle_rbd.c"
#undef double
#undef rtoc_robot_model

static void to_c(const double* x, int n, cplx* out) {
  for (int i = 0; i < n; ++i) out[i] = x ? x[i] : 0.0;
}
static void model_to_c(const rtoc_robot_model* m, orc_cs_model* c) {
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
/* Dq, Dv, Da: [nv + active rows] x nv column-major with leading dimension ld, as orc_rbd_linearize_fd_ex */
void orc_rbd_linearize_cs(const rtoc_robot_model* mr, int impact, const double* q, const double* v, const double* a, const double* fstack,
                          int nf, const double* u, int nu, unsigned active, const double* pref, const double* rref, double* Dq,
                          double* Dv, double* Da, int ld) {
  const double h = 1e-30;
  orc_cs_model mc;
  model_to_c(mr, &mc);
  const orc_cs_model* m = &mc;
  const int nv = m->nv, n = nv + active_rows(m, active);
  cplx qc[RTOC_MAX_JOINTS + 8], vc[RTOC_MAX_JOINTS + 6], ac[RTOC_MAX_JOINTS + 6], uc[RTOC_MAX_JOINTS + 6], fc[6 * RTOC_MAX_CONTACTS];
  cplx pc[3 * RTOC_MAX_CONTACTS], rc[9 * RTOC_MAX_CONTACTS], e[RTOC_MAX_JOINTS + 6], qp[RTOC_MAX_JOINTS + 8], xp[RTOC_MAX_JOINTS + 6];
  cplx res[RTOC_MAX_JOINTS + 6 * RTOC_MAX_CONTACTS];
  memset(uc, 0, sizeof uc), memset(fc, 0, sizeof fc);
  to_c(q, m->nq, qc), to_c(v, nv, vc), to_c(a, nv, ac), to_c(u, nu, uc), to_c(fstack, nf, fc);
  to_c(pref, 3 * m->ncontacts, pc);
  to_c(rref, 9 * m->ncontacts, rc);
  for (int j = 0; j < nv; ++j) {
    memset(e, 0, sizeof e);
    e[j] = h * _Complex_I;
    orc_cs_rbd_integrate(m, qc, e, 1.0, qp);
    orc_cs_rbd_eval_ex(m, impact, qp, vc, ac, fc, uc, active, pc, rref ? rc : NULL, res);
    for (int i = 0; i < n; ++i) Dq[i + (size_t)j * ld] = cimag(res[i]) / h;
    memcpy(xp, vc, sizeof(cplx) * nv);
    xp[j] += h * _Complex_I;
    orc_cs_rbd_eval_ex(m, impact, qc, xp, ac, fc, uc, active, pc, rref ? rc : NULL, res);
    for (int i = 0; i < n; ++i) Dv[i + (size_t)j * ld] = cimag(res[i]) / h;
    memcpy(xp, ac, sizeof(cplx) * nv);
    xp[j] += h * _Complex_I;
    orc_cs_rbd_eval_ex(m, impact, qc, vc, xp, fc, uc, active, pc, rref ? rc : NULL, res);
    for (int i = 0; i < n; ++i) Da[i + (size_t)j * ld] = cimag(res[i]) / h;
  }
}

// rigid_body.hpp -- batched linearisation of the contact / impact dynamics.
//
// Replaces, per (instance, grid point): Robot::RNEA / RNEADerivatives / RNEAImpact / RNEAImpactDerivatives (reference
// include/robotoc/robot/robot.hxx:524-624, thin wrappers over pinocchio::rnea / computeRNEADerivatives), setContactForces
// (:455-517), computeBaumgarteResidual / Derivatives (:291-360 -> point_contact.hxx:14-83) and computeImpactVelocity
// Residual / Derivatives (point_contact.hxx:84-117), i.e. linearizeContactDynamics (src/dynamics/contact_dynamics.cpp:12-33)
// and linearizeImpactDynamics (impact_dynamics.cpp:8-27) up to the multiplier terms.
//
// Mapping: one wave per grid point, one LANE per tangent direction (dof j, kind in {q, v, a}): 21 dofs per pass.  Every
// lane walks the kinematic tree depth first and carries the forward-mode derivative of Featherstone's recursive
// Newton-Euler along its direction:
//     v_i = X v_p + S vq                      dv_i = X dv_p - [own q] S_k x (X v_p) + [own v] S_k
//     a_i = X a_p + S aq + v_i x S vq         da_i = X da_p - [own q] S_k x (X a_p) + [own a] S_k + dv_i x S vq + [own v] v_i x S_k
//     f_i = Y (a_i + g_i) + v_i x* Y v_i - fext_i
//     f_p += X* f_i                           df_p += X* df_i + [own q] X* (S_k x* f_i)          tau_i = S^T f_i
// (d(X m)/dq_k = -S_k x (X m), d(X* f)/dq_k = X* (S_k x* f) for the joint's own coordinate k; q perturbed on the
// manifold, q (+) dq: local translation / rotation of a free-flyer root.)  The VALUES are the same in every lane and
// live once per tree level in LDS; the TANGENTS (dv, da, dg, df: 21 doubles) live per lane and level in LDS,
// lane-strided (conflict-free).  Only the current root-to-body path is live: LDS = levels x (21 x 64 + ~60) doubles.
// The contact rows ride on the visit of the body that carries the frame; d(position)/dq = R_of * (frame Jacobian), whose
// column j is the v-tangent of the frame velocity -- held by the neighbouring lane (point_contact.hxx:78-80).
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc_robot.h"

#ifndef RTOC_RBD_WAVES
#define RTOC_RBD_WAVES
#endif

namespace rtoc {
namespace rbd {

// per-joint / per-contact constants as the kernel reads them, packed by rtoc_set_robot_model: one coalesced copy into
// LDS per grid point instead of ~30 dependent L2 round trips per visited body
constexpr int JP = 32;  // doubles per joint: R 9, p 3, axis 3, mass 1, com 3, I 9 (28), type, idx_q, idx_v, depth
constexpr int CP = 16;  // doubles per contact: R 9, p 3, kp, kd, parent, type
struct DevModel {
  rtoc_robot_model m;
  int depth[RTOC_MAX_JOINTS];
  int nlevels;
  // storage plan of the tangent walk, per body: bit 0 = its parent is the body visited just before it (the parent's forward
  // tangents are still in registers), bit 1 = leaf (its force tangent is closed from registers), bits 4-7 = 1 + the LDS slot
  // its own forward tangents are kept in (bodies with two or more children; 0 = none), bits 8-11 = 1 + its parent's slot.
  // Slots are numbered by the count of branching ancestors: two bodies with the same count are never open at once.
  int walk[RTOC_MAX_JOINTS];
  int nbranch;
  // passes of the tangent walk: pass p carries the dofs [p dpp, (p + 1) dpp), three lanes each; pass_bodies[p] = the bodies
  // a direction of the pass can move or load (bit i: some dof of the pass sits on the path root -> i or in the subtree of i);
  // the other bodies are skipped by the whole wave (their columns are zero)
  int dpp, npass;
  unsigned long long pass_bodies[RTOC_MAX_JOINTS + 8];
  int pass_nvisit[RTOC_MAX_JOINTS + 8];                          // the same as lists, in depth-first order
  int pass_visit[RTOC_MAX_JOINTS + 8][RTOC_MAX_JOINTS];       // (ints: read through the scalar cache)
  double joint[RTOC_MAX_JOINTS][JP];
  double contact[RTOC_MAX_CONTACTS][CP];
  // per dof: the body it moves and its angular axis in that body's frame (zero for the linear dofs of a free-flyer);
  // per contact: the dofs on the path from the root to the contact's body (bit j): everything the world-aligned angular
  // Jacobian column of a contact frame needs besides the bodies' world rotations (contact_cone_vals_kernel)
  int dof_body[RTOC_MAX_JOINTS + 8];
  double dof_axis[RTOC_MAX_JOINTS + 8][3];
  unsigned long long contact_dofs[RTOC_MAX_CONTACTS];
};
// nbranch of a model (the number of LDS slots the walk needs for forward tangents): 1 + the largest count of branching
// ancestors of a branching body, 0 for a chain
inline int walk_plan(const rtoc_robot_model& m, int* walk) {
  int nchild[RTOC_MAX_JOINTS] = {}, slot[RTOC_MAX_JOINTS], nbranch = 0;
  for (int i = 1; i < m.njoints; ++i)
    if (m.parent[i] >= 0 && m.parent[i] < i) nchild[m.parent[i]]++;
  for (int i = 0; i < m.njoints; ++i) {
    const int par = (i > 0 && m.parent[i] >= 0 && m.parent[i] < i) ? m.parent[i] : -1;
    // slot[i]: the slot a branching body i would use = the number of branching bodies above it
    slot[i] = par < 0 ? 0 : slot[par] + (nchild[par] >= 2 ? 1 : 0);
    const int own = nchild[i] >= 2 ? slot[i] + 1 : 0;
    const int pslot = (par >= 0 && nchild[par] >= 2) ? slot[par] + 1 : 0;
    if (own > nbranch) nbranch = own;
    if (walk) walk[i] = ((par >= 0 && par == i - 1) ? 1 : 0) | (nchild[i] == 0 ? 2 : 0) | (own << 4) | (pslot << 8);
  }
  return nbranch;
}
inline void plan_passes(DevModel* h, int forced_dpp = 0);
inline void pack_model(DevModel* h) {
  const rtoc_robot_model& m = h->m;
  for (int i = 0; i < RTOC_MAX_JOINTS; ++i) h->walk[i] = 0;
  h->nbranch = walk_plan(m, h->walk);
  plan_passes(h);
  for (int i = 0; i < m.njoints; ++i) {
    double* o = h->joint[i];
    for (int k = 0; k < 9; ++k) o[k] = m.placement_R[i][k], o[19 + k] = m.inertia[i][k];
    for (int k = 0; k < 3; ++k) o[9 + k] = m.placement_p[i][k], o[12 + k] = m.axis[i][k], o[16 + k] = m.com[i][k];
    o[15] = m.mass[i];
    o[28] = m.type[i], o[29] = m.idx_q[i], o[30] = m.idx_v[i], o[31] = h->depth[i];
  }
  for (int j = 0; j < RTOC_MAX_JOINTS + 8; ++j) h->dof_body[j] = 0, h->dof_axis[j][0] = h->dof_axis[j][1] = h->dof_axis[j][2] = 0.0;
  for (int i = 0; i < m.njoints; ++i) {
    const int ndof = m.type[i] == RTOC_JOINT_FREE_FLYER ? 6 : 1;
    for (int k = 0; k < ndof; ++k) {
      const int j = m.idx_v[i] + k;
      if (j < 0 || j >= RTOC_MAX_JOINTS + 8) continue;
      h->dof_body[j] = i;
      if (ndof == 6) {
        if (k >= 3) h->dof_axis[j][k - 3] = 1.0;
      } else {
        for (int t = 0; t < 3; ++t) h->dof_axis[j][t] = m.axis[i][t];
      }
    }
  }
  for (int c = 0; c < m.ncontacts; ++c) {
    unsigned long long mask = 0;
    for (int i = m.contact_parent[c]; i >= 0 && i < m.njoints; i = m.parent[i]) {
      const int ndof = m.type[i] == RTOC_JOINT_FREE_FLYER ? 6 : 1;
      for (int k = 0; k < ndof; ++k)
        if (m.idx_v[i] + k >= 0 && m.idx_v[i] + k < 64) mask |= 1ull << (m.idx_v[i] + k);
      if (m.parent[i] == i) break;
    }
    h->contact_dofs[c] = mask;
  }
  for (int c = 0; c < m.ncontacts; ++c) {
    double* o = h->contact[c];
    for (int k = 0; k < 9; ++k) o[k] = m.contact_R[c][k];
    for (int k = 0; k < 3; ++k) o[9 + k] = m.contact_p[c][k];
    o[12] = m.contact_kp[c], o[13] = m.contact_kd[c], o[14] = m.contact_parent[c], o[15] = m.contact_type[c];
  }
}

struct V3 {
  double x, y, z;
};
struct SV {  // spatial motion or force: linear, angular
  V3 l, a;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ SV operator+(SV a, SV b) { return SV{a.l + b.l, a.a + b.a}; }
__device__ __forceinline__ SV operator-(SV a, SV b) { return SV{a.l - b.l, a.a - b.a}; }
__device__ __forceinline__ SV sv0() { return SV{mk(0, 0, 0), mk(0, 0, 0)}; }

struct M3 {  // row-major
  double m[9];
};
__device__ __forceinline__ V3 mul(const M3& R, V3 v) {
  return mk(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
__device__ __forceinline__ V3 mulT(const M3& R, V3 v) {
  return mk(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
            R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z);
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
// child-frame coordinates of a parent-frame motion, X = (R, p)
__device__ __forceinline__ SV act_inv(const M3& R, V3 p, SV m) { return SV{mulT(R, m.l - cross(p, m.a)), mulT(R, m.a)}; }
// parent-frame coordinates of a child-frame force
__device__ __forceinline__ SV act_f(const M3& R, V3 p, SV f) {
  const V3 l = mul(R, f.l);
  return SV{l, mul(R, f.a) + cross(p, l)};
}
__device__ __forceinline__ SV mcross(SV v, SV m) { return SV{cross(v.a, m.l) + cross(v.l, m.a), cross(v.a, m.a)}; }   // v x m
__device__ __forceinline__ SV fcross(SV v, SV f) { return SV{cross(v.a, f.l), cross(v.a, f.a) + cross(v.l, f.l)}; }   // v x* f
__device__ __forceinline__ SV inertia_mul(double mass, V3 c, const M3& I, SV v) {
  const V3 l = mass * (v.l - cross(c, v.a));
  return SV{l, mul(I, v.a) + cross(c, l)};
}
__device__ __forceinline__ M3 ldm3(const double* p) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.m[i] = p[i];
  return r;
}
__device__ __forceinline__ V3 ldv3(const double* p) { return mk(p[0], p[1], p[2]); }

// pinocchio::log6 of X = (R, p) and its derivative along the right perturbation X exp(twist) (what Jlog6 * twist is):
//   w = log3 R,  lin = p - w x p / 2 + beta w x (w x p),  beta(t) = 1/t^2 - cot(t/2) / (2 t)  (= the Jlog3 coefficient too)
//   dw = twist.a + w x twist.a / 2 + beta w x (w x twist.a),  dp = R twist.l,  dt = w.dw / t
__device__ __forceinline__ void log6_fwd(const M3& R, V3 p, SV twist, SV& val, SV& der) {
  const double tr = R.m[0] + R.m[4] + R.m[8];
  double c = 0.5 * (tr - 1.0);
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  const double th = acos(c);
  const double k = th < 1e-6 ? 0.5 + th * th / 12.0 : th / (2.0 * sin(th));
  const V3 w = mk(k * (R.m[7] - R.m[5]), k * (R.m[2] - R.m[6]), k * (R.m[3] - R.m[1]));
  const double t = sqrt(dot(w, w));
  double beta, dbeta;
  if (t < 1e-3) {
    beta = 1.0 / 12.0 + t * t / 720.0;
    dbeta = t / 360.0 + t * t * t / 7560.0;
  } else {
    const double h = 0.5 * t, ct = cos(h) / sin(h), cs2 = 1.0 / (sin(h) * sin(h));
    beta = 1.0 / (t * t) - ct / (2.0 * t);
    dbeta = -2.0 / (t * t * t) + ct / (2.0 * t * t) + cs2 / (4.0 * t);
  }
  const V3 wxp = cross(w, p), wxwxp = cross(w, wxp);
  val = SV{p - 0.5 * wxp + beta * wxwxp, w};
  const V3 ta = twist.a;
  const V3 dw = ta + 0.5 * cross(w, ta) + beta * cross(w, cross(w, ta));
  const V3 dp = mul(R, twist.l);
  const double dt = t > 1e-12 ? dot(w, dw) / t : 0.0;
  const V3 dlin = dp - 0.5 * (cross(dw, p) + cross(w, dp)) + (dbeta * dt) * wxwxp +
                  beta * (cross(dw, wxp) + cross(w, cross(dw, p)) + cross(w, cross(w, dp)));
  der = SV{dlin, dw};
}

// an int of the device model through the scalar cache: the model is read-only while a kernel runs, but the compiler cannot know
// (the kernel stores through other pointers) and would issue a vector load + v_readfirstlane on the walk's critical path
typedef const int __attribute__((address_space(4))) lin_const_int;
__device__ __forceinline__ int sload_int(const int* p) { return *(lin_const_int*)(unsigned long long)p; }

struct LinArgs {
  const DevModel* model;
  const double* sol;
  double* cdd;
  const rtoc_grid* grid;
  const unsigned* active;    // [nstages]
  const double* positions;   // [nstages][ncontacts][3] or nullptr
  const double* rotations;   // [nstages][ncontacts][9] or nullptr (surface contacts: desired rotation)
  int nstages, batch;
  int sol_stride, cdd_stride;
  int o_q, o_v, o_a, o_u, o_f;                 // RTOC_BUF_SOL field offsets
  int o_idc, o_didda, o_dcda, o_didcdqv;       // RTOC_BUF_CDD field offsets
  int ldv, nf_max;                             // leading dimensions of DIDCDQV / DCDA
  int nlevels, nbranch, dpp, nv, nq, njoints, ncontacts, nu;
  double gx, gy, gz;                           // gravity
  // multiplier terms of linearizeContactDynamics / linearizeImpactDynamics (kkt == nullptr: left out)
  double* kkt;
  int kkt_stride, o_lx, o_lu;                  // RTOC_BUF_KKT: lx = [lq; lv], lu
  int o_la, o_lf, o_lup;                       // RTOC_BUF_CDD: la (ldv on impact grids), lf, lu_passive
  int o_beta, o_mu, o_nup;                     // RTOC_BUF_SOL
  // UnconstrDynamics::linearizeUnconstrDynamics (unconstr_dynamics.cpp:52-64): the multiplier terms carry dt, and the
  // records follow the convention of rtoc_unconstr_condense (la lives in KKT.lu, lu in CDD.la)
  int unconstr;
  double scale;
  const double* vals;   // PRE: [batch * nstages][njoints][64] of the dynamics traversal (rbd_values_kernel)
  const double* vals2;  //      the same for the kinematics traversal of impact grids
};

// per-level storage in LDS
constexpr int VAL_DOUBLES = 64;  // R 9, p 3, oR 9, op 3, v 6, a 6, g 3, f 6, vpar 6, apar 6 -> 57, padded
constexpr int TAN_SLOTS = 21;    // dv 6, da 6, dg 3, df 6
constexpr int FWD_SLOTS = 15;    // dv, da, dg: read by the children only, so the deepest level keeps none
constexpr int DF_SLOTS = 6;
__host__ __device__ constexpr int lin_pad8(int n) { return (n + 7) & ~7; }
// lanes per tangent slot: three per dof of a pass plus a column the idle lanes share, even (quadrupeds: 54 -> 56; 21 dofs: 64)
__host__ __device__ constexpr int lin_lane_stride(int dpp) { return (3 * dpp + 2) & ~1; }
constexpr int LIN_MAX_DPP = 21;
// What decides the speed of this kernel is how many grid points a CU holds at once (the walk is one long dependent
// instruction stream per wave, issue-bound): only what the walk cannot carry in registers lives in LDS -- the forward tangents
// (dv, da, dg) of the bodies with two or more children (nbranch slots: a body whose parent was visited just before it takes
// them from registers) and the force tangents df of the open non-leaf levels (a leaf is closed from registers).  ANYmal:
// 1 slot + 3 levels = 22 KB (was 4 + 4 levels = 38 KB), iCub: 2 slots + 10 levels (was 11 + 11 = 128 KB); DESIGN.md 3.4.
// pre: the walk reads the values of the recursion from rbd_values_kernel (PRE): no q, v, a, f, u staging, and of the joint
// constants only axis .. depth (JP_PRE doubles from JP_PRE_OFF on) -- 20,000 B for ANYmal: EIGHT waves per CU (8 x 20,480 B).
constexpr int JP_PRE_OFF = 12, JP_PRE = JP - JP_PRE_OFF;
__host__ __device__ constexpr size_t lin_lds_bytes(int nlevels, int nbranch, int njoints, int ncontacts, int nv, int dpp, bool pre) {
  return sizeof(double) * ((size_t)nlevels * VAL_DOUBLES + (size_t)(nbranch * FWD_SLOTS + (nlevels > 1 ? nlevels - 1 : 0) * DF_SLOTS) * lin_lane_stride(dpp) +
                           (pre ? 0 : lin_pad8(nv + 1) + 3 * lin_pad8(nv) + lin_pad8(6 * ncontacts)) + lin_pad8(nv) + 2 * lin_pad8(6 * ncontacts) +
                           njoints * (pre ? JP_PRE : JP) + ncontacts * CP);
}

// bodies a pass with the dofs [j0, j1) has to visit
inline unsigned long long pass_body_mask(const rtoc_robot_model& m, int j0, int j1) {
  unsigned long long mask = 0;
  for (int b = 0; b < m.njoints; ++b) {
    const int ndof = m.type[b] == RTOC_JOINT_FREE_FLYER ? 6 : 1;
    if (m.idx_v[b] + ndof <= j0 || m.idx_v[b] >= j1) continue;   // no dof of body b in the pass
    for (int i = 0; i < m.njoints; ++i) {
      bool up = false, down = false;   // b above-or-at i; b below i
      for (int k = i; k >= 0; k = m.parent[k]) {
        if (k == b) up = true;
        if (m.parent[k] < 0 || m.parent[k] >= k) break;
      }
      for (int k = b; k >= 0; k = m.parent[k]) {
        if (k == i) down = true;
        if (m.parent[k] < 0 || m.parent[k] >= k) break;
      }
      if (up || down) mask |= 1ull << i;
    }
  }
  return mask;
}
// dofs per pass: what minimises (bodies visited over all passes) / (waves a CU holds).  The walk is one dependent instruction
// stream per wave, so a CU's rate is its resident waves -- set by the LDS of the per-lane tangents, i.e. by the lanes of a pass --
// over the visits per grid point: ANYmal one pass of 18 dofs (8 waves per CU), iCub 3 passes of 12 instead of 2 of 21.
// max_waves: what the registers of the kernel allow per CU (8; 4 with surface contacts: 256 VGPRs + AGPRs).
inline int choose_dofs_per_pass(const rtoc_robot_model& m, int nlevels, int nbranch, int max_waves) {
  const int dmax = m.nv < LIN_MAX_DPP ? m.nv : LIN_MAX_DPP, dmin = dmax < 6 ? dmax : 6;
  int best = dmax;
  double best_cost = 1e300;
  for (int d = dmax; d >= dmin && d >= 1; --d) {   // ties: the larger pass
    const size_t bytes = (lin_lds_bytes(nlevels, nbranch, m.njoints, m.ncontacts, m.nv, d, true) + 1279) / 1280 * 1280;
    int waves = (int)(160 * 1024 / bytes);
    waves = waves > max_waves ? max_waves : waves;
    if (waves < 1) continue;
    int visits = m.njoints;   // the first pass visits every body
    for (int j0 = d; j0 < m.nv; j0 += d) visits += __builtin_popcountll(pass_body_mask(m, j0, j0 + d < m.nv ? j0 + d : m.nv));
    const double cost = (double)visits / waves;
    if (cost < best_cost - 1e-9) best_cost = cost, best = d;
  }
  return best;
}
inline void plan_passes(DevModel* h, int forced_dpp) {
  const rtoc_robot_model& m = h->m;
  bool surf = false;
  for (int c = 0; c < m.ncontacts; ++c) surf = surf || m.contact_type[c] == RTOC_CONTACT_SURFACE;
  h->dpp = forced_dpp > 0 ? (forced_dpp < m.nv ? forced_dpp : (m.nv < LIN_MAX_DPP ? m.nv : LIN_MAX_DPP)) : choose_dofs_per_pass(m, h->nlevels, h->nbranch, surf ? 4 : 8);
  for (int p = 0; p < RTOC_MAX_JOINTS + 8; ++p) {
    h->pass_bodies[p] = 0, h->pass_nvisit[p] = 0;
    for (int i = 0; i < RTOC_MAX_JOINTS; ++i) h->pass_visit[p][i] = 0;
  }
  h->npass = (m.nv + h->dpp - 1) / h->dpp;
  for (int p = 0; p < h->npass; ++p) h->pass_bodies[p] = pass_body_mask(m, p * h->dpp, (p + 1) * h->dpp < m.nv ? (p + 1) * h->dpp : m.nv);
  h->pass_bodies[0] |= m.njoints >= 64 ? ~0ull : (1ull << m.njoints) - 1;   // the first pass writes the values of the contact rows: every body
  for (int p = 0; p < h->npass; ++p)
    for (int i = 0; i < m.njoints; ++i)
      if ((h->pass_bodies[p] >> i) & 1ull) h->pass_visit[p][h->pass_nvisit[p]++] = i;
}

// ---------------------------------------------------------------------------------------------------------------------
// Values of the recursion, ahead of the tangent walk (PRE mode of linearize_contact_dynamics_kernel).
//
// The walk below is one dependent instruction stream per wave, and the lane-invariant VALUES of the recursion (joint
// transforms incl. a sincos per joint, placements, velocities, accelerations, forces) were ~40 % of it, recomputed by every
// lane, body after body.  They are level-parallel instead: here lanes = bodies, one pass down the levels of the tree
// (placements, v, a, own force) and one pass up the bodies in reverse depth-first order (forces of the children), several
// grid points per wave (64 / next power of two >= njoints).  Out: 64 doubles per body --
//   0 R, 9 p, 12 oR, 21 op, 24 v, 30 a, 36 g, 39 f (TOTAL: own + children - contact forces), 45 body index,
//   46 v_parent in body coordinates, 52 a_parent likewise, 58 h = I v
// the block the walk copies into its per-level value slots when it visits the body -- and the inverse-dynamics residual ID
// (the value part of RTOC_CDD_IDC).  trav: 0 = the dynamics traversal, 1 = the kinematics traversal at v + dv of impact
// grids (impact_stage.cpp:61), other grids idle.
struct ValArgs {
  const DevModel* model;
  const double* sol;
  double* cdd;
  double* vals;         // [batch * nstages][njoints][64]
  const rtoc_grid* grid;
  const unsigned* active;
  int nstages, batch, nv, nu, njoints, ncontacts, nlevels, gs, trav, unconstr;
  int nsel, sel[16];    // nsel > 0: only the grid points sel[0..nsel) of every instance (the impact grids of trav 1)
  int sol_stride, cdd_stride;
  int o_q, o_v, o_a, o_u, o_f, o_idc;
  double gx, gy, gz;
  const double* positions;   // [nstages][ncontacts][3] or nullptr: desired contact positions (the value rows of C below)
  const double* rotations;   // [nstages][ncontacts][9] or nullptr (surface contacts)
};
constexpr int VAL_SLOTS = 64;

static __global__ __launch_bounds__(64) void rbd_values_kernel(ValArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];   // [G][njoints][64]
  const int lane = threadIdx.x, GS = a.gs, G = 64 / GS, nb = a.njoints, ncon = a.ncontacts;
  const int grp = lane / GS, i = lane % GS;
  const int nst1 = a.nsel > 0 ? a.nsel : a.nstages - 1;   // grid points per instance this launch covers
  const long long item = (long long)blockIdx.x * G + grp, nitems = (long long)a.batch * nst1;
  const bool gvalid = item < nitems;
  const int b = gvalid ? (int)(item / nst1) : 0;
  const int st = !gvalid ? 0 : (a.nsel > 0 ? a.sel[item % nst1] : (int)(item % nst1));
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const bool dyn = a.trav == 0;
  const bool on = gvalid && i < nb && (dyn || impact);
  const int ib = i < nb ? i : 0;
  double* const V = smem + (size_t)grp * nb * VAL_SLOTS;
  double* const me = V + (size_t)ib * VAL_SLOTS;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const sr = a.sol + rec * a.sol_stride;
  const unsigned active = a.active[st];
  const int nv = a.nv, nu = a.nu;
  // ---- this body's constants and joint state ----
  const double* const jm = &a.model->joint[ib][0];
  const int type = (int)jm[28], iq = (int)jm[29], iv = (int)jm[30], depth = (int)jm[31];
  const int par = a.model->m.parent[ib];
  const bool ff = type == RTOC_JOINT_FREE_FLYER;
  const V3 ax = ldv3(jm + 12);
  M3 Rj;
  V3 pj = mk(0, 0, 0);
  SV vj, aj;
  if (ff) {
    const double x = sr[a.o_q + iq + 3], y = sr[a.o_q + iq + 4], z = sr[a.o_q + iq + 5], w = sr[a.o_q + iq + 6];
    Rj.m[0] = 1 - 2 * (y * y + z * z), Rj.m[1] = 2 * (x * y - z * w), Rj.m[2] = 2 * (x * z + y * w);
    Rj.m[3] = 2 * (x * y + z * w), Rj.m[4] = 1 - 2 * (x * x + z * z), Rj.m[5] = 2 * (y * z - x * w);
    Rj.m[6] = 2 * (x * z - y * w), Rj.m[7] = 2 * (y * z + x * w), Rj.m[8] = 1 - 2 * (x * x + y * y);
    pj = ldv3(sr + a.o_q + iq);
    vj = SV{ldv3(sr + a.o_v + iv), ldv3(sr + a.o_v + iv + 3)};
    aj = SV{ldv3(sr + a.o_a + iv), ldv3(sr + a.o_a + iv + 3)};
    if (impact && !dyn) vj = vj + aj;  // kinematics at v + dv
  } else {
    const double th = sr[a.o_q + iq], c = cos(th), s = sin(th), t = 1.0 - c;
    Rj.m[0] = t * ax.x * ax.x + c, Rj.m[1] = t * ax.x * ax.y - s * ax.z, Rj.m[2] = t * ax.x * ax.z + s * ax.y;
    Rj.m[3] = t * ax.x * ax.y + s * ax.z, Rj.m[4] = t * ax.y * ax.y + c, Rj.m[5] = t * ax.y * ax.z - s * ax.x;
    Rj.m[6] = t * ax.x * ax.z - s * ax.y, Rj.m[7] = t * ax.y * ax.z + s * ax.x, Rj.m[8] = t * ax.z * ax.z + c;
    const double vq = (impact && !dyn) ? sr[a.o_v + iv] + sr[a.o_a + iv] : sr[a.o_v + iv];
    vj = SV{mk(0, 0, 0), vq * ax};
    aj = SV{mk(0, 0, 0), sr[a.o_a + iv] * ax};
  }
  if (impact && dyn) vj = sv0();   // impact model: v = 0
  if (impact && !dyn) aj = sv0();  // velocity-level rows only
  const M3 Rp = ldm3(jm);
  const M3 R = mul(Rp, Rj);
  const V3 p = mul(Rp, pj) + ldv3(jm + 9);
  const double mass = jm[15];
  const V3 com = ldv3(jm + 16);
  const M3 I = ldm3(jm + 19);
  auto st_v3 = [&](double* d, V3 x) { d[0] = x.x, d[1] = x.y, d[2] = x.z; };
  auto st_sv6 = [&](double* d, SV x) { st_v3(d, x.l), st_v3(d + 3, x.a); };
  auto ld_sv6 = [&](const double* d) { return SV{ldv3(d), ldv3(d + 3)}; };
  // ---- down the levels: placements, velocities, accelerations, own forces ----
  for (int d = 0; d < a.nlevels; ++d) {
    if (on && depth == d) {
      M3 oR = R;
      V3 op = p;
      SV vpar = sv0(), apar = sv0();
      V3 gi = mulT(R, mk(-a.gx, -a.gy, -a.gz));
      if (d > 0) {
        const double* const pa = V + (size_t)par * VAL_SLOTS;
        const M3 oRp = ldm3(pa + 12);
        oR = mul(oRp, R);
        op = ldv3(pa + 21) + mul(oRp, p);
        vpar = act_inv(R, p, ld_sv6(pa + 24));
        apar = act_inv(R, p, ld_sv6(pa + 30));
        gi = mulT(R, ldv3(pa + 36));
      }
      if (impact) gi = mk(0, 0, 0);  // the impact model has no gravity
      const SV v = vpar + vj;
      const SV acc = apar + aj + mcross(v, vj);
      const SV h = inertia_mul(mass, com, I, v);
      SV f = inertia_mul(mass, com, I, SV{acc.l + gi, acc.a}) + fcross(v, h);
      int roff = 0;
      for (int c = 0; c < ncon; ++c) {
        const double* const cm = &a.model->contact[c][0];
        const bool con_on = (active >> c) & 1u;
        const bool surf = (int)cm[15] == RTOC_CONTACT_SURFACE;
        if (con_on && (int)cm[14] == ib) {
          const SV fc = SV{ldv3(sr + a.o_f + roff), surf ? ldv3(sr + a.o_f + roff + 3) : mk(0, 0, 0)};
          f = f - act_f(ldm3(cm), ldv3(cm + 9), fc);
        }
        roff += con_on ? (surf ? 6 : 3) : 0;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) me[k] = R.m[k], me[12 + k] = oR.m[k];
      st_v3(me + 9, p), st_v3(me + 21, op);
      st_sv6(me + 24, v), st_sv6(me + 30, acc);
      st_v3(me + 36, gi);
      st_sv6(me + 39, f);
      me[45] = (double)ib;
      st_sv6(me + 46, vpar), st_sv6(me + 52, apar), st_sv6(me + 58, h);
    }
    __syncthreads();
  }
  // ---- up the bodies, children before parents (reverse depth-first order): total forces ----
  for (int k = nb - 1; k >= 1; --k) {
    const int pk = a.model->m.parent[k];
    if (on && ib == pk) {
      const double* const ch = V + (size_t)k * VAL_SLOTS;
      st_sv6(me + 39, ld_sv6(me + 39) + act_f(ldm3(ch), ldv3(ch + 9), ld_sv6(ch + 39)));
    }
    __syncthreads();
  }
  // ---- ID = S^T f - [0; u] (the value rows of RTOC_CDD_IDC; contact_dynamics.cpp:21-24, impact_dynamics.cpp:12-14) ----
  if (on && dyn) {
    double* const cr = a.cdd + rec * a.cdd_stride;
    const SV f = ld_sv6(me + 39);
    if (ff) {
      const double fv[6] = {f.l.x, f.l.y, f.l.z, f.a.x, f.a.y, f.a.z};
#pragma unroll
      for (int k = 0; k < 6; ++k) cr[a.o_idc + iv + k] = fv[k] - ((!impact && iv + k >= nv - nu) ? sr[a.o_u + iv + k - (nv - nu)] : 0.0);
    } else {
      cr[a.o_idc + iv] = dot(ax, f.a) - ((!impact && iv >= nv - nu) ? sr[a.o_u + iv - (nv - nu)] : 0.0);
    }
  }
  // ---- C = the Baumgarte residual of the contacts this body carries (point_contact.hxx:14-31, surface_contact.hxx:12-29), on
  //      impact grids the contact velocity at v + dv from the kinematics traversal (point_contact.hxx:84-96): the value rows
  //      nv.. of RTOC_CDD_IDC.  Lane-invariant in the tangent walk, which used to evaluate them in every lane. ----
  if (on && (impact ? !dyn : dyn)) {
    double* const cr = a.cdd + rec * a.cdd_stride;
    const SV v = ld_sv6(me + 24), acc = ld_sv6(me + 30);
    const M3 oR = ldm3(me + 12);
    const V3 op = ldv3(me + 21);
    int roff = 0;
    for (int c = 0; c < ncon; ++c) {
      const double* const cm = &a.model->contact[c][0];
      const bool con_on = (active >> c) & 1u;
      const bool surf = (int)cm[15] == RTOC_CONTACT_SURFACE;
      const int nr = surf ? 6 : 3;
      if (con_on && (int)cm[14] == ib) {
        const M3 Rf = ldm3(cm);
        const V3 pf = ldv3(cm + 9);
        const SV vf = act_inv(Rf, pf, v);
        SV C = vf;
        if (!impact) {
          const SV af = act_inv(Rf, pf, acc);
          const double kp = cm[12], kd = cm[13];
          const V3 pw = op + mul(oR, pf);
          const V3 pr = a.positions ? ldv3(a.positions + ((size_t)st * ncon + c) * 3) : mk(0, 0, 0);
          if (!surf) {
            C.l = af.l + cross(vf.a, vf.l) + kd * vf.l + kp * (pw - pr);
          } else {
            M3 Rdt;   // transpose of the desired rotation
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) Rdt.m[3 * r + cc] = a.rotations ? a.rotations[((size_t)st * ncon + c) * 9 + 3 * cc + r] : (r == cc ? 1.0 : 0.0);
            SV lg, dlg;
            log6_fwd(mul(Rdt, mul(oR, Rf)), mul(Rdt, pw - pr), sv0(), lg, dlg);
            C = SV{af.l + kd * vf.l + kp * lg.l, af.a + kd * vf.a + kp * lg.a};
          }
        }
        const double Cv[6] = {C.l.x, C.l.y, C.l.z, C.a.x, C.a.y, C.a.z};
        for (int t = 0; t < nr; ++t) cr[a.o_idc + nv + roff + t] = Cv[t];
      }
      roff += con_on ? nr : 0;
    }
  }
  // ---- the blocks out, coalesced ----
  const int per = nb * VAL_SLOTS;
  for (int e = lane; e < G * per; e += 64) {
    const int g2 = e / per;
    const long long it2 = (long long)blockIdx.x * G + g2;
    if (it2 >= nitems) continue;
    const int b2 = (int)(it2 / nst1), st2 = a.nsel > 0 ? a.sel[it2 % nst1] : (int)(it2 % nst1);
    if (!dyn && a.grid[st2].type != RTOC_GRID_IMPACT) continue;
    a.vals[((size_t)b2 * a.nstages + st2) * per + (e - g2 * per)] = smem[e];
  }
}

// SURF: the model has surface contacts (6 rows, Log6 of the placement error); compiled out for point-contact robots, where
// its registers and branches cost 10 % of the kernel
// PRE: the values of the recursion come from rbd_values_kernel (a.vals / a.vals2): the visit copies the body's block into
// the level's value slots instead of computing it, and the force accumulation / ID rows are not repeated here.
template <bool SURF, bool PRE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((PRE && !SURF) ? 2 : 1))) void linearize_contact_dynamics_kernel(LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const int nlev = a.nlevels, nv = a.nv, nb = a.njoints, ncon = a.ncontacts;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const unsigned active = a.active[st];
  double* const lval = smem;                                      // [nlev][VAL_DOUBLES]
  const int LW = lin_lane_stride(a.dpp);
  double* const lfwd = lval + (size_t)nlev * VAL_DOUBLES;         // [nbranch][FWD_SLOTS][LW]: dv, da, dg of the branching bodies
  double* const ldf = lfwd + (size_t)a.nbranch * FWD_SLOTS * LW;  // [nlev - 1][DF_SLOTS][LW]: df of the open non-leaf levels
  double* const sq = ldf + (size_t)(nlev > 1 ? nlev - 1 : 0) * DF_SLOTS * LW;   // q, v, a, f, u of the grid point
  double* const sv = sq + (PRE ? 0 : lin_pad8(nv + 1));   // (PRE: the staging vectors of the values are not allocated)
  double* const sa = sv + (PRE ? 0 : lin_pad8(nv));
  double* const sf = sa + (PRE ? 0 : lin_pad8(nv));
  double* const su = sf + (PRE ? 0 : lin_pad8(6 * ncon));
  double* const sbeta = su + (PRE ? 0 : lin_pad8(nv));  // multipliers of the dynamics (beta) and of the contact rows (mu)
  double* const smu = sbeta + lin_pad8(nv);
  double* const slf = smu + lin_pad8(6 * ncon);         // dC/da beta, accumulated over the passes
  constexpr int JPW = PRE ? JP_PRE : JP, JOFF = PRE ? JP_PRE_OFF : 0;
  double* const sjm = slf + lin_pad8(6 * ncon);         // model: [njoints][JPW] (PRE: axis .. depth only), then [ncontacts][CP]
  double* const scm = sjm + a.njoints * JPW;
  const bool aug = a.kkt != nullptr;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const sr = a.sol + rec * a.sol_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  const int nu = a.nu;
  {
    const double* const gj = &a.model->joint[0][0];
    const double* const gc = &a.model->contact[0][0];
    for (int e = lane; e < nb * JPW; e += 64) sjm[e] = gj[(e / JPW) * JP + JOFF + e % JPW];
    for (int e = lane; e < ncon * CP; e += 64) scm[e] = gc[e];
  }
  if constexpr (!PRE) {
    for (int e = lane; e < a.nq; e += 64) sq[e] = sr[a.o_q + e];
    for (int e = lane; e < nv; e += 64) {
      sv[e] = sr[a.o_v + e];
      sa[e] = sr[a.o_a + e];
    }
    for (int e = lane; e < g.dimf; e += 64) sf[e] = sr[a.o_f + e];
    for (int e = lane; e < nu; e += 64) su[e] = sr[a.o_u + e];
  }
  if (aug) {
    for (int e = lane; e < nv; e += 64) sbeta[e] = sr[a.o_beta + e];
    for (int e = lane; e < g.dimf; e += 64) smu[e] = sr[a.o_mu + e];
    for (int e = lane; e < 6 * ncon; e += 64) slf[e] = 0.0;
  }
  __syncthreads();
  const V3 grav = mk(a.gx, a.gy, a.gz);
  // impact grids: a dynamics traversal (zero gravity, zero velocity, acceleration = dv; robot.hxx:590-624) and a
  // kinematics traversal at v + dv for the contact-velocity rows (impact_stage.cpp:61); other grids: one traversal
  const int ntrav = impact ? 2 : 1;
  for (int trav = 0; trav < ntrav; ++trav) {
    const bool dyn = trav == 0;                 // writes ID and its derivatives
    const bool rows = !impact || trav == 1;     // writes C and its derivatives
    for (int j0 = 0, ps = 0; j0 < nv; j0 += a.dpp, ++ps) {
      const int j = j0 + lane / 3, kind = lane % 3;  // 0: q, 1: v, 2: a
      const bool lane_on = lane < 3 * a.dpp && j < nv;
      // (without the values pre-pass every pass accumulates the forces of all bodies: no skipping)
      const unsigned long long visit_mask = PRE ? a.model->pass_bodies[ps] : ~0ull;
      int top = -1;
      double wsum = 0.0;  // this lane's column of [dID; dC] against [beta; mu]
      // body of the level that is being closed / visited is kept in LDS as an int in the value block
      auto LV = [&](int lev, int k) -> double& { return lval[lev * VAL_DOUBLES + k]; };
      // forward-tangent slot / force-tangent level of this lane; lanes beyond the stride (idle ones) share the last column
      const int ln = lane < LW ? lane : LW - 1;
      auto FT = [&](int slot, int k) -> double& { return lfwd[((size_t)slot * FWD_SLOTS + k) * LW + ln]; };
      auto DT = [&](int lev, int k) -> double& { return ldf[((size_t)lev * DF_SLOTS + k) * LW + ln]; };
      auto ld_sv = [&](int lev, int k0) { return SV{mk(LV(lev, k0), LV(lev, k0 + 1), LV(lev, k0 + 2)), mk(LV(lev, k0 + 3), LV(lev, k0 + 4), LV(lev, k0 + 5))}; };
      auto st_sv = [&](int lev, int k0, SV x) {
        LV(lev, k0) = x.l.x, LV(lev, k0 + 1) = x.l.y, LV(lev, k0 + 2) = x.l.z, LV(lev, k0 + 3) = x.a.x, LV(lev, k0 + 4) = x.a.y, LV(lev, k0 + 5) = x.a.z;
      };
      auto ld_ft = [&](int slot, int k0) { return SV{mk(FT(slot, k0), FT(slot, k0 + 1), FT(slot, k0 + 2)), mk(FT(slot, k0 + 3), FT(slot, k0 + 4), FT(slot, k0 + 5))}; };
      auto st_ft = [&](int slot, int k0, SV x) {
        FT(slot, k0) = x.l.x, FT(slot, k0 + 1) = x.l.y, FT(slot, k0 + 2) = x.l.z, FT(slot, k0 + 3) = x.a.x, FT(slot, k0 + 4) = x.a.y, FT(slot, k0 + 5) = x.a.z;
      };
      auto ld_dt = [&](int lev) { return SV{mk(DT(lev, 0), DT(lev, 1), DT(lev, 2)), mk(DT(lev, 3), DT(lev, 4), DT(lev, 5))}; };
      auto st_dt = [&](int lev, SV x) { DT(lev, 0) = x.l.x, DT(lev, 1) = x.l.y, DT(lev, 2) = x.l.z, DT(lev, 3) = x.a.x, DT(lev, 4) = x.a.y, DT(lev, 5) = x.a.z; };
      // value slots: 0 R, 9 p, 12 oR, 21 op, 24 v, 30 a, 36 g, 39 f, 45 body index
      // forward-tangent slots: 0 dv, 6 da, 12 dg
      SV tdv = sv0(), tda = sv0(), cdf = sv0();   // forward tangents of the body visited last; force tangent of an open leaf
      V3 tdg = mk(0, 0, 0);
      bool topreg = false;                        // the force tangent of level `top` is cdf (a leaf), not in LDS
      auto JM = [&](int i, int k) -> const double& { return sjm[i * JPW + k - JOFF]; };
      auto unit_twist = [&](int i, int k) -> SV {  // S_k of joint i
        if ((int)JM(i, 28) == RTOC_JOINT_FREE_FLYER)
          return SV{mk(k == 0, k == 1, k == 2), mk(k == 3, k == 4, k == 5)};
        return SV{mk(0, 0, 0), ldv3(&JM(i, 12))};
      };
      auto close = [&](int lev, bool from_reg) {
        const int i = (int)LV(lev, 45);
        const M3 R = ldm3(&LV(lev, 0));
        const V3 p = ldv3(&LV(lev, 9));
        const SV f = ld_sv(lev, 39), df = from_reg ? cdf : ld_dt(lev);
        const int iv = (int)JM(i, 30);
        const bool cff = (int)JM(i, 28) == RTOC_JOINT_FREE_FLYER;
        const bool own = lane_on && j >= iv && j < iv + (cff ? 6 : 1);
        if (dyn) {
          // tau = S^T f: value (lane 0) and this lane's column
          double* const dcol = kind == 2 ? cr + a.o_didda + (size_t)j * nv : cr + a.o_didcdqv + (size_t)(kind == 1 ? nv + j : j) * a.ldv;
          if (cff) {
            const double fv[6] = {f.l.x, f.l.y, f.l.z, f.a.x, f.a.y, f.a.z}, dv6[6] = {df.l.x, df.l.y, df.l.z, df.a.x, df.a.y, df.a.z};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              if (!PRE && lane == 0 && j0 == 0) cr[a.o_idc + iv + k] = fv[k] - ((!impact && iv + k >= nv - nu) ? su[iv + k - (nv - nu)] : 0.0);
              if (lane_on) dcol[iv + k] = dv6[k];
              if (aug) wsum += dv6[k] * sbeta[iv + k];
            }
          } else {
            const V3 ax = ldv3(&JM(i, 12));
            if (!PRE && lane == 0 && j0 == 0) cr[a.o_idc + iv] = dot(ax, f.a) - ((!impact && iv >= nv - nu) ? su[iv - (nv - nu)] : 0.0);
            if (lane_on) dcol[iv] = dot(ax, df.a);
            if (aug) wsum += dot(ax, df.a) * sbeta[iv];
          }
        }
        if (lev > 0) {
          SV dfp = act_f(R, p, df);
          if (own && kind == 0) {   // a body with a parent is a revolute joint: S = (0, axis)
            const V3 ax = ldv3(&JM(i, 12));
            dfp = dfp + act_f(R, p, SV{cross(ax, f.l), cross(ax, f.a)});
          }
          if (!PRE) st_sv(lev - 1, 39, ld_sv(lev - 1, 39) + act_f(R, p, f));   // PRE: the block already holds the total force
          st_dt(lev - 1, ld_dt(lev - 1) + dfp);
        }
      };
      // bodies no direction of this pass moves or loads: their entries of the pass's columns are zero
      if (PRE && ps > 0)
        for (int i = 0; i < nb; ++i) {
          if ((visit_mask >> i) & 1ull) continue;
          const int iv = (int)JM(i, 30);
          if (dyn && lane_on) {
            double* const dcol = kind == 2 ? cr + a.o_didda + (size_t)j * nv : cr + a.o_didcdqv + (size_t)(kind == 1 ? nv + j : j) * a.ldv;
            const int ndof = (int)JM(i, 28) == RTOC_JOINT_FREE_FLYER ? 6 : 1;
            for (int k = 0; k < ndof; ++k) dcol[iv + k] = 0.0;
          }
          if (rows) {
            int roff = 0;
            for (int c = 0; c < ncon; ++c) {
              const bool on = (active >> c) & 1u;
              const int nr = (SURF && (int)scm[c * CP + 15] == RTOC_CONTACT_SURFACE) ? 6 : 3;
              if (on && (int)scm[c * CP + 14] == i && lane_on)
                for (int t = 0; t < nr; ++t) {
                  if (kind == 2 || (impact && kind == 1)) cr[a.o_dcda + (size_t)j * a.nf_max + roff + t] = 0.0;
                  if (kind != 2) cr[a.o_didcdqv + (size_t)(kind == 1 ? nv + j : j) * a.ldv + nv + roff + t] = 0.0;
                }
              roff += on ? nr : 0;
            }
          }
        }
      const double* const vblk = PRE ? (dyn ? a.vals : a.vals2) + rec * (size_t)nb * VAL_SLOTS : nullptr;
      double pv = PRE ? vblk[lane] : 0.0;   // body 0 is in every pass
      const int* const vlist = a.model->pass_visit[ps];
      const int nvisit = PRE ? sload_int(a.model->pass_nvisit + ps) : nb;
      for (int t = 0; t < nvisit; ++t) {
        const int i = PRE ? sload_int(vlist + t) : t;   // the bodies a direction of this pass moves or loads, depth first
        const int d = (int)JM(i, 31);
        while (top >= d) {
          close(top, topreg);
          topreg = false;
          --top;
        }
        // ---- visit body i at level d ----
        const int wk = sload_int(a.model->walk + i);
        const bool chain = wk & 1, leaf = wk & 2;
        const int sslot = ((wk >> 4) & 15) - 1, pslot = ((wk >> 8) & 15) - 1;
        const int iq = (int)JM(i, 29), iv = (int)JM(i, 30);
        const bool ff = (int)JM(i, 28) == RTOC_JOINT_FREE_FLYER;
        const bool own = lane_on && j >= iv && j < iv + (ff ? 6 : 1);
        M3 R, oR;
        V3 p, op, gi;
        SV vpar, apar, v, acc, vj;
        if constexpr (PRE) {
          // the body's block from rbd_values_kernel (requested one body ahead) into the level's value slots
          LV(d, lane) = pv;
          if (t + 1 < nvisit) pv = vblk[(size_t)sload_int(vlist + t + 1) * VAL_SLOTS + lane];
          __builtin_amdgcn_wave_barrier();
          R = ldm3(&LV(d, 0)), oR = ldm3(&LV(d, 12));
          p = ldv3(&LV(d, 9)), op = ldv3(&LV(d, 21)), gi = ldv3(&LV(d, 36));
          v = ld_sv(d, 24), acc = ld_sv(d, 30), vpar = ld_sv(d, 46), apar = ld_sv(d, 52);
          vj = v - vpar;
        } else {
        M3 Rj;
        V3 pj = mk(0, 0, 0);
        SV aj;  // S aq (vj = S vq)
        if (ff) {
          const double x = sq[iq + 3], y = sq[iq + 4], z = sq[iq + 5], w = sq[iq + 6];
          Rj.m[0] = 1 - 2 * (y * y + z * z), Rj.m[1] = 2 * (x * y - z * w), Rj.m[2] = 2 * (x * z + y * w);
          Rj.m[3] = 2 * (x * y + z * w), Rj.m[4] = 1 - 2 * (x * x + z * z), Rj.m[5] = 2 * (y * z - x * w);
          Rj.m[6] = 2 * (x * z - y * w), Rj.m[7] = 2 * (y * z + x * w), Rj.m[8] = 1 - 2 * (x * x + y * y);
          pj = mk(sq[iq], sq[iq + 1], sq[iq + 2]);
          vj = SV{mk(sv[iv], sv[iv + 1], sv[iv + 2]), mk(sv[iv + 3], sv[iv + 4], sv[iv + 5])};
          aj = SV{mk(sa[iv], sa[iv + 1], sa[iv + 2]), mk(sa[iv + 3], sa[iv + 4], sa[iv + 5])};
          if (impact && !dyn) vj = vj + aj;  // kinematics at v + dv
        } else {
          const V3 ax = ldv3(&JM(i, 12));
          const double th = sq[iq], c = cos(th), s = sin(th), t = 1.0 - c;
          Rj.m[0] = t * ax.x * ax.x + c, Rj.m[1] = t * ax.x * ax.y - s * ax.z, Rj.m[2] = t * ax.x * ax.z + s * ax.y;
          Rj.m[3] = t * ax.x * ax.y + s * ax.z, Rj.m[4] = t * ax.y * ax.y + c, Rj.m[5] = t * ax.y * ax.z - s * ax.x;
          Rj.m[6] = t * ax.x * ax.z - s * ax.y, Rj.m[7] = t * ax.y * ax.z + s * ax.x, Rj.m[8] = t * ax.z * ax.z + c;
          const double vq = (impact && !dyn) ? sv[iv] + sa[iv] : sv[iv];
          vj = SV{mk(0, 0, 0), vq * ax};
          aj = SV{mk(0, 0, 0), sa[iv] * ax};
        }
        if (impact && dyn) vj = sv0();   // impact model: v = 0
        if (impact && !dyn) aj = sv0();  // velocity-level rows only
        const M3 Rp = ldm3(&JM(i, 0));
        R = mul(Rp, Rj);
        p = mul(Rp, pj) + ldv3(&JM(i, 9));
        oR = R;
        op = p;
        vpar = sv0(), apar = sv0();
        gi = mulT(R, mk(-grav.x, -grav.y, -grav.z));
        if (d > 0) {
          const M3 oRp = ldm3(&LV(d - 1, 12));
          oR = mul(oRp, R);
          op = ldv3(&LV(d - 1, 21)) + mul(oRp, p);
          vpar = act_inv(R, p, ld_sv(d - 1, 24));
          apar = act_inv(R, p, ld_sv(d - 1, 30));
          gi = mulT(R, ldv3(&LV(d - 1, 36)));
        }
        if (impact) gi = mk(0, 0, 0);  // the impact model has no gravity
        v = vpar + vj;
        acc = apar + aj + mcross(v, vj);
        }
        SV dvp = sv0(), dap = sv0();
        V3 dgp = mk(0, 0, 0);
        if (d > 0) {
          if (!chain) {   // the parent is not the body visited just before: its tangents are in its slot
            tdv = ld_ft(pslot, 0), tda = ld_ft(pslot, 6);
            tdg = mk(FT(pslot, 12), FT(pslot, 13), FT(pslot, 14));
          }
          dvp = act_inv(R, p, tdv);
          dap = act_inv(R, p, tda);
          dgp = mulT(R, tdg);
        }
        if (impact) dgp = mk(0, 0, 0);
        SV dv = dvp, da = dap;
        V3 dg = dgp;
        const bool vdir = (kind == 1 || (impact && !dyn && kind == 2)) && !(impact && dyn);   // the lane's direction moves v
        if (ff) {
          if (own) {
            const SV S = unit_twist(i, j - iv);
            if (kind == 0) {
              dv = dv - mcross(S, vpar);
              da = da - mcross(S, apar);
              dg = dg - cross(S.a, gi);
            } else if (kind == 1) {
              if (!(impact && dyn)) dv = dv + S;
            } else {
              if (!(impact && !dyn)) da = da + S;
              if (impact && !dyn) dv = dv + S;  // d(v + dv)/d(dv)
            }
          }
          da = da + mcross(dv, vj);
          if (own && vdir) da = da + mcross(v, unit_twist(i, j - iv));
        } else {
          // revolute joint: S = (0, axis), so S x m = (axis x m.l, axis x m.a) and m x S = (m.l x axis, m.a x axis)
          const V3 ax = ldv3(&JM(i, 12));
          if (own) {
            if (kind == 0) {
              dv = dv - SV{cross(ax, vpar.l), cross(ax, vpar.a)};
              da = da - SV{cross(ax, apar.l), cross(ax, apar.a)};
              dg = dg - cross(ax, gi);
            } else if (kind == 1) {
              if (!(impact && dyn)) dv.a = dv.a + ax;
            } else {
              if (!(impact && !dyn)) da.a = da.a + ax;
              if (impact && !dyn) dv.a = dv.a + ax;  // d(v + dv)/d(dv)
            }
          }
          da = da + SV{cross(dv.l, vj.a), cross(dv.a, vj.a)};   // dv x vj with vj = (0, vq axis)
          if (own && vdir) da = da + SV{cross(v.l, ax), cross(v.a, ax)};
        }
        // own force and its tangent
        const double mass = JM(i, 15);
        const V3 com = ldv3(&JM(i, 16));
        const M3 I = ldm3(&JM(i, 19));
        const SV h = PRE ? ld_sv(d, 58) : inertia_mul(mass, com, I, v);
        SV f = PRE ? sv0() : inertia_mul(mass, com, I, SV{acc.l + gi, acc.a}) + fcross(v, h);   // PRE: the block holds the total force
        const SV df = inertia_mul(mass, com, I, SV{da.l + dg, da.a}) + fcross(dv, h) + fcross(v, inertia_mul(mass, com, I, dv));
        // contacts carried by this body; roff = rows of the active contacts before c (3 per point, 6 per surface contact)
        int roff = 0;
        for (int c = 0; c < ncon; ++c) {
          const bool on = (active >> c) & 1u;
          const bool surf = SURF && (int)scm[c * CP + 15] == RTOC_CONTACT_SURFACE;
          const int nr = surf ? 6 : 3;
          if (on && (int)scm[c * CP + 14] == i) {
            const M3 Rf = ldm3(&scm[c * CP]);
            const V3 pf = ldv3(&scm[c * CP + 9]);
            // the contact force / wrench is given in the LOCAL contact frame (point_contact.cpp:55-60, surface_contact.cpp)
            const SV fc = PRE ? sv0() : SV{mk(sf[roff], sf[roff + 1], sf[roff + 2]), surf ? mk(sf[roff + 3], sf[roff + 4], sf[roff + 5]) : mk(0, 0, 0)};
            if (!PRE) f = f - act_f(Rf, pf, fc);
            if (rows) {
              const SV vf = act_inv(Rf, pf, v), dvf = act_inv(Rf, pf, dv);
              SV C, dC;  // angular parts only used by surface contacts
              if (impact) {
                C = vf;
                dC = dvf;
              } else {
                const SV af = act_inv(Rf, pf, acc), daf = act_inv(Rf, pf, da);
                const double kp = scm[c * CP + 12], kd = scm[c * CP + 13];
                const M3 oRf = mul(oR, Rf);
                const V3 pw = op + mul(oR, pf);
                const V3 pr = a.positions ? ldv3(a.positions + ((size_t)st * ncon + c) * 3) : mk(0, 0, 0);
                // the frame Jacobian column of dof j is the v-tangent of vf, one lane up (kind 0 lanes only use it)
                const SV jc = SV{mk(__shfl_down(dvf.l.x, 1, 64), __shfl_down(dvf.l.y, 1, 64), __shfl_down(dvf.l.z, 1, 64)),
                                 mk(__shfl_down(dvf.a.x, 1, 64), __shfl_down(dvf.a.y, 1, 64), __shfl_down(dvf.a.z, 1, 64))};
                if (!surf) {
                  // classical linear acceleration + kd v + kp (p - p_desired)   (point_contact.hxx:14-31, :33-83)
                  C.l = af.l + cross(vf.a, vf.l) + kd * vf.l + kp * (pw - pr);
                  dC.l = daf.l + cross(dvf.a, vf.l) + cross(vf.a, dvf.l) + kd * dvf.l;
                  if (kind == 0) dC.l = dC.l + kp * mul(oRf, jc.l);
                  C.a = mk(0, 0, 0), dC.a = mk(0, 0, 0);
                } else {
                  // spatial acceleration + kd v + kp Log6(X_desired^-1 X_frame)   (surface_contact.hxx:12-29, :31-68)
                  M3 Rd;
                  if (a.rotations) {
                    Rd = ldm3(a.rotations + ((size_t)st * ncon + c) * 9);
                  } else {
#pragma unroll
                    for (int e = 0; e < 9; ++e) Rd.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
                  }
                  M3 Rdt;
#pragma unroll
                  for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) Rdt.m[3 * r + cc] = Rd.m[3 * cc + r];
                  SV lg, dlg;
                  log6_fwd(mul(Rdt, oRf), mul(Rdt, pw - pr), jc, lg, dlg);
                  C = SV{af.l + kd * vf.l + kp * lg.l, af.a + kd * vf.a + kp * lg.a};
                  dC = SV{daf.l + kd * dvf.l, daf.a + kd * dvf.a};
                  if (kind == 0) dC = SV{dC.l + kp * dlg.l, dC.a + kp * dlg.a};
                }
              }
              const double Cv[6] = {C.l.x, C.l.y, C.l.z, C.a.x, C.a.y, C.a.z}, dCv[6] = {dC.l.x, dC.l.y, dC.l.z, dC.a.x, dC.a.y, dC.a.z};
              const int r0 = nv + roff;
              if (aug) {
                // lf -= dC/da beta (contact_dynamics.cpp:38; impact: dC/dv, impact_dynamics.cpp:21): sum over the a-lanes
                const bool al = lane_on && kind == 2;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                  if (t < nr) {
                    wsum += dCv[t] * smu[roff + t];
                    double rx = al ? dCv[t] * sbeta[j] : 0.0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) rx += __shfl_xor(rx, off, 64);
                    if (lane == 0) slf[roff + t] += rx;
                  }
                }
              }
#pragma unroll
              for (int t = 0; t < 6; ++t) {
                if (t < nr) {
                  if (!PRE && lane == 0 && j0 == 0) cr[a.o_idc + r0 + t] = Cv[t];   // PRE: rbd_values_kernel wrote the values
                  if (lane_on) {
                    // impact: dC/dv = dC/d(dv) goes where the condensation reads it (the v block of DIDCDQV) and into DCDA
                    if (kind == 2 || (impact && kind == 1)) cr[a.o_dcda + (size_t)j * a.nf_max + (r0 - nv) + t] = dCv[t];
                    if (kind != 2) cr[a.o_didcdqv + (size_t)(kind == 1 ? nv + j : j) * a.ldv + r0 + t] = dCv[t];
                  }
                }
              }
            }
          }
          roff += on ? nr : 0;
        }
        // ---- store the level ----
        if constexpr (!PRE) {
#pragma unroll
          for (int k = 0; k < 9; ++k) LV(d, k) = R.m[k], LV(d, 12 + k) = oR.m[k];
          LV(d, 9) = p.x, LV(d, 10) = p.y, LV(d, 11) = p.z;
          LV(d, 21) = op.x, LV(d, 22) = op.y, LV(d, 23) = op.z;
          st_sv(d, 24, v);
          st_sv(d, 30, acc);
          LV(d, 36) = gi.x, LV(d, 37) = gi.y, LV(d, 38) = gi.z;
          st_sv(d, 39, f);
          LV(d, 45) = (double)i;
        }
        tdv = dv, tda = da, tdg = dg;
        if (sslot >= 0) {
          st_ft(sslot, 0, dv);
          st_ft(sslot, 6, da);
          FT(sslot, 12) = dg.x, FT(sslot, 13) = dg.y, FT(sslot, 14) = dg.z;
        }
        if (leaf) cdf = df;
        else st_dt(d, df);
        top = d;
        topreg = leaf;
      }
      while (top >= 0) {
        close(top, topreg);
        topreg = false;
        --top;
      }
      if (aug && lane_on) {
        // lq / lv / la (ldv) += (this lane's column)^T [beta; mu]  (contact_dynamics.cpp:35-37,49-51; impact_dynamics.cpp:19-25)
        double* const kr = a.kkt + rec * a.kkt_stride;
        double* const dst = kind == 0 ? kr + a.o_lx + j : kind == 1 ? kr + a.o_lx + nv + j : a.unconstr ? kr + a.o_lu + j : cr + a.o_la + j;
        if (!(impact && dyn && kind == 1)) *dst += a.scale * wsum;
      }
    }
  }
  if (aug) {
    __syncthreads();
    double* const kr = a.kkt + rec * a.kkt_stride;
    if (lane < g.dimf) cr[a.o_lf + lane] -= slf[lane];
    if (a.unconstr) {
      if (lane < nv) cr[a.o_la + lane] -= a.scale * sbeta[lane];                         // lu -= dt beta (:63)
    } else if (!impact && lane < nu) kr[a.o_lu + lane] -= sbeta[nv - nu + lane];           // lu -= beta (actuated part)
    if (nu < nv && lane < nv - nu) cr[a.o_lup + lane] = impact ? 0.0 : sr[a.o_nup + lane] - sbeta[lane];  // lu_passive
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// UnconstrIntermediateStage / UnconstrTerminalStage::evalKKT up to the dynamics (reference src/unconstr/
// unconstr_intermediate_stage.cpp:64-78, unconstr_terminal_stage.cpp): kkt_matrix / kkt_residual.setZero(), the
// ConfigurationSpaceCost of a fixed-base robot (src/cost/configuration_space_cost.cpp:274-324 stage, :343-378 terminal:
// diagonal weights, q - q_ref Euclidean) and linearizeUnconstrForwardEuler (src/dynamics/unconstr_state_equation.cpp:8-24,
// 56-62).  rtoc_linearize in unconstrained mode then adds the dynamics terms.  Records in the convention of
// rtoc_unconstr_condense: KKT.Quu := Qaa, KKT.lu := la, CDD.Qaa := diag(Quu), CDD.la := lu.
// cost: [q_ref | v_ref | u_ref | wq | wv | wa | wu | wq_terminal | wv_terminal | impact weights x 3], nv + 1 doubles each
// (the table of contact_eval_kkt.hpp).
struct UkArgs {
  const double* sol;
  double* kkt;
  double* cdd;
  const double* cost;
  const double* x0;   // [batch][2 nv] initial state of the horizon (q, v of updateSolution(t, q, v)); may be nullptr
  double* dx0;        // [batch][2 nv] computeInitialStateDirection: x0 - s[0].x
  int nstages, batch, nv;
  double dt;
  int sol_stride, kkt_stride, cdd_stride;
  int o_q, o_v, o_a, o_u, o_lmd, o_gmm;
  int o_qxx, o_qxu, o_quu, o_fx, o_lx, o_lu;
  int o_qaa, o_la, o_mj;
  double* cost_out;   // [batch][nstages] value of the stage / terminal cost (the line search's evalOCP reads it), or nullptr
};

static __global__ __launch_bounds__(64) void unconstr_eval_kkt_kernel(UkArgs a) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x / a.nstages, st = blockIdx.x % a.nstages;
  if (b >= a.batch) return;
  const int nv = a.nv, nx = 2 * nv;
  const bool terminal = st == a.nstages - 1;
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  const double* const sn = s + a.sol_stride;  // next grid point (not read on the terminal one)
  double* const kr = a.kkt + rec * a.kkt_stride;
  double* const cr = a.cdd + rec * a.cdd_stride;
  const int M = nv + 1;
  const double *qr = a.cost, *vr = qr + M, *ur = vr + M, *wq = ur + M, *wv = wq + M, *wa = wv + M, *wu = wa + M,
               *wqf = wu + M, *wvf = wqf + M;
  const double dt = a.dt;
  // Hessian blocks: zero, then the diagonals (Qqq, Qvv; Qaa in the Quu slot)
  for (int e = lane; e < nx * nx; e += 64) {
    const int r = e % nx, c = e / nx;
    double v = 0.0;
    if (r == c) v = terminal ? (r < nv ? wqf[r] : wvf[r - nv]) : dt * (r < nv ? wq[r] : wv[r - nv]);
    kr[a.o_qxx + e] = v;
  }
  for (int e = lane; e < nx * nv; e += 64) kr[a.o_qxu + e] = 0.0;
  for (int e = lane; e < nv * nv; e += 64) kr[a.o_quu + e] = (!terminal && e % nv == e / nv) ? dt * wa[e % nv] : 0.0;
  for (int e = lane; e < nv * nv; e += 64) cr[a.o_mj + e] = 0.0;  // off-diagonal part of Quu: a diagonal cost
  for (int i = lane; i < nv; i += 64) {
    const double q = s[a.o_q + i], v = s[a.o_v + i], lmd = s[a.o_lmd + i], gmm = s[a.o_gmm + i];
    if (terminal) {
      kr[a.o_lx + i] = wqf[i] * (q - qr[i]) - lmd;        // evalTerminalCostDerivatives + ...ForwardEulerTerminal
      kr[a.o_lx + nv + i] = wvf[i] * (v - vr[i]) - gmm;
      kr[a.o_fx + i] = 0.0, kr[a.o_fx + nv + i] = 0.0;
      kr[a.o_lu + i] = 0.0;
      cr[a.o_qaa + i] = 0.0, cr[a.o_la + i] = 0.0;
    } else {
      const double acc = s[a.o_a + i], u = s[a.o_u + i];
      const double qn = sn[a.o_q + i], vn = sn[a.o_v + i], lmdn = sn[a.o_lmd + i], gmmn = sn[a.o_gmm + i];
      kr[a.o_fx + i] = q + dt * v - qn;                                            // Fq (:60)
      kr[a.o_fx + nv + i] = v + dt * acc - vn;                                     // Fv (:61)
      kr[a.o_lx + i] = dt * wq[i] * (q - qr[i]) + (lmdn - lmd);                    // lq
      kr[a.o_lx + nv + i] = dt * wv[i] * (v - vr[i]) + (dt * lmdn + gmmn - gmm);   // lv
      kr[a.o_lu + i] = dt * wa[i] * acc + dt * gmmn;                               // la (in the lu slot)
      cr[a.o_la + i] = dt * wu[i] * (u - ur[i]);                                   // lu (in CDD.la)
      cr[a.o_qaa + i] = dt * wu[i];                                                // diag(Quu)
    }
  }
  if (a.cost_out) {
    // ConfigurationSpaceCost::evalStageCost / evalTerminalCost (configuration_space_cost.cpp:251-271, :323-338): the VALUE
    double l = 0.0;
    for (int i = lane; i < nv; i += 64) {
      const double dq = s[a.o_q + i] - qr[i], dv = s[a.o_v + i] - vr[i];
      if (terminal) {
        l += wqf[i] * dq * dq + wvf[i] * dv * dv;
      } else {
        const double acc = s[a.o_a + i], du = s[a.o_u + i] - ur[i];
        l += wq[i] * dq * dq + wv[i] * dv * dv + wa[i] * acc * acc + wu[i] * du * du;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
    if (lane == 0) a.cost_out[rec] = (terminal ? 0.5 : 0.5 * dt) * l;
  }
  if (st == 0 && a.x0 && a.dx0) {
    for (int i = lane; i < nv; i += 64) {
      a.dx0[(size_t)b * nx + i] = a.x0[(size_t)b * nx + i] - s[a.o_q + i];
      a.dx0[(size_t)b * nx + nv + i] = a.x0[(size_t)b * nx + nv + i] - s[a.o_v + i];
    }
  }
}

}  // namespace rbd
}  // namespace rtoc

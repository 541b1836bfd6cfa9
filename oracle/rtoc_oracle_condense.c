/*
 * rtoc_oracle_condense.c -- CPU restatement of robotoc's per-stage KKT condensation /
 * expansion (TEST INFRASTRUCTURE ONLY, see rtoc_oracle.c for the rules and the parity status).
 *
 * Follows, line by line:
 *   Robot::computeMJtJinv            include/robotoc/robot/robot.hxx:642-684
 *                                    (Pinocchio's sparse Cholesky of M is replaced by a dense
 *                                    LLT: same matrix, different elimination order; validated by
 *                                    the defining identity [[M,J^T],[J,0]] * MJtJinv = I)
 *   condenseContactDynamics          src/dynamics/contact_dynamics.cpp:55-164
 *   expandContactDynamicsPrimal/Dual src/dynamics/contact_dynamics.cpp:167-202
 *   condenseImpactDynamics           src/dynamics/impact_dynamics.cpp:38-80
 *   expandImpactDynamicsPrimal/Dual  src/dynamics/impact_dynamics.cpp:83-96
 *   IntermediateStage::evalKKT tail  src/ocp/intermediate_stage.cpp:140-148
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/rtoc.h"

#define AT(A, ld, i, j) (A)[(i) + (size_t)(j) * (ld)]

static int c_llt(double* A, int n, int lda) {
  int bad = 0;
  for (int j = 0; j < n; ++j) {
    double d = AT(A, lda, j, j);
    for (int k = 0; k < j; ++k) d -= AT(A, lda, j, k) * AT(A, lda, j, k);
    if (!(d > 0.0)) bad = 1;
    const double ljj = sqrt(d);
    AT(A, lda, j, j) = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = AT(A, lda, i, j);
      for (int k = 0; k < j; ++k) v -= AT(A, lda, i, k) * AT(A, lda, j, k);
      AT(A, lda, i, j) = v / ljj;
    }
  }
  return bad;
}

static void c_llt_solve(const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
  for (int c = 0; c < nrhs; ++c) {
    double* b = B + (size_t)c * ldb;
    for (int i = 0; i < n; ++i) {
      double v = b[i];
      for (int k = 0; k < i; ++k) v -= AT(L, ldl, i, k) * b[k];
      b[i] = v / AT(L, ldl, i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = b[i];
      for (int k = i + 1; k < n; ++k) v -= AT(L, ldl, k, i) * b[k];
      b[i] = v / AT(L, ldl, i, i);
    }
  }
}

/* C(MxN) = beta*C + alpha*op(A)op(B), generic strides; small sizes */
static void c_gemm(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                   const double* B, int ldb, double beta, double* C, int ldc) {
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < M; ++i) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) {
        const double a = ta ? AT(A, lda, k, i) : AT(A, lda, i, k);
        const double b = tb ? AT(B, ldb, j, k) : AT(B, ldb, k, j);
        acc += a * b;
      }
      AT(C, ldc, i, j) = (beta == 0.0 ? 0.0 : beta * AT(C, ldc, i, j)) + alpha * acc;
    }
}

/* Robot::computeMJtJinv, robot.hxx:642-684.  M nv x nv (ld nv), J nf x nv (ld ldj),
 * out (nv+nf) x (nv+nf) with ld ldo.  Returns nonzero if a factorisation failed. */
int orc_compute_MJtJinv(int nv, int nf, const double* M, const double* J, int ldj, double damping,
                        double* out, int ldo) {
  int bad = 0;
  double* Lm = (double*)malloc(sizeof(double) * nv * nv);
  double* Minv = (double*)calloc((size_t)nv * nv, sizeof(double));
  memcpy(Lm, M, sizeof(double) * nv * nv);
  bad |= c_llt(Lm, nv, nv);
  for (int i = 0; i < nv; ++i) Minv[i + (size_t)i * nv] = 1.0;
  c_llt_solve(Lm, nv, nv, Minv, nv, nv); /* topLeft = M^-1 (:674-676) */
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nv; ++i) AT(out, ldo, i, j) = Minv[i + (size_t)j * nv];
  if (nf > 0) {
    double* JMinv = (double*)malloc(sizeof(double) * nf * nv);
    double* S = (double*)malloc(sizeof(double) * nf * nf);
    double* BR = (double*)calloc((size_t)nf * nf, sizeof(double));
    c_gemm(0, 0, nf, nv, nv, 1.0, J, ldj, Minv, nv, 0.0, JMinv, nf);   /* bottomLeft = J M^-1 (:677) */
    c_gemm(0, 1, nf, nf, nv, 1.0, JMinv, nf, J, ldj, 0.0, S, nf);      /* JMinvJt (:660-661) */
    for (int i = 0; i < nf; ++i) S[i + (size_t)i * nf] += damping;     /* (:662-664) */
    bad |= c_llt(S, nf, nf);                                           /* (:665) */
    for (int i = 0; i < nf; ++i) BR[i + (size_t)i * nf] = -1.0;        /* bottomRight = -I (:673) */
    c_llt_solve(S, nf, nf, BR, nf, nf);                                /* = -(JMinvJt)^-1 (:675) */
    /* topRight = bottomLeft^T * (-bottomRight)  (:678) */
    for (int j = 0; j < nf; ++j)
      for (int i = 0; i < nv; ++i) {
        double acc = 0.0;
        for (int k = 0; k < nf; ++k) acc += JMinv[k + (size_t)i * nf] * (-BR[k + (size_t)j * nf]);
        AT(out, ldo, i, nv + j) = acc;
      }
    /* topLeft -= topRight * bottomLeft (:679) */
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nv; ++i) {
        double acc = 0.0;
        for (int k = 0; k < nf; ++k) acc += AT(out, ldo, i, nv + k) * JMinv[k + (size_t)j * nf];
        AT(out, ldo, i, j) -= acc;
      }
    /* bottomLeft = topRight^T (:680) ; bottomRight */
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nf; ++i) AT(out, ldo, nv + i, j) = AT(out, ldo, j, nv + i);
    for (int j = 0; j < nf; ++j)
      for (int i = 0; i < nf; ++i) AT(out, ldo, nv + i, nv + j) = BR[i + (size_t)j * nf];
    free(JMinv); free(S); free(BR);
  }
  free(Lm); free(Minv);
  return bad;
}

typedef struct {
  double *dIDda, *D, *dCda, *IDC, *Qaa, *Qff, *Qqf, *la, *lf, *ha, *hf, *Phia, *lup;
  double *Lam, *LD, *Lr, *Qafqv, *Qafu, *laf, *Qxup, *Quuptr, *haf;
} cdd_view;

static cdd_view cdd_at(const rtoc_layout* L, double* r) {
  const int* o = L->cdd.off;
  cdd_view v = {r + o[RTOC_CDD_DIDDA], r + o[RTOC_CDD_DIDCDQV], r + o[RTOC_CDD_DCDA],
                r + o[RTOC_CDD_IDC],   r + o[RTOC_CDD_QAA],     r + o[RTOC_CDD_QFF],
                r + o[RTOC_CDD_QQF],   r + o[RTOC_CDD_LA],      r + o[RTOC_CDD_LF],
                r + o[RTOC_CDD_HA],    r + o[RTOC_CDD_HF],      r + o[RTOC_CDD_PHIA],
                r + o[RTOC_CDD_LUP],   r + o[RTOC_CDD_MJTJINV], r + o[RTOC_CDD_MJD],
                r + o[RTOC_CDD_MJIDC], r + o[RTOC_CDD_QAFQV],   r + o[RTOC_CDD_QAFU],
                r + o[RTOC_CDD_LAF],   r + o[RTOC_CDD_QXUP],    r + o[RTOC_CDD_QUUPTR],
                r + o[RTOC_CDD_HAF]};
  return v;
}

/* condenseContactDynamics (contact_dynamics.cpp:55-164) followed by the STO scalings of
 * IntermediateStage::evalKKT (intermediate_stage.cpp:140-148).  One stage, in place. */
unsigned orc_condense_stage(const rtoc_layout* L, const rtoc_grid* g, double* kkt_rec,
                            double* cdd_rec, double damping) {
  const int nv = L->dims.nv, nu = L->dims.nu, np = L->dims.np, nx = L->nx;
  const int nf = g->dimf, nvf = nv + nf, ns = g->dims;
  const int ldv = L->nvf_max, ldf = L->dims.nf_max, lds = L->dims.ns_max;
  const double dt = g->dt;
  const int* ko = L->kkt.off;
  double* Fxx = kkt_rec + ko[RTOC_KKT_FXX];
  double* Fvu = kkt_rec + ko[RTOC_KKT_FVU];
  double* Qxx = kkt_rec + ko[RTOC_KKT_QXX];
  double* Qxu = kkt_rec + ko[RTOC_KKT_QXU];
  double* Quu = kkt_rec + ko[RTOC_KKT_QUU];
  double* Fx = kkt_rec + ko[RTOC_KKT_FX];
  double* lx = kkt_rec + ko[RTOC_KKT_LX];
  double* lu = kkt_rec + ko[RTOC_KKT_LU];
  double* hx = kkt_rec + ko[RTOC_KKT_HX];
  double* hu = kkt_rec + ko[RTOC_KKT_HU];
  double* fx = kkt_rec + ko[RTOC_KKT_FFX];
  double* scal = kkt_rec + ko[RTOC_KKT_SCAL];
  double* Phix = kkt_rec + ko[RTOC_KKT_PHIX];
  double* Phiu = kkt_rec + ko[RTOC_KKT_PHIU];
  double* Phit = kkt_rec + ko[RTOC_KKT_PHIT];
  double* Pres = kkt_rec + ko[RTOC_KKT_PRES];
  cdd_view c = cdd_at(L, cdd_rec);
  unsigned stat = 0;
  /* :63-65 */
  if (orc_compute_MJtJinv(nv, nf, c.dIDda, c.dCda, ldf, damping, c.Lam, ldv)) stat |= RTOC_STAT_M_NOT_SPD;
  c_gemm(0, 0, nvf, nx, nvf, 1.0, c.Lam, ldv, c.D, ldv, 0.0, c.LD, ldv);
  c_gemm(0, 0, nvf, 1, nvf, 1.0, c.Lam, ldv, c.IDC, nvf, 0.0, c.Lr, nvf);
  /* Qafqv :67-74 */
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nv; ++i) AT(c.Qafqv, ldv, i, j) = -c.Qaa[i] * AT(c.LD, ldv, i, j);
  if (nf > 0) {
    c_gemm(0, 0, nf, nx, nf, -1.0, c.Qff, ldf, c.LD + nv, ldv, 0.0, c.Qafqv + nv, ldv);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nf; ++i) AT(c.Qafqv, ldv, nv + i, j) -= AT(c.Qqf, nv, j, i);
  }
  /* Qafu_full :75-80 */
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nv; ++i) AT(c.Qafu, ldv, i, j) = c.Qaa[i] * AT(c.Lam, ldv, i, j);
  if (nf > 0) c_gemm(0, 0, nf, nv, nf, 1.0, c.Qff, ldf, c.Lam + nv, ldv, 0.0, c.Qafu + nv, ldv);
  /* laf :81-88 */
  for (int i = 0; i < nv; ++i) c.laf[i] = c.la[i] - c.Qaa[i] * c.Lr[i];
  for (int i = 0; i < nf; ++i) {
    double acc = 0.0;
    for (int k = 0; k < nf; ++k) acc += AT(c.Qff, ldf, i, k) * c.Lr[nv + k];
    c.laf[nv + i] = -c.lf[i] - acc;
  }
  /* Qxx :90-93 */
  c_gemm(1, 0, nx, nx, nvf, -1.0, c.LD, ldv, c.Qafqv, ldv, 1.0, Qxx, nx);
  if (nf > 0) c_gemm(0, 0, nv, nx, nf, 1.0, c.Qqf, nv, c.LD + nv, ldv, 1.0, Qxx, nx);
  /* Qxu (+ passive part) :94-109 */
  if (np > 0) {
    c_gemm(1, 0, nx, np, nvf, -1.0, c.LD, ldv, c.Qafu, ldv, 0.0, c.Qxup, nx);
    if (nf > 0) c_gemm(0, 0, nv, np, nf, -1.0, c.Qqf, nv, c.Lam + nv, ldv, 1.0, c.Qxup, nx);
  }
  c_gemm(1, 0, nx, nu, nvf, -1.0, c.LD, ldv, c.Qafu + (size_t)np * ldv, ldv, 1.0, Qxu, nx);
  if (nf > 0)
    c_gemm(0, 0, nv, nu, nf, -1.0, c.Qqf, nv, c.Lam + nv + (size_t)np * ldv, ldv, 1.0, Qxu, nx);
  /* lx :110-113 */
  c_gemm(1, 0, nx, 1, nvf, -1.0, c.LD, ldv, c.laf, nvf, 1.0, lx, nx);
  if (nf > 0) c_gemm(0, 0, nv, 1, nf, 1.0, c.Qqf, nv, c.Lr + nv, nf, 1.0, lx, nx);
  /* Quu, lu (+ passive) :115-130 */
  if (np > 0) {
    c_gemm(0, 0, np, nu, nvf, 1.0, c.Lam, ldv, c.Qafu + (size_t)np * ldv, ldv, 0.0, c.Quuptr, np);
    c_gemm(0, 0, np, 1, nvf, 1.0, c.Lam, ldv, c.laf, nvf, 1.0, c.lup, np);
  }
  c_gemm(0, 0, nu, nu, nvf, 1.0, c.Lam + np, ldv, c.Qafu + (size_t)np * ldv, ldv, 1.0, Quu, nu);
  c_gemm(0, 0, nu, 1, nvf, 1.0, c.Lam + np, ldv, c.laf, nvf, 1.0, lu, nu);
  /* dynamics :132-136 */
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nv; ++i) {
      AT(Fxx, nx, nv + i, j) = -dt * AT(c.LD, ldv, i, j);
      AT(Fxx, nx, nv + i, nv + j) = -dt * AT(c.LD, ldv, i, nv + j) + (i == j ? 1.0 : 0.0);
    }
  for (int j = 0; j < nu; ++j)
    for (int i = 0; i < nv; ++i) AT(Fvu, nv, i, j) = dt * AT(c.Lam, ldv, i, np + j);
  for (int i = 0; i < nv; ++i) Fx[nv + i] -= dt * c.Lr[i];
  /* switching constraint :138-153 */
  if (ns > 0) {
    c_gemm(0, 0, ns, nx, nv, -1.0, c.Phia, lds, c.LD, ldv, 1.0, Phix, lds);
    c_gemm(0, 0, ns, nu, nv, 1.0, c.Phia, lds, c.Lam + (size_t)np * ldv, ldv, 0.0, Phiu, lds);
    for (int i = 0; i < ns; ++i) {
      double acc = 0.0;
      for (int k = 0; k < nv; ++k) acc += AT(c.Phia, lds, i, k) * c.Lr[k];
      Phit[i] -= acc;
      Pres[i] -= acc;
    }
  }
  /* STO sensitivities :156-163 */
  for (int i = 0; i < nv; ++i) c.haf[i] = c.ha[i];
  for (int i = 0; i < nf; ++i) c.haf[nv + i] = -c.hf[i];
  {
    double acc = 0.0;
    for (int i = 0; i < nvf; ++i) acc += c.Lr[i] * c.haf[i];
    scal[RTOC_KKT_SCAL_H] -= acc;
  }
  c_gemm(1, 0, nx, 1, nvf, -1.0, c.LD, ldv, c.haf, nvf, 1.0, hx, nx);
  if (nf > 0) c_gemm(0, 0, nv, 1, nf, 1.0 / dt, c.Qqf, nv, c.Lr + nv, nf, 1.0, hx, nx);
  c_gemm(0, 0, nu, 1, nvf, 1.0, c.Lam + np, ldv, c.haf, nvf, 1.0, hu, nu);
  /* IntermediateStage::evalKKT tail, intermediate_stage.cpp:140-148 */
  {
    const double inv = 1.0 / (double)g->num_grids_in_phase;
    scal[RTOC_KKT_SCAL_H] *= inv;
    for (int i = 0; i < nx; ++i) hx[i] *= inv;
    for (int i = 0; i < nu; ++i) hu[i] *= inv;
    for (int i = 0; i < nx; ++i) fx[i] *= inv;
    scal[RTOC_KKT_SCAL_QTT] *= inv * inv;
    scal[RTOC_KKT_SCAL_QTT_PREV] = -scal[RTOC_KKT_SCAL_QTT];
    if (g->switching_constraint)
      for (int i = 0; i < ns; ++i) Phit[i] *= inv;
  }
  return stat;
}

/* condenseImpactDynamics, impact_dynamics.cpp:38-80.  dIDda slot = dIDddv, dCda slot = dCdv
 * (the reference reads dCdv() = dIDCdqv().bottomRightCorner: the D block rows nv.., cols nv..),
 * Qaa slot = Qdvdv.diagonal(), la slot = ldv. */
unsigned orc_condense_impact_stage(const rtoc_layout* L, const rtoc_grid* g, double* kkt_rec,
                                   double* cdd_rec, double damping) {
  const int nv = L->dims.nv, nx = L->nx;
  const int nf = g->dimf, nvf = nv + nf;
  const int ldv = L->nvf_max, ldf = L->dims.nf_max;
  const int* ko = L->kkt.off;
  double* Fxx = kkt_rec + ko[RTOC_KKT_FXX];
  double* Qxx = kkt_rec + ko[RTOC_KKT_QXX];
  double* Fx = kkt_rec + ko[RTOC_KKT_FX];
  double* lx = kkt_rec + ko[RTOC_KKT_LX];
  cdd_view c = cdd_at(L, cdd_rec);
  unsigned stat = 0;
  const double* dCdv = c.D + nv + (size_t)nv * ldv; /* nf x nv, ld ldv */
  if (orc_compute_MJtJinv(nv, nf, c.dIDda, dCdv, ldv, damping, c.Lam, ldv)) stat |= RTOC_STAT_M_NOT_SPD;
  /* :44-50: left half dense product, right half only through dCdv */
  c_gemm(0, 0, nvf, nv, nvf, 1.0, c.Lam, ldv, c.D, ldv, 0.0, c.LD, ldv);
  c_gemm(0, 0, nv, nv, nf, 1.0, c.Lam + (size_t)nv * ldv, ldv, dCdv, ldv, 0.0, c.LD + (size_t)nv * ldv, ldv);
  c_gemm(0, 0, nf, nv, nf, 1.0, c.Lam + nv + (size_t)nv * ldv, ldv, dCdv, ldv, 0.0,
         c.LD + nv + (size_t)nv * ldv, ldv);
  c_gemm(0, 0, nvf, 1, nvf, 1.0, c.Lam, ldv, c.IDC, nvf, 0.0, c.Lr, nvf);
  /* :52-63 */
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nv; ++i) AT(c.Qafqv, ldv, i, j) = -c.Qaa[i] * AT(c.LD, ldv, i, j);
  c_gemm(0, 0, nf, nx, nf, -1.0, c.Qff, ldf, c.LD + nv, ldv, 0.0, c.Qafqv + nv, ldv);
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nf; ++i) AT(c.Qafqv, ldv, nv + i, j) -= AT(c.Qqf, nv, j, i);
  for (int i = 0; i < nv; ++i) c.laf[i] = c.la[i] - c.Qaa[i] * c.Lr[i];
  for (int i = 0; i < nf; ++i) {
    double acc = 0.0;
    for (int k = 0; k < nf; ++k) acc += AT(c.Qff, ldf, i, k) * c.Lr[nv + k];
    c.laf[nv + i] = -c.lf[i] - acc;
  }
  /* :65-72 */
  c_gemm(1, 0, nx, nx, nvf, -1.0, c.LD, ldv, c.Qafqv, ldv, 1.0, Qxx, nx);
  c_gemm(0, 0, nv, nx, nf, 1.0, c.Qqf, nv, c.LD + nv, ldv, 1.0, Qxx, nx);
  c_gemm(1, 0, nx, 1, nvf, -1.0, c.LD, ldv, c.laf, nvf, 1.0, lx, nx);
  c_gemm(0, 0, nv, 1, nf, 1.0, c.Qqf, nv, c.Lr + nv, nf, 1.0, lx, nx);
  /* :74-77 */
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nv; ++i) {
      AT(Fxx, nx, nv + i, j) = -AT(c.LD, ldv, i, j);
      AT(Fxx, nx, nv + i, nv + j) = (i == j ? 1.0 : 0.0) - AT(c.LD, ldv, i, nv + j);
    }
  for (int i = 0; i < nv; ++i) Fx[nv + i] -= c.Lr[i];
  return stat;
}

/* expandContactDynamicsPrimal + expandContactDynamicsDual (contact_dynamics.cpp:167-202) and the
 * impact forms (impact_dynamics.cpp:83-96).  dgmm_next = d_next.dlmdgmm[nv:], dts as computed by
 * IntermediateStage::expandDual (intermediate_stage.cpp:172-180). */
void orc_expand_stage(const rtoc_layout* L, const rtoc_grid* g, double* cdd_rec, double* dir_rec,
                      const double* dir_next_rec) {
  const int nv = L->dims.nv, nu = L->dims.nu, np = L->dims.np, nx = L->nx;
  const int nf = g->dimf, nvf = nv + nf, ns = g->dims;
  const int ldv = L->nvf_max, lds = L->dims.ns_max;
  const int impact = g->type == RTOC_GRID_IMPACT;
  const double dt = g->dt;
  cdd_view c = cdd_at(L, cdd_rec);
  const int* o = L->dir.off;
  double* dx = dir_rec + o[RTOC_DIR_DX];
  double* du = dir_rec + o[RTOC_DIR_DU];
  double* dxi = dir_rec + o[RTOC_DIR_DXI];
  double* dts = dir_rec + o[RTOC_DIR_DTS];
  double* daf = dir_rec + o[RTOC_DIR_DAF];
  double* dbetamu = dir_rec + o[RTOC_DIR_DBETAMU];
  double* dnup = dir_rec + o[RTOC_DIR_DNUP];
  const double* dgmm_next = dir_next_rec + o[RTOC_DIR_DLMDGMM] + nv;
  /* primal :167-174 / impact :83-88 */
  c_gemm(0, 0, nvf, 1, nx, -1.0, c.LD, ldv, dx, nx, 0.0, daf, nvf);
  if (!impact) c_gemm(0, 0, nvf, 1, nu, 1.0, c.Lam + (size_t)np * ldv, ldv, du, nu, 1.0, daf, nvf);
  for (int i = 0; i < nvf; ++i) daf[i] -= c.Lr[i];
  for (int i = 0; i < nf; ++i) daf[nv + i] *= -1.0;
  /* dual :177-201 / impact :91-96 */
  if (!impact && np > 0) {
    for (int i = 0; i < np; ++i) dnup[i] = -c.lup[i];
    c_gemm(0, 0, np, 1, nu, -1.0, c.Quuptr, np, du, nu, 1.0, dnup, np);
    c_gemm(1, 0, np, 1, nx, -1.0, c.Qxup, nx, dx, nx, 1.0, dnup, np);
    c_gemm(0, 0, np, 1, nv, -dt, c.Lam, ldv, dgmm_next, nv, 1.0, dnup, np);
  }
  c_gemm(0, 0, nvf, 1, nx, 1.0, c.Qafqv, ldv, dx, nx, 1.0, c.laf, nvf);
  if (!impact) {
    c_gemm(0, 0, nvf, 1, nu, 1.0, c.Qafu + (size_t)np * ldv, ldv, du, nu, 1.0, c.laf, nvf);
    for (int i = 0; i < nv; ++i) c.laf[i] += dt * dgmm_next[i];
    if (ns > 0) c_gemm(1, 0, nv, 1, ns, 1.0, c.Phia, lds, dxi, ns, 1.0, c.laf, nv);
    double dtsv = 0.0;
    if (g->num_grids_in_phase > 0) dtsv = (dts[1] - dts[0]) / (double)g->num_grids_in_phase;
    if (dtsv < -DBL_EPSILON || dtsv > DBL_EPSILON)
      for (int i = 0; i < nvf; ++i) c.laf[i] += dtsv * c.haf[i];
  } else {
    for (int i = 0; i < nv; ++i) c.laf[i] += dgmm_next[i];
  }
  c_gemm(0, 0, nvf, 1, nvf, -1.0, c.Lam, ldv, c.laf, nvf, 0.0, dbetamu, nvf);
}

/* batch drivers */
void orc_condense_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                        double* kkt, double* cdd, double damping, unsigned* stat) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    unsigned st = 0;
    for (int i = 0; i < nstages - 1; ++i) {
      double* kr = kkt + ((size_t)b * nstages + i) * L->kkt.stride;
      double* cr = cdd + ((size_t)b * nstages + i) * L->cdd.stride;
      if (grid[i].type == RTOC_GRID_IMPACT)
        st |= orc_condense_impact_stage(L, &grid[i], kr, cr, damping);
      else
        st |= orc_condense_stage(L, &grid[i], kr, cr, damping);
    }
    if (stat) stat[b] = st;
  }
}

void orc_expand_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                      double* cdd, double* dir) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nstages - 1; ++i) {
      double* cr = cdd + ((size_t)b * nstages + i) * L->cdd.stride;
      double* dr = dir + ((size_t)b * nstages + i) * L->dir.stride;
      orc_expand_stage(L, &grid[i], cr, dr, dr + L->dir.stride);
    }
}

/* ------------------------------------------------------------------------- */
/* PDIPM slack/dual elimination for the joint-limit (box) rows                  */
/*   include/robotoc/constraints/pdipm.hxx:66-69,121-142,176-186                 */
/*   src/constraints/joint_*_limit.cpp (condenseSlackAndDual / expandSlackAndDual) */
/*   stage mask: src/constraints/constraints_data.cpp:20-45                      */
/* ------------------------------------------------------------------------- */
static int row_active(const rtoc_box_row* r, const rtoc_grid* g) {
  if (g->type == RTOC_GRID_IMPACT || g->type == RTOC_GRID_TERMINAL) return 0;
  return g->time_stage >= r->level;
}

/* Constraints::condenseSlackAndDual (constraints.cpp:322-357) for box rows.  cdd_rec: the ContactDynamicsData record
 * whose Qaa diagonal / la the acceleration limits act on (joint_acceleration_lower_limit.cpp:69-77, _upper_limit.cpp:69-77);
 * may be NULL when there are no RTOC_VAR_A rows. */
void orc_pdipm_condense_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows,
                              int nrows, double* kkt_rec, double* con_rec, double* cdd_rec) {
  const int nv = L->dims.nv, nu = L->dims.nu, nx = L->nx;
  const int* ko = L->kkt.off;
  const int* o = L->con.off;
  double* Qxx = kkt_rec + ko[RTOC_KKT_QXX];
  double* Quu = kkt_rec + ko[RTOC_KKT_QUU];
  double* lx = kkt_rec + ko[RTOC_KKT_LX];
  double* lu = kkt_rec + ko[RTOC_KKT_LU];
  for (int r = 0; r < nrows; ++r) {
    if (!row_active(&rows[r], g)) continue;
    const double slack = con_rec[o[RTOC_CON_SLACK] + r], dual = con_rec[o[RTOC_CON_DUAL] + r];
    const double res = con_rec[o[RTOC_CON_RESIDUAL] + r], cmpl = con_rec[o[RTOC_CON_CMPL] + r];
    const double cond = (dual * res - cmpl) / slack; /* pdipm.hxx:66-69 */
    con_rec[o[RTOC_CON_COND] + r] = cond;
    const int idx = rows[r].index;
    if (rows[r].var == RTOC_VAR_U) {
      AT(Quu, nu, idx, idx) += dual / slack;
      lu[idx] += rows[r].sign * cond;
    } else if (rows[r].var == RTOC_VAR_A) {
      cdd_rec[L->cdd.off[RTOC_CDD_QAA] + idx] += dual / slack; /* Qaa.diagonal().tail(dimc) += dual / slack */
      cdd_rec[L->cdd.off[RTOC_CDD_LA] + idx] += rows[r].sign * cond; /* la.tail(dimc) -/+= cond */
    } else {
      const int k = rows[r].var == RTOC_VAR_V ? nv + idx : idx;
      AT(Qxx, nx, k, k) += dual / slack;
      lx[k] += rows[r].sign * cond;
    }
  }
}

/* Constraints::expandSlackAndDual + maxSlackStepSize / maxDualStepSize
 * (constraints.cpp:360-458, pdipm.hxx:121-142,176-186). steps[0] = primal, steps[1] = dual
 * are min-reduced in place (start them at 1). */
void orc_pdipm_expand_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows,
                            int nrows, const double* dir_rec, double* con_rec, double tau,
                            double* steps) {
  const int nv = L->dims.nv;
  const int* o = L->con.off;
  const double* dx = dir_rec + L->dir.off[RTOC_DIR_DX];
  const double* du = dir_rec + L->dir.off[RTOC_DIR_DU];
  for (int r = 0; r < nrows; ++r) {
    if (!row_active(&rows[r], g)) continue;
    const int idx = rows[r].index;
    /* (acceleration limits: dslack = +/- d.da().tail(dimc) - residual, joint_acceleration_lower_limit.cpp:80-85) */
    const double dz = rows[r].var == RTOC_VAR_U ? du[idx]
                                                : (rows[r].var == RTOC_VAR_V ? dx[nv + idx]
                                                                             : (rows[r].var == RTOC_VAR_A ? dir_rec[L->dir.off[RTOC_DIR_DAF] + idx] : dx[idx]));
    const double slack = con_rec[o[RTOC_CON_SLACK] + r], dual = con_rec[o[RTOC_CON_DUAL] + r];
    const double res = con_rec[o[RTOC_CON_RESIDUAL] + r], cmpl = con_rec[o[RTOC_CON_CMPL] + r];
    const double dslack = -rows[r].sign * dz - res;
    const double ddual = -(dual * dslack + cmpl) / slack;
    con_rec[o[RTOC_CON_DSLACK] + r] = dslack;
    con_rec[o[RTOC_CON_DDUAL] + r] = ddual;
    const double fs = -tau * (slack / dslack), fd = -tau * (dual / ddual);
    if (fs > 0 && fs < 1 && fs < steps[0]) steps[0] = fs;
    if (fd > 0 && fd < 1 && fd < steps[1]) steps[1] = fd;
  }
}

/* updateSlack / updateDual (constraints_impl.hxx:167-182) */
void orc_pdipm_update_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows, int nrows,
                            double* con_rec, double primal_step, double dual_step) {
  const int* o = L->con.off;
  for (int r = 0; r < nrows; ++r) {
    if (!row_active(&rows[r], g)) continue;
    con_rec[o[RTOC_CON_SLACK] + r] += primal_step * con_rec[o[RTOC_CON_DSLACK] + r];
    con_rec[o[RTOC_CON_DUAL] + r] += dual_step * con_rec[o[RTOC_CON_DDUAL] + r];
  }
}

void orc_pdipm_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                     const rtoc_box_row* rows, int nrows, double* kkt, double* con, double* dir,
                     double tau, double* steps, int phase, double* cdd) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    if (phase == 1) {
      steps[2 * b] = 1.0;
      steps[2 * b + 1] = 1.0;
    }
    for (int i = 0; i < nstages - 1; ++i) {
      double* kr = kkt ? kkt + ((size_t)b * nstages + i) * L->kkt.stride : 0;
      double* cr = con + ((size_t)b * nstages + i) * L->con.stride;
      double* dr = dir ? dir + ((size_t)b * nstages + i) * L->dir.stride : 0;
      if (phase == 0) orc_pdipm_condense_stage(L, &grid[i], rows, nrows, kr, cr, cdd ? cdd + ((size_t)b * nstages + i) * L->cdd.stride : 0);
      if (phase == 1) orc_pdipm_expand_stage(L, &grid[i], rows, nrows, dr, cr, tau, steps + 2 * b);
      if (phase == 2) orc_pdipm_update_stage(L, &grid[i], rows, nrows, cr, steps[2 * b], steps[2 * b + 1]);
    }
  }
}

/* ======================================================================================
 * Floating-base corrections of the linearised state equation
 * (src/dynamics/state_equation.cpp:68-109, src/dynamics/impact_state_equation.cpp:57-72).
 * se3 record: Fqq_inv (6x6 col-major) at RTOC_SE3_FQQ_INV, Fqq_prev_inv at RTOC_SE3_FQQ_PREV_INV.
 * ====================================================================================== */
static void neg_matmul6(const double* inv, const double* in, int ldin, double* out, int ldout, int ncols) {
  /* out = -inv * in  (state_equation.cpp:81: noalias() = -Fqq_inv * Fqq_tmp) */
  for (int j = 0; j < ncols; ++j)
    for (int i = 0; i < 6; ++i) {
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) acc += inv[i + 6 * k] * in[k + (size_t)ldin * j];
      out[i + (size_t)ldout * j] = -acc;
    }
}

/* correctLinearizeStateEquation (:68-88) / correctLinearizeImpactStateEquation (impact :57-72) */
void orc_correct_state_equation_stage(const rtoc_layout* L, const rtoc_grid* g, const double* se3_rec,
                                      double* kkt_rec) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nx = L->nx, nv = L->dims.nv;
  const double* inv = se3_rec + RTOC_SE3_FQQ_INV;
  double* Fxx = kkt_rec + L->kkt.off[RTOC_KKT_FXX];
  double* Fx = kkt_rec + L->kkt.off[RTOC_KKT_FX];
  double* fx = kkt_rec + L->kkt.off[RTOC_KKT_FFX];
  double tmp[36], v[6];
  for (int j = 0; j < 6; ++j)
    for (int i = 0; i < 6; ++i) tmp[i + 6 * j] = Fxx[i + (size_t)nx * j]; /* Fqq_tmp (:80) */
  neg_matmul6(inv, tmp, 6, Fxx, nx, 6);                                    /* (:81) */
  if (g->type != RTOC_GRID_IMPACT)
    for (int j = 0; j < 6; ++j)
      for (int i = 0; i < 6; ++i) Fxx[i + (size_t)nx * (nv + j)] = -g->dt * inv[i + 6 * j]; /* (:82) */
  for (int i = 0; i < 6; ++i) v[i] = Fx[i];
  neg_matmul6(inv, v, 6, Fx, nx, 1); /* (:83-84) */
  if (g->type != RTOC_GRID_IMPACT) {
    for (int i = 0; i < 6; ++i) v[i] = fx[i];
    neg_matmul6(inv, v, 6, fx, nx, 1); /* (:85-86) */
  }
}

/* correctCostateDirection (:91-96): dlmd.head<6>() = -Fqq_prev_inv^T dlmd.head<6>() */
void orc_correct_costate_stage(const rtoc_layout* L, const double* se3_rec, double* dir_rec) {
  const double* inv = se3_rec + RTOC_SE3_FQQ_PREV_INV;
  double* dl = dir_rec + L->dir.off[RTOC_DIR_DLMDGMM];
  double t[6];
  for (int i = 0; i < 6; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 6; ++k) acc += inv[k + 6 * i] * dl[k];
    t[i] = acc;
  }
  for (int i = 0; i < 6; ++i) dl[i] = -t[i];
}

/* computeInitialStateDirection, floating-base part (:99-109) */
void orc_initial_state_direction(const double* se3_rec0, double* dx0) {
  const double* inv = se3_rec0 + RTOC_SE3_FQQ_PREV_INV;
  double t[6];
  for (int i = 0; i < 6; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 6; ++k) acc += inv[i + 6 * k] * dx0[k];
    t[i] = acc;
  }
  for (int i = 0; i < 6; ++i) dx0[i] = -t[i];
}

void orc_state_correction_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                                const double* se3, double* kkt, double* dir, double* dx0) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    for (int i = 0; i < nstages; ++i) {
      const double* sr = se3 + ((size_t)b * nstages + i) * RTOC_SE3_STRIDE;
      if (kkt) orc_correct_state_equation_stage(L, &grid[i], sr, kkt + ((size_t)b * nstages + i) * L->kkt.stride);
      if (dir) orc_correct_costate_stage(L, sr, dir + ((size_t)b * nstages + i) * L->dir.stride);
    }
    if (dx0) orc_initial_state_direction(se3 + (size_t)b * nstages * RTOC_SE3_STRIDE, dx0 + (size_t)b * L->nx);
  }
}

/* ======================================================================================
 * UnconstrDynamics (src/dynamics/unconstr_dynamics.cpp:67-104).  Record conventions as in
 * include/rtoc.h (rtoc_unconstr_condense): KKT.Quu/lu/Qxu = Qaa/la/[Qqa;Qva]; CDD.dIDCdqv =
 * [dID_dq | dID_dv], CDD.dIDda = dID_da, CDD.IDC = ID, CDD.Qaa = diag(Quu), CDD.la = lu (torques).
 * ====================================================================================== */
void orc_unconstr_condense_stage(const rtoc_layout* L, double* kkt_rec, const double* cdd_rec) {
  const int nv = L->dims.nv, nx = L->nx;
  const double* dq = cdd_rec + L->cdd.off[RTOC_CDD_DIDCDQV];
  const double* dv = dq + (size_t)nv * nv;
  const double* da = cdd_rec + L->cdd.off[RTOC_CDD_DIDDA];
  const double* ID = cdd_rec + L->cdd.off[RTOC_CDD_IDC];
  const double* w = cdd_rec + L->cdd.off[RTOC_CDD_QAA];
  const double* lut = cdd_rec + L->cdd.off[RTOC_CDD_LA];
  double* Qxx = kkt_rec + L->kkt.off[RTOC_KKT_QXX];
  double* Qxu = kkt_rec + L->kkt.off[RTOC_KKT_QXU];
  double* Qaa = kkt_rec + L->kkt.off[RTOC_KKT_QUU];
  double* lx = kkt_rec + L->kkt.off[RTOC_KKT_LX];
  double* la = kkt_rec + L->kkt.off[RTOC_KKT_LU];
  double luc[64];
  for (int i = 0; i < nv; ++i) luc[i] = lut[i] + w[i] * ID[i]; /* (:70-71) */
  for (int c = 0; c < nv; ++c) {                               /* (:72-74) */
    double aq = 0.0, av = 0.0, aa = 0.0;
    for (int i = 0; i < nv; ++i) {
      aq += dq[i + (size_t)c * nv] * luc[i];
      av += dv[i + (size_t)c * nv] * luc[i];
      aa += da[i + (size_t)c * nv] * luc[i];
    }
    lx[c] += aq;
    lx[nv + c] += av;
    la[c] += aa;
  }
  for (int c = 0; c < nv; ++c)
    for (int r = 0; r < nv; ++r) {
      double qq = 0.0, qv = 0.0, vv = 0.0, aa = 0.0, qu = 0.0, vu = 0.0;
      for (int i = 0; i < nv; ++i) {
        qq += dq[i + (size_t)r * nv] * (w[i] * dq[i + (size_t)c * nv]); /* dID_dq^T Quu_dID_dq (:79) */
        qv += dq[i + (size_t)r * nv] * (w[i] * dv[i + (size_t)c * nv]); /* (:80) */
        vv += dv[i + (size_t)r * nv] * (w[i] * dv[i + (size_t)c * nv]); /* (:82) */
        aa += da[i + (size_t)r * nv] * (w[i] * da[i + (size_t)c * nv]); /* (:83) */
        qu += (w[i] * dq[i + (size_t)r * nv]) * da[i + (size_t)c * nv]; /* Quu_dID_dq^T dID_da (:86) */
        vu += (w[i] * dv[i + (size_t)r * nv]) * da[i + (size_t)c * nv]; /* (:87) */
      }
      const double qqv = Qxx[r + (size_t)(nv + c) * nx] + qv;
      Qxx[r + (size_t)c * nx] += qq;
      Qxx[r + (size_t)(nv + c) * nx] = qqv;
      Qxx[(nv + c) + (size_t)r * nx] = qqv; /* Qvq = Qqv^T (:81) */
      Qxx[(nv + r) + (size_t)(nv + c) * nx] += vv;
      Qaa[r + (size_t)c * nv] += aa;
      Qxu[r + (size_t)c * nx] = qu;
      Qxu[(nv + r) + (size_t)c * nx] = vu;
    }
}

/* expandPrimal (:91-96) + expandDual (:99-104) */
void orc_unconstr_expand_stage(const rtoc_layout* L, const double* cdd_rec, double* dir_rec, double dt) {
  const int nv = L->dims.nv;
  const double* dq = cdd_rec + L->cdd.off[RTOC_CDD_DIDCDQV];
  const double* dv = dq + (size_t)nv * nv;
  const double* da = cdd_rec + L->cdd.off[RTOC_CDD_DIDDA];
  const double* ID = cdd_rec + L->cdd.off[RTOC_CDD_IDC];
  const double* w = cdd_rec + L->cdd.off[RTOC_CDD_QAA];
  const double* lut = cdd_rec + L->cdd.off[RTOC_CDD_LA];
  const double* dx = dir_rec + L->dir.off[RTOC_DIR_DX];
  double* du = dir_rec + L->dir.off[RTOC_DIR_DU];
  double* daf = dir_rec + L->dir.off[RTOC_DIR_DAF];
  double* dbeta = dir_rec + L->dir.off[RTOC_DIR_DBETAMU];
  double dacc[64];
  for (int i = 0; i < nv; ++i) {
    dacc[i] = du[i]; /* the Riccati control is the acceleration direction */
    daf[i] = dacc[i];
  }
  for (int i = 0; i < nv; ++i) {
    double t = ID[i], acc = 0.0;
    for (int k = 0; k < nv; ++k) acc += dq[i + (size_t)k * nv] * dx[k];
    t += acc;
    acc = 0.0;
    for (int k = 0; k < nv; ++k) acc += dv[i + (size_t)k * nv] * dx[nv + k];
    t += acc;
    acc = 0.0;
    for (int k = 0; k < nv; ++k) acc += da[i + (size_t)k * nv] * dacc[k];
    t += acc;
    du[i] = t;
  }
  /* expandDual (unconstr_dynamics.cpp:99-104): the full Quu; diagonal in QAA, off-diagonal part in the MJTJINV field */
  const double* Qfull = cdd_rec + L->cdd.off[RTOC_CDD_MJTJINV];
  for (int i = 0; i < nv; ++i) {
    double off = 0.0;
    for (int k = 0; k < nv; ++k)
      if (k != i) off += Qfull[i + (size_t)k * nv] * du[k];
    dbeta[i] = (lut[i] + (w[i] * du[i] + off)) / dt;
  }
}

void orc_unconstr_dynamics_batch(const rtoc_layout* L, int nstages, int batch, double* kkt, double* cdd,
                                 double* dir, double dt, int expand) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nstages - 1; ++i) {
      const size_t rec = (size_t)b * nstages + i;
      if (!expand)
        orc_unconstr_condense_stage(L, kkt + rec * L->kkt.stride, cdd + rec * L->cdd.stride);
      else
        orc_unconstr_expand_stage(L, cdd + rec * L->cdd.stride, dir + rec * L->dir.stride, dt);
    }
}

/* ======================================================================================
 * Friction-cone PDIPM rows (src/constraints/friction_cone.cpp:194-268; ImpactFrictionCone alike).
 * cone record: include/rtoc_layout.h (RTOC_BUF_CONE), compacted over the active contacts;
 * constraint rows row0 + 5k + j of the con record, row0 = nc_max - 5*max_contacts.
 * ====================================================================================== */
void orc_cone_condense_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                             const double* cone_rec, double* kkt_rec, double* cdd_rec, double* con_rec) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nv = L->dims.nv, nx = L->nx, nfp = L->dims.nf_max > 0 ? L->dims.nf_max : 1;
  const int nact = g->dimf / contact_dim, row0 = L->dims.nc_max - 5 * max_contacts;
  const int* o = L->con.off;
  double* Qxx = kkt_rec + L->kkt.off[RTOC_KKT_QXX];
  double* lx = kkt_rec + L->kkt.off[RTOC_KKT_LX];
  double* Qff = cdd_rec + L->cdd.off[RTOC_CDD_QFF];
  double* Qqf = cdd_rec + L->cdd.off[RTOC_CDD_QQF];
  double* lf = cdd_rec + L->cdd.off[RTOC_CDD_LF];
  for (int k = 0; k < nact; ++k) {
    const double* dq = cone_rec + (size_t)k * 5 * nv;
    const double* df = cone_rec + rtoc_cone_dgdf_off(nv, max_contacts) + k * 15;
    const int r0 = row0 + 5 * k, stack = k * contact_dim;
    double cond[5], rr[5];
    for (int j = 0; j < 5; ++j) {
      const double slack = con_rec[o[RTOC_CON_SLACK] + r0 + j], dual = con_rec[o[RTOC_CON_DUAL] + r0 + j];
      cond[j] = (dual * con_rec[o[RTOC_CON_RESIDUAL] + r0 + j] - con_rec[o[RTOC_CON_CMPL] + r0 + j]) / slack;
      con_rec[o[RTOC_CON_COND] + r0 + j] = cond[j]; /* (:202) */
      rr[j] = dual / slack;                          /* (:211-212) */
    }
    for (int c = 0; c < nv; ++c) { /* lq += dg_dq^T cond (:206) */
      double acc = 0.0;
      for (int j = 0; j < 5; ++j) acc += dq[j + 5 * c] * cond[j];
      lx[c] += acc;
    }
    for (int m = 0; m < 3; ++m) { /* lf += dg_df^T cond (:207-208) */
      double acc = 0.0;
      for (int j = 0; j < 5; ++j) acc += df[j + 5 * m] * cond[j];
      lf[stack + m] += acc;
    }
    for (int c = 0; c < nv; ++c)
      for (int r = 0; r < nv; ++r) { /* Qqq += dg_dq^T (r dg_dq) (:213,:215-216) */
        double acc = 0.0;
        for (int j = 0; j < 5; ++j) acc += dq[j + 5 * r] * (rr[j] * dq[j + 5 * c]);
        Qxx[r + (size_t)c * nx] += acc;
      }
    for (int m = 0; m < 3; ++m)
      for (int r = 0; r < nv; ++r) { /* Qqf += dg_dq^T (r dg_df) (:214,:217-218) */
        double acc = 0.0;
        for (int j = 0; j < 5; ++j) acc += dq[j + 5 * r] * (rr[j] * df[j + 5 * m]);
        Qqf[r + (size_t)(stack + m) * nv] += acc;
      }
    for (int n = 0; n < 3; ++n)
      for (int m = 0; m < 3; ++m) { /* Qff += dg_df^T (r dg_df) (:219-220) */
        double acc = 0.0;
        for (int j = 0; j < 5; ++j) acc += df[j + 5 * m] * (rr[j] * df[j + 5 * n]);
        Qff[(stack + m) + (size_t)(stack + n) * nfp] += acc;
      }
  }
}

/* expandSlackAndDual (:238-268) + maxSlackStepSize / maxDualStepSize; steps min-reduced in place */
void orc_cone_expand_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                           const double* cone_rec, const double* dir_rec, double* con_rec, double tau,
                           double* steps) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nv = L->dims.nv;
  const int nact = g->dimf / contact_dim, row0 = L->dims.nc_max - 5 * max_contacts;
  const int* o = L->con.off;
  const double* dqv = dir_rec + L->dir.off[RTOC_DIR_DX];
  const double* dfv = dir_rec + L->dir.off[RTOC_DIR_DAF] + nv;
  for (int k = 0; k < nact; ++k) {
    const double* dq = cone_rec + (size_t)k * 5 * nv;
    const double* df = cone_rec + rtoc_cone_dgdf_off(nv, max_contacts) + k * 15;
    const int stack = k * contact_dim;
    for (int j = 0; j < 5; ++j) {
      const int r = row0 + 5 * k + j;
      double accq = 0.0, accf = 0.0;
      for (int c = 0; c < nv; ++c) accq += dq[j + 5 * c] * dqv[c];
      for (int m = 0; m < 3; ++m) accf += df[j + 5 * m] * dfv[stack + m];
      const double slack = con_rec[o[RTOC_CON_SLACK] + r], dual = con_rec[o[RTOC_CON_DUAL] + r];
      const double dslack = -accq - accf - con_rec[o[RTOC_CON_RESIDUAL] + r];
      const double ddual = -(dual * dslack + con_rec[o[RTOC_CON_CMPL] + r]) / slack;
      con_rec[o[RTOC_CON_DSLACK] + r] = dslack;
      con_rec[o[RTOC_CON_DDUAL] + r] = ddual;
      const double fs = -tau * (slack / dslack), fd = -tau * (dual / ddual);
      if (fs > 0 && fs < 1 && fs < steps[0]) steps[0] = fs;
      if (fd > 0 && fd < 1 && fd < steps[1]) steps[1] = fd;
    }
  }
}

void orc_cone_update_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                           double* con_rec, double primal_step, double dual_step) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nact = g->dimf / contact_dim, row0 = L->dims.nc_max - 5 * max_contacts;
  const int* o = L->con.off;
  for (int r = row0; r < row0 + 5 * nact; ++r) {
    con_rec[o[RTOC_CON_SLACK] + r] += primal_step * con_rec[o[RTOC_CON_DSLACK] + r];
    con_rec[o[RTOC_CON_DUAL] + r] += dual_step * con_rec[o[RTOC_CON_DDUAL] + r];
  }
}

/* phase 0 condense, 1 expand (steps min-reduced, NOT reset here), 2 update */
void orc_cone_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch, int max_contacts,
                    int contact_dim, const double* cone, double* kkt, double* cdd, double* con,
                    const double* dir, double tau, double* steps, int phase) {
  const size_t cs = (size_t)rtoc_cone_stride(L->dims.nv, max_contacts);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nstages - 1; ++i) {
      const size_t rec = (size_t)b * nstages + i;
      if (phase == 0)
        orc_cone_condense_stage(L, &grid[i], max_contacts, contact_dim, cone + rec * cs,
                                kkt + rec * L->kkt.stride, cdd + rec * L->cdd.stride, con + rec * L->con.stride);
      else if (phase == 1)
        orc_cone_expand_stage(L, &grid[i], max_contacts, contact_dim, cone + rec * cs,
                              dir + rec * L->dir.stride, con + rec * L->con.stride, tau, steps + 2 * b);
      else
        orc_cone_update_stage(L, &grid[i], max_contacts, contact_dim, con + rec * L->con.stride, steps[2 * b],
                              steps[2 * b + 1]);
    }
}

/* ======================================================================================
 * Contact-wrench-cone PDIPM rows (src/constraints/contact_wrench_cone.cpp): g = cone f on the 6-d
 * wrench of every active surface contact, 17 rows each.  cone record: 17 x 6 per active contact
 * (ld 17) at k*102; constraint rows row0 + 17k + j, row0 = nc_max - 17*max_contacts.
 * ====================================================================================== */
/* computeCone (:282-303), written out row by row */
void orc_wrench_cone_matrix(double X, double Y, double mu, double* out) {
  const double c = (X + Y) * mu;
  const double rows[17][6] = {
      {0, 0, -1, 0, 0, 0},        {-1, 0, -mu, 0, 0, 0},      {1, 0, -mu, 0, 0, 0},       {0, -1, -mu, 0, 0, 0},
      {0, 1, -mu, 0, 0, 0},       {0, 0, -Y, -1, 0, 0},       {0, 0, -Y, 1, 0, 0},        {0, 0, -X, 0, -1, 0},
      {0, 0, -X, 0, 1, 0},        {-Y, -X, -c, mu, mu, -1},   {-Y, X, -c, mu, -mu, -1},   {Y, -X, -c, -mu, mu, -1},
      {Y, X, -c, -mu, -mu, -1},   {Y, X, -c, mu, mu, 1},      {Y, -X, -c, mu, -mu, 1},    {-Y, X, -c, -mu, mu, 1},
      {-Y, -X, -c, -mu, -mu, 1}};
  for (int j = 0; j < 17; ++j)
    for (int m = 0; m < 6; ++m) out[j + 17 * m] = rows[j][m];
}

/* condenseSlackAndDual (:209-238) */
void orc_wrench_condense_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, const double* cone_rec,
                               double* cdd_rec, double* con_rec) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nfp = L->dims.nf_max > 0 ? L->dims.nf_max : 1;
  const int nact = g->dimf / 6, row0 = L->dims.nc_max - 17 * max_contacts;
  const int* o = L->con.off;
  double* Qff = cdd_rec + L->cdd.off[RTOC_CDD_QFF];
  double* lf = cdd_rec + L->cdd.off[RTOC_CDD_LF];
  for (int r = row0; r < row0 + 17 * max_contacts; ++r) con_rec[o[RTOC_CON_COND] + r] = 0.0; /* (:213) */
  for (int k = 0; k < nact; ++k) {
    const double* J = cone_rec + (size_t)k * 102;
    const int r0 = row0 + 17 * k, stack = 6 * k;
    double cond[17], rr[17];
    for (int j = 0; j < 17; ++j) {
      const double slack = con_rec[o[RTOC_CON_SLACK] + r0 + j], dual = con_rec[o[RTOC_CON_DUAL] + r0 + j];
      rr[j] = dual / slack; /* (:224-225) */
      cond[j] = (dual * con_rec[o[RTOC_CON_RESIDUAL] + r0 + j] - con_rec[o[RTOC_CON_CMPL] + r0 + j]) / slack;
      con_rec[o[RTOC_CON_COND] + r0 + j] = cond[j]; /* (:228) */
    }
    for (int n = 0; n < 6; ++n)
      for (int m = 0; m < 6; ++m) { /* Qff block += cone^T diag(r) cone (:226-227) */
        double acc = 0.0;
        for (int j = 0; j < 17; ++j) acc += J[j + 17 * m] * (rr[j] * J[j + 17 * n]);
        Qff[(stack + m) + (size_t)(stack + n) * nfp] += acc;
      }
    for (int m = 0; m < 6; ++m) { /* lf += cone^T cond (:229-230) */
      double acc = 0.0;
      for (int j = 0; j < 17; ++j) acc += J[j + 17 * m] * cond[j];
      lf[stack + m] += acc;
    }
  }
}

/* expandSlackAndDual (:241-270) + maxSlackStepSize / maxDualStepSize; steps min-reduced in place */
void orc_wrench_expand_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, const double* cone_rec,
                             const double* dir_rec, double* con_rec, double tau, double* steps) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nact = g->dimf / 6, row0 = L->dims.nc_max - 17 * max_contacts;
  if (nact == 0) return; /* the GPU path skips grid points without active contacts */
  const int* o = L->con.off;
  const double* dfv = dir_rec + L->dir.off[RTOC_DIR_DAF] + L->dims.nv;
  for (int r = row0; r < row0 + 17 * max_contacts; ++r) { /* (:247-248) */
    con_rec[o[RTOC_CON_DSLACK] + r] = 1.0;
    con_rec[o[RTOC_CON_DDUAL] + r] = 1.0;
  }
  for (int k = 0; k < nact; ++k) {
    const double* J = cone_rec + (size_t)k * 102;
    for (int j = 0; j < 17; ++j) {
      const int r = row0 + 17 * k + j;
      double acc = 0.0;
      for (int m = 0; m < 6; ++m) acc += J[j + 17 * m] * dfv[6 * k + m];
      const double slack = con_rec[o[RTOC_CON_SLACK] + r], dual = con_rec[o[RTOC_CON_DUAL] + r];
      const double dslack = -acc - con_rec[o[RTOC_CON_RESIDUAL] + r]; /* (:260-262) */
      const double ddual = -(dual * dslack + con_rec[o[RTOC_CON_CMPL] + r]) / slack;
      con_rec[o[RTOC_CON_DSLACK] + r] = dslack;
      con_rec[o[RTOC_CON_DDUAL] + r] = ddual;
      const double fs = -tau * (slack / dslack), fd = -tau * (dual / ddual);
      if (fs > 0 && fs < 1 && fs < steps[0]) steps[0] = fs;
      if (fd > 0 && fd < 1 && fd < steps[1]) steps[1] = fd;
    }
  }
}

void orc_wrench_update_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, double* con_rec,
                             double primal_step, double dual_step) {
  if (g->type == RTOC_GRID_TERMINAL) return;
  const int nact = g->dimf / 6, row0 = L->dims.nc_max - 17 * max_contacts;
  const int* o = L->con.off;
  for (int r = row0; r < row0 + 17 * nact; ++r) {
    con_rec[o[RTOC_CON_SLACK] + r] += primal_step * con_rec[o[RTOC_CON_DSLACK] + r];
    con_rec[o[RTOC_CON_DUAL] + r] += dual_step * con_rec[o[RTOC_CON_DDUAL] + r];
  }
}

/* phase 0 condense, 1 expand (steps min-reduced, NOT reset here), 2 update */
void orc_wrench_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch, int max_contacts,
                      const double* cone, double* cdd, double* con, const double* dir, double tau, double* steps,
                      int phase) {
  const size_t cs = (size_t)rtoc_wrench_cone_stride(max_contacts);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nstages - 1; ++i) {
      const size_t rec = (size_t)b * nstages + i;
      if (phase == 0)
        orc_wrench_condense_stage(L, &grid[i], max_contacts, cone + rec * cs, cdd + rec * L->cdd.stride,
                                  con + rec * L->con.stride);
      else if (phase == 1)
        orc_wrench_expand_stage(L, &grid[i], max_contacts, cone + rec * cs, dir + rec * L->dir.stride,
                                con + rec * L->con.stride, tau, steps + 2 * b);
      else
        orc_wrench_update_stage(L, &grid[i], max_contacts, con + rec * L->con.stride, steps[2 * b],
                                steps[2 * b + 1]);
    }
}

/* ======================================================================================
 * KKT error of one instance (src/ocp/intermediate_stage.cpp:132, impact_stage.cpp, terminal_stage.cpp;
 * SplitKKTResidual::KKTError split_kkt_residual.hxx:90-104; ContactDynamicsData::KKTError
 * contact_dynamics_data.hpp:204-206; ConstraintComponentData::KKTError constraint_component_data.hpp:122-124;
 * OCPSolver::KKTError ocp_solver.cpp:429-431 without the STO term), on pre-condensation records.
 * cdd / con / rows may be NULL.
 * ====================================================================================== */
static double sqn(const double* p, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += p[i] * p[i];
  return s;
}

double orc_kkt_error(const rtoc_layout* L, const rtoc_grid* grid, int nstages, const double* kkt,
                     const double* cdd, const double* con, const rtoc_box_row* rows, int nrows,
                     int cone_contacts, int cone_dim, int cone_rows) {
  const int nv = L->dims.nv, nu = L->dims.nu, np = L->dims.np, nx = L->nx;
  double err = 0.0;
  for (int i = 0; i < nstages; ++i) {
    const rtoc_grid* g = &grid[i];
    const double* kr = kkt + (size_t)i * L->kkt.stride;
    err += sqn(kr + L->kkt.off[RTOC_KKT_LX], nx);
    if (g->type == RTOC_GRID_TERMINAL) continue;
    err += sqn(kr + L->kkt.off[RTOC_KKT_FX], nx);
    const int impact = g->type == RTOC_GRID_IMPACT;
    if (!impact) {
      err += sqn(kr + L->kkt.off[RTOC_KKT_LU], nu);
      if (g->dims > 0) err += sqn(kr + L->kkt.off[RTOC_KKT_PRES], g->dims);
    }
    if (cdd) {
      const double* cr = cdd + (size_t)i * L->cdd.stride;
      err += sqn(cr + L->cdd.off[RTOC_CDD_LA], nv);
      err += sqn(cr + L->cdd.off[RTOC_CDD_LF], g->dimf);
      err += sqn(cr + L->cdd.off[RTOC_CDD_IDC], nv + g->dimf);
      if (!impact) err += sqn(cr + L->cdd.off[RTOC_CDD_LUP], np);
    }
    if (con) {
      const double* nr = con + (size_t)i * L->con.stride;
      const int* o = L->con.off;
      for (int r = 0; r < nrows; ++r)
        if (row_active(&rows[r], g)) {
          const double x = nr[o[RTOC_CON_RESIDUAL] + r], y = nr[o[RTOC_CON_CMPL] + r];
          err += x * x + y * y;
        }
      if (cone_contacts > 0) {
        const int row0 = L->dims.nc_max - cone_rows * cone_contacts, n = cone_rows * (g->dimf / cone_dim);
        for (int r = row0; r < row0 + n; ++r) {
          const double x = nr[o[RTOC_CON_RESIDUAL] + r], y = nr[o[RTOC_CON_CMPL] + r];
          err += x * x + y * y;
        }
      }
    }
  }
  return sqrt(err);
}

/* ======================================================================================
 * SplitSolution::integrate (src/core/split_solution.cpp:58-90): every member advanced by step x direction;
 * q on the manifold (robot.integrateConfiguration, :62): joints additively, a free-flyer base by the SE(3)
 * exponential (orc_se3_integrate, rtoc_oracle_rbd.c: pinocchio::integrate restated, parity unpinned).
 * ====================================================================================== */
void orc_se3_integrate(const double* q7, const double* v6, double scale, double* out7);
void orc_integrate_solution_stage(const rtoc_layout* L, const rtoc_grid* g, double step, const double* dir_rec,
                                  double* sol_rec) {
  const int nv = L->dims.nv, nu = L->dims.nu, np = L->dims.np;
  const int impact = g->type == RTOC_GRID_IMPACT;
  const int* so = L->sol.off;
  const int* dof = L->dir.off;
  const double* dx = dir_rec + dof[RTOC_DIR_DX];
  const double* dl = dir_rec + dof[RTOC_DIR_DLMDGMM];
  const double* daf = dir_rec + dof[RTOC_DIR_DAF];
  const double* dbm = dir_rec + dof[RTOC_DIR_DBETAMU];
  const int nb = np == 6 ? 6 : 0;
  if (nb && step != 0.0) { /* step 0: the iterate is kept as it is */
    double q7[7];
    orc_se3_integrate(sol_rec + so[RTOC_SOL_Q], dx, step, q7);
    for (int i = 0; i < 7; ++i) sol_rec[so[RTOC_SOL_Q] + i] = q7[i];
  }
  for (int i = 0; i < nv - nb; ++i) sol_rec[so[RTOC_SOL_Q] + (nb ? 7 : 0) + i] += step * dx[nb + i]; /* (:62) */
  for (int i = 0; i < nv; ++i) sol_rec[so[RTOC_SOL_V] + i] += step * dx[nv + i];                      /* (:63) */
  for (int i = 0; i < nv; ++i) sol_rec[so[RTOC_SOL_A] + i] += step * daf[i];                          /* (:65 / :71) */
  if (!impact) {
    for (int i = 0; i < nu; ++i) sol_rec[so[RTOC_SOL_U] + i] += step * dir_rec[dof[RTOC_DIR_DU] + i]; /* (:67) */
  } else {
    for (int i = 0; i < nu; ++i) sol_rec[so[RTOC_SOL_U] + i] = 0.0;                                   /* (:72) */
  }
  for (int i = 0; i < nv; ++i) {
    sol_rec[so[RTOC_SOL_LMD] + i] += step * dl[i];       /* (:74) */
    sol_rec[so[RTOC_SOL_GMM] + i] += step * dl[nv + i];  /* (:75) */
    sol_rec[so[RTOC_SOL_BETA] + i] += step * dbm[i];     /* (:76) */
  }
  if (np == 6 && !impact)
    for (int i = 0; i < np; ++i) sol_rec[so[RTOC_SOL_NUP] + i] += step * dir_rec[dof[RTOC_DIR_DNUP] + i];
  for (int i = 0; i < g->dimf; ++i) {
    sol_rec[so[RTOC_SOL_F] + i] += step * daf[nv + i];   /* (:81) */
    sol_rec[so[RTOC_SOL_MU] + i] += step * dbm[nv + i];  /* (:83) */
  }
  if (!impact)
    for (int i = 0; i < g->dims; ++i) sol_rec[so[RTOC_SOL_XI] + i] += step * dir_rec[dof[RTOC_DIR_DXI] + i];
}

void orc_integrate_solution_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                                  const double* steps, const double* dir, double* sol) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < nstages; ++i) {
      const size_t rec = (size_t)b * nstages + i;
      orc_integrate_solution_stage(L, &grid[i], steps[2 * b], dir + rec * L->dir.stride, sol + rec * L->sol.stride);
    }
}

/* ======================================================================================
 * SwitchingTimeOptimization::evalKKT, the part downstream of the STO cost / dwell-time constraints
 * (src/sto/switching_time_optimization.cpp:105-137): scatter of the per-event gradient lt and Hessian diagonal
 * diag(Qtt_) into the stage right after an impact (:108-112) / the lift stage (:113-117), then the STO term of the
 * KKT error: squared differences of the per-phase Hamiltonian sums across events with STO enabled (:120-136).
 * GridInfo::phase = number of impact / lift grids up to and including the grid (time_discretization.cpp:70-126).
 * kkt: ONE instance's records AFTER the condensation (DirectMultipleShooting::evalKKT runs first, ocp_solver.cpp:118-119).
 * Returns the squared STO KKT error (PerformanceIndex::kkt_error of the STO problem without its constraint part).
 * ====================================================================================== */
double orc_sto_eval_kkt(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt, const double* lt,
                        const double* qtt_diag, int num_events) {
  const int N = nstages - 1;
  const int so = L->kkt.off[RTOC_KKT_SCAL];
  int event_index = 0;
  for (int i = 0; i < N && event_index < num_events; ++i) {
    if (grid[i].type == RTOC_GRID_IMPACT) {
      double* sc = kkt + (size_t)(i + 1) * L->kkt.stride + so;
      sc[RTOC_KKT_SCAL_H] -= lt[event_index];
      sc[RTOC_KKT_SCAL_QTT] += qtt_diag[event_index];
      ++event_index;
    } else if (grid[i].type == RTOC_GRID_LIFT) {
      double* sc = kkt + (size_t)i * L->kkt.stride + so;
      sc[RTOC_KKT_SCAL_H] -= lt[event_index];
      sc[RTOC_KKT_SCAL_QTT] += qtt_diag[event_index];
      ++event_index;
    }
  }
  double h[64];
  for (int p = 0; p < 64; ++p) h[p] = 0.0;
  int phase = 0;
  for (int i = 0; i < N; ++i) {
    if (grid[i].type == RTOC_GRID_IMPACT || grid[i].type == RTOC_GRID_LIFT) ++phase;
    if (phase < 64) h[phase] += kkt[(size_t)i * L->kkt.stride + so + RTOC_KKT_SCAL_H];
  }
  double err = 0.0;
  event_index = 0;
  for (int i = 0; i < N; ++i) {
    if ((grid[i].type == RTOC_GRID_IMPACT && grid[i + 1].sto) || (grid[i].type == RTOC_GRID_LIFT && grid[i].sto)) {
      const double hdiff = h[event_index] - h[event_index + 1];
      err += hdiff * hdiff;
      ++event_index;
    }
  }
  return err;
}

/*
 * rtoc_oracle.c -- CPU restatement of robotoc's Riccati / KKT-condensation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (robotoc_amd/, include/) may
 * link, import or call this file; only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it, and only as the checker / the timed
 * CPU baseline -- never as the thing shipped.
 *
 * PARITY STATUS: pinned against the reference's own SOURCES, not against a robotoc binary built with
 * Eigen + Pinocchio (both absent from the image, CMakeLists.txt:27-31; the reference ships no golden
 * vectors, its tests are differential tests on srand(time(0)) inputs).  What pins this file:
 *   (o)  oracle/_ref/librtoc_ref.so -- robotoc's src/riccati, src/dynamics, src/core .cpp files compiled
 *        where they lie (oracle/Makefile.ref) against oracle/ref_shim (an eager stand-in for the Eigen API,
 *        dimension-only stand-ins for Robot / OCP / TimeDiscretization): tests/test_oracle_vs_reference.py
 *        runs both on the same seeded inputs (1e-15 ... 1e-10 on every grid kind incl. switching constraints,
 *        STO terms, phase transitions, condensation and expansion), tests/test_golden_ref.py checks this file
 *        and the HIP path against the fixtures those sources generated (tests/golden/ref_*.npz);
 *   (i)  the closed-form expectations of the reference's own unit tests,
 *        re-stated in numpy in tests/test_oracle_reference_identities.py
 *        (test/riccati/backward_riccati_recursion_factorizer_test.cpp:31-138,
 *         test/riccati/riccati_factorizer_test.cpp:36-400,
 *         test/riccati/unconstr_riccati_recursion_test.cpp:61-106,
 *         test/dynamics/contact_dynamics_test.cpp:87-201);
 *   (ii) a dense assembly of the whole-horizon KKT system solved with LAPACK
 *        (tests/test_oracle_dense_kkt.py) -- the horizon-level check the
 *        reference lacks (test/riccati/riccati_recursion_test.cpp:56-63 is empty).
 *   Not pinned by (o): Eigen's summation order, Pinocchio's arithmetic (computeMJtJinv's elimination
 *   order), the PDIPM row classes and the evalKKT-tail scalings (closed-form tests only).
 *
 * Every function cites the reference lines it follows; operation order inside a
 * function follows the cited lines (products accumulate in k-order).
 * All matrices column-major; LQRPolicy::K row-major.  Plain C99, fp64.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/rtoc.h"

/* ------------------------------------------------------------------------- */
/* tiny dense helpers (column-major)                                          */
/* ------------------------------------------------------------------------- */

/* C(MxN) = beta*C + alpha * op(A) * op(B);  ta/tb = 0 (N) or 1 (T).
 * Loop order j,k,i keeps the innermost loop contiguous for the N case so that
 * gcc -O3 -march=native vectorises it (this file doubles as the CPU baseline). */
static void gemm(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                 const double* B, int ldb, double beta, double* C, int ldc) {
  int j0 = 0;
  if (!ta) {
    /* four columns of C per pass over A (each a[i] is loaded once for four FMAs); the summation order of
     * every element is the k order of the plain loop below, so results are bit-identical to it */
    for (; j0 + 4 <= N; j0 += 4) {
      double* c0 = C + (size_t)j0 * ldc;
      double* c1 = c0 + ldc;
      double* c2 = c1 + ldc;
      double* c3 = c2 + ldc;
      if (beta == 0.0) {
        for (int i = 0; i < M; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.0;
      } else if (beta != 1.0) {
        for (int i = 0; i < M; ++i) {
          c0[i] *= beta;
          c1[i] *= beta;
          c2[i] *= beta;
          c3[i] *= beta;
        }
      }
      for (int k = 0; k < K; ++k) {
        const double b0 = alpha * (tb ? B[j0 + (size_t)k * ldb] : B[k + (size_t)j0 * ldb]);
        const double b1 = alpha * (tb ? B[j0 + 1 + (size_t)k * ldb] : B[k + (size_t)(j0 + 1) * ldb]);
        const double b2 = alpha * (tb ? B[j0 + 2 + (size_t)k * ldb] : B[k + (size_t)(j0 + 2) * ldb]);
        const double b3 = alpha * (tb ? B[j0 + 3 + (size_t)k * ldb] : B[k + (size_t)(j0 + 3) * ldb]);
        const double* a = A + (size_t)k * lda;
#pragma omp simd
        for (int i = 0; i < M; ++i) {
          const double av = a[i];
          c0[i] += av * b0;
          c1[i] += av * b1;
          c2[i] += av * b2;
          c3[i] += av * b3;
        }
      }
    }
  }
  for (int j = j0; j < N; ++j) {
    double* c = C + (size_t)j * ldc;
    if (beta == 0.0) {
      for (int i = 0; i < M; ++i) c[i] = 0.0;
    } else if (beta != 1.0) {
      for (int i = 0; i < M; ++i) c[i] *= beta;
    }
    if (!ta) {
      for (int k = 0; k < K; ++k) {
        const double b = alpha * (tb ? B[j + (size_t)k * ldb] : B[k + (size_t)j * ldb]);
        const double* a = A + (size_t)k * lda;
#pragma omp simd
        for (int i = 0; i < M; ++i) c[i] += a[i] * b;
      }
    } else {
      int i0 = 0;
      if (!tb) {
        /* four dot products per pass over column j of B */
        const double* b = B + (size_t)j * ldb;
        for (; i0 + 4 <= M; i0 += 4) {
          const double* a0 = A + (size_t)i0 * lda;
          const double* a1 = a0 + lda;
          const double* a2 = a1 + lda;
          const double* a3 = a2 + lda;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
          for (int k = 0; k < K; ++k) {
            const double bk = b[k];
            s0 += a0[k] * bk;
            s1 += a1[k] * bk;
            s2 += a2[k] * bk;
            s3 += a3[k] * bk;
          }
          c[i0] += alpha * s0;
          c[i0 + 1] += alpha * s1;
          c[i0 + 2] += alpha * s2;
          c[i0 + 3] += alpha * s3;
        }
      }
      for (int i = i0; i < M; ++i) {
        const double* a = A + (size_t)i * lda;
        double acc = 0.0;
        if (!tb) {
          const double* b = B + (size_t)j * ldb;
#pragma omp simd reduction(+ : acc)
          for (int k = 0; k < K; ++k) acc += a[k] * b[k];
        } else {
          for (int k = 0; k < K; ++k) acc += a[k] * B[j + (size_t)k * ldb];
        }
        c[i] += alpha * acc;
      }
    }
  }
}

/* y = beta*y + alpha*op(A)*x */
static void gemv(int ta, int M, int N, double alpha, const double* A, int lda, const double* x,
                 double beta, double* y) {
  gemm(ta, 0, ta ? N : M, 1, ta ? M : N, alpha, A, lda, x, ta ? M : N, beta, y, ta ? N : M);
}

static double dot(int n, const double* a, const double* b) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* Cholesky A = L L^T, lower triangle in place (upper left untouched). Returns 0 on success.
 * Same factorisation as Eigen::LLT<MatrixXd> (riccati_factorizer.cpp:49). */
static int llt(double* A, int n, int lda) {
  int bad = 0;
  for (int j = 0; j < n; ++j) {
    double d = A[j + (size_t)j * lda];
    for (int k = 0; k < j; ++k) d -= A[j + (size_t)k * lda] * A[j + (size_t)k * lda];
    if (!(d > 0.0)) bad = 1;
    const double ljj = sqrt(d);
    A[j + (size_t)j * lda] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i + (size_t)j * lda];
      for (int k = 0; k < j; ++k) v -= A[i + (size_t)k * lda] * A[j + (size_t)k * lda];
      A[i + (size_t)j * lda] = v / ljj;
    }
  }
  return bad;
}

/* B <- (L L^T)^{-1} B, B is n x nrhs */
static void llt_solve(const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
  for (int c = 0; c < nrhs; ++c) {
    double* b = B + (size_t)c * ldb;
    for (int i = 0; i < n; ++i) {
      double v = b[i];
      for (int k = 0; k < i; ++k) v -= L[i + (size_t)k * ldl] * b[k];
      b[i] = v / L[i + (size_t)i * ldl];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = b[i];
      for (int k = i + 1; k < n; ++k) v -= L[k + (size_t)i * ldl] * b[k];
      b[i] = v / L[i + (size_t)i * ldl];
    }
  }
}

static int has_nan(const double* x, int n) {
  for (int i = 0; i < n; ++i)
    if (!isfinite(x[i])) return 1;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* record views                                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
  double *Fxx, *Fvu, *Qxx, *Qxu, *Quu, *Fx, *lx, *lu, *fx, *hx, *hu, *scal, *Phix, *Phiu, *Phit,
      *Pres;
} kkt_view;

typedef struct {
  double *P, *s, *Psi, *Phi, *psi_x, *phi_x, *psi_u, *phi_u, *scal, *K, *k, *T, *W, *M, *m, *mt,
      *mt_next, *dtsdx;
} ric_view;

typedef struct {
  double *dx, *du, *dlmdgmm, *dxi, *dts, *daf, *dbetamu, *dnup;
} dir_view;

static kkt_view kkt_at(const rtoc_layout* L, double* base, int stage) {
  double* r = base + (size_t)stage * L->kkt.stride;
  const int* o = L->kkt.off;
  kkt_view v = {r + o[RTOC_KKT_FXX], r + o[RTOC_KKT_FVU], r + o[RTOC_KKT_QXX], r + o[RTOC_KKT_QXU],
                r + o[RTOC_KKT_QUU], r + o[RTOC_KKT_FX],  r + o[RTOC_KKT_LX],  r + o[RTOC_KKT_LU],
                r + o[RTOC_KKT_FFX], r + o[RTOC_KKT_HX],  r + o[RTOC_KKT_HU],  r + o[RTOC_KKT_SCAL],
                r + o[RTOC_KKT_PHIX], r + o[RTOC_KKT_PHIU], r + o[RTOC_KKT_PHIT],
                r + o[RTOC_KKT_PRES]};
  return v;
}

static ric_view ric_rec(const rtoc_layout* L, double* r) {
  const int* o = L->ric.off;
  ric_view v = {r + o[RTOC_RIC_P],    r + o[RTOC_RIC_S],    r + o[RTOC_RIC_PSI],
                r + o[RTOC_RIC_PHI],  r + o[RTOC_RIC_PSIX], r + o[RTOC_RIC_PHIX],
                r + o[RTOC_RIC_PSIU], r + o[RTOC_RIC_PHIU], r + o[RTOC_RIC_SCAL],
                r + o[RTOC_RIC_K],    r + o[RTOC_RIC_KV],   r + o[RTOC_RIC_T],
                r + o[RTOC_RIC_W],    r + o[RTOC_RIC_M],    r + o[RTOC_RIC_MV],
                r + o[RTOC_RIC_MT],   r + o[RTOC_RIC_MTN],  r + o[RTOC_RIC_DTSDX]};
  return v;
}

static ric_view ric_at(const rtoc_layout* L, double* base, int stage) {
  return ric_rec(L, base + (size_t)stage * L->ric.stride);
}

static dir_view dir_at(const rtoc_layout* L, double* base, int stage) {
  double* r = base + (size_t)stage * L->dir.stride;
  const int* o = L->dir.off;
  dir_view v = {r + o[RTOC_DIR_DX],  r + o[RTOC_DIR_DU],  r + o[RTOC_DIR_DLMDGMM],
                r + o[RTOC_DIR_DXI], r + o[RTOC_DIR_DTS], r + o[RTOC_DIR_DAF],
                r + o[RTOC_DIR_DBETAMU], r + o[RTOC_DIR_DNUP]};
  return v;
}

/* scratch of one Riccati factorizer: AtP_, BtP_, GK_, Pf_ (brrf.cpp:10-13) and
 * SplitConstrainedRiccatiFactorization (split_constrained_riccati_factorization.hxx:10-23) */
typedef struct {
  double *AtP, *BtP, *GK, *Pf, *Lg, *Ginv, *DGinv, *S, *Ls, *SinvDGinv, *DtM, *KtDtM, *tmp;
} ric_scratch;

static ric_scratch scratch_alloc(const rtoc_layout* L) {
  const int nx = L->nx, nu = L->dims.nu, ns = L->dims.ns_max > 0 ? L->dims.ns_max : 1;
  ric_scratch w;
  w.AtP = (double*)calloc((size_t)nx * nx, sizeof(double));
  w.BtP = (double*)calloc((size_t)nu * nx + 1, sizeof(double));
  w.GK = (double*)calloc((size_t)nu * nx + 1, sizeof(double));
  w.Pf = (double*)calloc((size_t)nx, sizeof(double));
  w.Lg = (double*)calloc((size_t)nu * nu + 1, sizeof(double));
  w.Ginv = (double*)calloc((size_t)nu * nu + 1, sizeof(double));
  w.DGinv = (double*)calloc((size_t)ns * nu + 1, sizeof(double));
  w.S = (double*)calloc((size_t)ns * ns, sizeof(double));
  w.Ls = (double*)calloc((size_t)ns * ns, sizeof(double));
  w.SinvDGinv = (double*)calloc((size_t)ns * nu + 1, sizeof(double));
  w.DtM = (double*)calloc((size_t)nu * nx + 1, sizeof(double));
  w.KtDtM = (double*)calloc((size_t)nx * nx, sizeof(double));
  w.tmp = (double*)calloc((size_t)nx * (nx > ns ? nx : ns), sizeof(double));
  return w;
}

static void scratch_free(ric_scratch* w) {
  free(w->AtP); free(w->BtP); free(w->GK); free(w->Pf); free(w->Lg); free(w->Ginv);
  free(w->DGinv); free(w->S); free(w->Ls); free(w->SinvDGinv); free(w->DtM); free(w->KtDtM);
  free(w->tmp);
}

/* ------------------------------------------------------------------------- */
/* A.1  intermediate / lift stage                                             */
/* ------------------------------------------------------------------------- */

/* BackwardRiccatiRecursionFactorizer::factorizeKKTMatrix, brrf.cpp:31-45 */
static void factorize_kkt_matrix(const rtoc_layout* L, const ric_view* nx_, kkt_view* q,
                                 ric_scratch* w) {
  const int nv = L->dims.nv, nu = L->dims.nu, nx = L->nx;
  /* AtP = Fxx^T P+ ; BtP = Fvu^T P+[v,:]                              :34-35 */
  gemm(1, 0, nx, nx, nx, 1.0, q->Fxx, nx, nx_->P, nx, 0.0, w->AtP, nx);
  gemm(1, 0, nu, nx, nv, 1.0, q->Fvu, nv, nx_->P + nv, nx, 0.0, w->BtP, nu);
  /* Qxx += AtP Fxx                                                    :37   */
  gemm(0, 0, nx, nx, nx, 1.0, w->AtP, nx, q->Fxx, nx, 1.0, q->Qxx, nx);
  /* Qxu += AtP[:,v] Fvu                                               :39   */
  gemm(0, 0, nx, nu, nv, 1.0, w->AtP + (size_t)nv * nx, nx, q->Fvu, nv, 1.0, q->Qxu, nx);
  /* Quu += BtP[:,v] Fvu                                               :41   */
  gemm(0, 0, nu, nu, nv, 1.0, w->BtP + (size_t)nv * nu, nu, q->Fvu, nv, 1.0, q->Quu, nu);
  /* lu += BtP Fx ; lu -= Fvu^T s+[v]                                  :43-44 */
  gemv(0, nu, nx, 1.0, w->BtP, nu, q->Fx, 1.0, q->lu);
  gemv(1, nv, nu, -1.0, q->Fvu, nv, nx_->s + nv, 1.0, q->lu);
}

/* impact form, brrf.cpp:69-75 */
static void factorize_kkt_matrix_impact(const rtoc_layout* L, const ric_view* nx_, kkt_view* q,
                                        ric_scratch* w) {
  const int nx = L->nx;
  gemm(1, 0, nx, nx, nx, 1.0, q->Fxx, nx, nx_->P, nx, 0.0, w->AtP, nx);
  gemm(0, 0, nx, nx, nx, 1.0, w->AtP, nx, q->Fxx, nx, 1.0, q->Qxx, nx);
}

/* factorizeRiccatiFactorization, brrf.cpp:78-91 (with_policy=1) / :146-157 (impact, 0) */
static void factorize_riccati(const rtoc_layout* L, const ric_view* nx_, kkt_view* q,
                              const ric_view* r, ric_scratch* w, int with_policy) {
  const int nu = L->dims.nu, nx = L->nx;
  if (with_policy) {
    /* GK = Quu K ; Qxx -= K^T GK   (K row-major nu x nx == column-major nx x nu "K^T") */
    /* GK(i,j) = sum_l Quu(i,l) K(l,j) = sum_l Quu(i,l) Kt(j,l)                          */
    gemm(0, 1, nu, nx, nu, 1.0, q->Quu, nu, r->K, nx, 0.0, w->GK, nu);
    gemm(0, 0, nx, nx, nu, -1.0, r->K, nx, w->GK, nu, 1.0, q->Qxx, nx);
  }
  /* P = 0.5 (Qxx + Qxx^T)                                             :85 / :151 */
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nx; ++i)
      r->P[i + (size_t)j * nx] = 0.5 * (q->Qxx[i + (size_t)j * nx] + q->Qxx[j + (size_t)i * nx]);
  /* s = Fxx^T s+ - AtP Fx - lx (- Qxu k)                              :87-90 / :153-155 */
  gemv(1, nx, nx, 1.0, q->Fxx, nx, nx_->s, 0.0, r->s);
  gemv(0, nx, nx, -1.0, w->AtP, nx, q->Fx, 1.0, r->s);
  for (int i = 0; i < nx; ++i) r->s[i] -= q->lx[i];
  if (with_policy) gemv(0, nx, nu, -1.0, q->Qxu, nx, r->k, 1.0, r->s);
}

/* factorizeHamiltonian, brrf.cpp:48-66 */
static void factorize_hamiltonian(const rtoc_layout* L, const ric_view* nx_, const kkt_view* q,
                                  ric_view* r, ric_scratch* w, int sto_next) {
  const int nv = L->dims.nv, nu = L->dims.nu, nx = L->nx;
  gemv(0, nx, nx, 1.0, w->AtP, nx, q->fx, 0.0, r->psi_x);
  gemv(0, nu, nx, 1.0, w->BtP, nu, q->fx, 0.0, r->psi_u);
  for (int i = 0; i < nx; ++i) r->psi_x[i] += q->hx[i];
  for (int i = 0; i < nu; ++i) r->psi_u[i] += q->hu[i];
  gemv(1, nx, nx, 1.0, q->Fxx, nx, nx_->Psi, 1.0, r->psi_x);
  gemv(1, nv, nu, 1.0, q->Fvu, nv, nx_->Psi + nv, 1.0, r->psi_u);
  if (sto_next) {
    gemv(1, nx, nx, 1.0, q->Fxx, nx, nx_->Phi, 0.0, r->phi_x);
    gemv(1, nv, nu, 1.0, q->Fvu, nv, nx_->Phi + nv, 0.0, r->phi_u);
  } else {
    memset(r->phi_x, 0, sizeof(double) * nx);
    memset(r->phi_u, 0, sizeof(double) * nu);
  }
}

/* factorizeSTOFactorization, brrf.cpp:94-143 */
static void factorize_sto(const rtoc_layout* L, const ric_view* nx_, const kkt_view* q,
                          ric_view* r, ric_scratch* w, int sto_next) {
  const int nu = L->dims.nu, nx = L->nx;
  double* sc = r->scal;
  const double* scn = nx_->scal;
  /* Psi = psi_x + K^T psi_u ; Phi likewise                             :100-108 */
  memcpy(r->Psi, r->psi_x, sizeof(double) * nx);
  gemv(0, nx, nu, 1.0, r->K, nx, r->psi_u, 1.0, r->Psi);
  if (sto_next) {
    memcpy(r->Phi, r->phi_x, sizeof(double) * nx);
    gemv(0, nx, nu, 1.0, r->K, nx, r->phi_u, 1.0, r->Phi);
  } else {
    memset(r->Phi, 0, sizeof(double) * nx);
  }
  /* xi                                                                 :110-115 */
  gemv(0, nx, nx, 1.0, nx_->P, nx, q->fx, 0.0, w->Pf);
  double xi = dot(nx, q->fx, w->Pf);
  xi += q->scal[RTOC_KKT_SCAL_QTT];
  xi += 2 * dot(nx, nx_->Psi, q->fx);
  xi += dot(nu, r->T, r->psi_u);
  xi += scn[RTOC_RIC_SCAL_XI];
  sc[RTOC_RIC_SCAL_XI] = xi;
  if (sto_next) {
    double chi = q->scal[RTOC_KKT_SCAL_QTT_PREV];
    chi += dot(nx, nx_->Phi, q->fx);
    chi += dot(nu, r->T, r->phi_u);
    chi += scn[RTOC_RIC_SCAL_CHI];
    sc[RTOC_RIC_SCAL_CHI] = chi;
    double rho = dot(nu, r->W, r->phi_u);
    rho += scn[RTOC_RIC_SCAL_RHO];
    sc[RTOC_RIC_SCAL_RHO] = rho;
  } else {
    sc[RTOC_RIC_SCAL_CHI] = 0.0;
    sc[RTOC_RIC_SCAL_RHO] = 0.0;
  }
  /* eta                                                                :129-134 */
  gemv(0, nx, nx, 1.0, nx_->P, nx, q->Fx, 0.0, w->Pf);
  for (int i = 0; i < nx; ++i) w->Pf[i] -= nx_->s[i];
  double eta = dot(nx, q->fx, w->Pf);
  eta += q->scal[RTOC_KKT_SCAL_H];
  eta += dot(nx, nx_->Psi, q->Fx);
  eta += dot(nu, r->psi_u, r->k);
  eta += scn[RTOC_RIC_SCAL_ETA];
  sc[RTOC_RIC_SCAL_ETA] = eta;
  if (sto_next) {
    double iota = dot(nx, nx_->Phi, q->Fx);
    iota += dot(nu, r->phi_u, r->k);
    iota += scn[RTOC_RIC_SCAL_IOTA];
    sc[RTOC_RIC_SCAL_IOTA] = iota;
  } else {
    sc[RTOC_RIC_SCAL_IOTA] = 0.0;
  }
}

/* RiccatiFactorizer::backwardRiccatiRecursion (7-arg), riccati_factorizer.cpp:44-142.
 * ns = kkt_matrix.dims().  Returns RTOC_STAT_* bits. */
static unsigned backward_stage(const rtoc_layout* L, const ric_view* nx_, kkt_view* q, ric_view* r,
                               ric_scratch* w, int ns, int sto, int sto_next) {
  const int nu = L->dims.nu, nx = L->nx, ldn = L->dims.ns_max;
  unsigned stat = 0;
  factorize_kkt_matrix(L, nx_, q, w);
  /* llt_.compute(Quu)                                                   :49 */
  memcpy(w->Lg, q->Quu, sizeof(double) * nu * nu);
  if (llt(w->Lg, nu, nu)) stat |= RTOC_STAT_QUU_NOT_SPD;
  if (ns == 0) {
    /* K = -llt.solve(Qxu^T) ; k = -llt.solve(lu)                        :55-56 */
    /* row-major K (nu x nx) is stored as column-major Kt (nx x nu): work on tmp = Qxu^T */
    double* X = w->tmp; /* nu x nx, ld nu */
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nu; ++i) X[i + (size_t)j * nu] = q->Qxu[j + (size_t)i * nx];
    llt_solve(w->Lg, nu, nu, X, nx, nu);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nu; ++i) r->K[(size_t)i * nx + j] = -X[i + (size_t)j * nu];
    memcpy(r->k, q->lu, sizeof(double) * nu);
    llt_solve(w->Lg, nu, nu, r->k, 1, nu);
    for (int i = 0; i < nu; ++i) r->k[i] = -r->k[i];
  } else {
    /* Schur complement                                                 :58-77 */
    double* Phix = q->Phix; /* ns x nx, ld ldn */
    double* Phiu = q->Phiu; /* ns x nu, ld ldn */
    /* Ginv = llt.solve(I) */
    memset(w->Ginv, 0, sizeof(double) * nu * nu);
    for (int i = 0; i < nu; ++i) w->Ginv[i + (size_t)i * nu] = 1.0;
    llt_solve(w->Lg, nu, nu, w->Ginv, nu, nu);
    /* DGinv^T = llt.solve(Phiu^T)  -> DGinv (ns x nu, ld ns) */
    double* Xt = w->tmp; /* nu x ns */
    for (int j = 0; j < ns; ++j)
      for (int i = 0; i < nu; ++i) Xt[i + (size_t)j * nu] = Phiu[j + (size_t)i * ldn];
    llt_solve(w->Lg, nu, nu, Xt, ns, nu);
    for (int j = 0; j < nu; ++j)
      for (int i = 0; i < ns; ++i) w->DGinv[i + (size_t)j * ns] = Xt[j + (size_t)i * nu];
    /* S = DGinv Phiu^T */
    gemm(0, 1, ns, ns, nu, 1.0, w->DGinv, ns, Phiu, ldn, 0.0, w->S, ns);
    memcpy(w->Ls, w->S, sizeof(double) * ns * ns);
    if (llt(w->Ls, ns, ns)) stat |= RTOC_STAT_S_NOT_SPD;
    /* SinvDGinv = llt_s.solve(DGinv) */
    memcpy(w->SinvDGinv, w->DGinv, sizeof(double) * ns * nu);
    llt_solve(w->Ls, ns, ns, w->SinvDGinv, nu, ns);
    /* Ginv -= SinvDGinv^T DGinv */
    gemm(1, 0, nu, nu, ns, -1.0, w->SinvDGinv, ns, w->DGinv, ns, 1.0, w->Ginv, nu);
    /* K = -Ginv Qxu^T - SinvDGinv^T Phix    (stored row-major => Kt = -Qxu Ginv^T - Phix^T SinvDGinv) */
    gemm(0, 1, nx, nu, nu, -1.0, q->Qxu, nx, w->Ginv, nu, 0.0, r->K, nx);
    gemm(1, 0, nx, nu, ns, -1.0, Phix, ldn, w->SinvDGinv, ns, 1.0, r->K, nx);
    /* k = -Ginv lu - SinvDGinv^T P */
    gemv(0, nu, nu, -1.0, w->Ginv, nu, q->lu, 0.0, r->k);
    gemv(1, ns, nu, -1.0, w->SinvDGinv, ns, q->Pres, 1.0, r->k);
    /* M = llt_s.solve(Phix) - SinvDGinv Qxu^T                               */
    double* Mt = w->tmp; /* ns x nx contiguous ld ns */
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < ns; ++i) Mt[i + (size_t)j * ns] = Phix[i + (size_t)j * ldn];
    llt_solve(w->Ls, ns, ns, Mt, nx, ns);
    gemm(0, 1, ns, nx, nu, -1.0, w->SinvDGinv, ns, q->Qxu, nx, 1.0, Mt, ns);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < ns; ++i) r->M[i + (size_t)j * ldn] = Mt[i + (size_t)j * ns];
    /* m = llt_s.solve(P) - SinvDGinv lu */
    memcpy(r->m, q->Pres, sizeof(double) * ns);
    llt_solve(w->Ls, ns, ns, r->m, 1, ns);
    gemv(0, ns, nu, -1.0, w->SinvDGinv, ns, q->lu, 1.0, r->m);
    if (has_nan(r->m, ns)) stat |= RTOC_STAT_NAN;
    for (int j = 0; j < nx; ++j)
      if (has_nan(r->M + (size_t)j * ldn, ns)) stat |= RTOC_STAT_NAN;
  }
  if (has_nan(r->K, nu * nx) || has_nan(r->k, nu)) stat |= RTOC_STAT_NAN;
  factorize_riccati(L, nx_, q, r, w, 1);
  if (ns > 0) {
    /* DtM = Phiu^T M ; KtDtM = K^T DtM ; P -= KtDtM + KtDtM^T ; s -= Phix^T m   :83-89 */
    gemm(1, 0, nu, nx, ns, 1.0, q->Phiu, ldn, r->M, ldn, 0.0, w->DtM, nu);
    gemm(0, 0, nx, nx, nu, 1.0, r->K, nx, w->DtM, nu, 0.0, w->KtDtM, nx);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nx; ++i)
        r->P[i + (size_t)j * nx] -= w->KtDtM[i + (size_t)j * nx] + w->KtDtM[j + (size_t)i * nx];
    gemv(1, ns, nx, -1.0, q->Phix, ldn, r->m, 1.0, r->s);
  }
  if (!sto) {
    /* :99-105 */
    memset(r->Psi, 0, sizeof(double) * nx);
    r->scal[RTOC_RIC_SCAL_XI] = 0.0;
    r->scal[RTOC_RIC_SCAL_CHI] = 0.0;
    r->scal[RTOC_RIC_SCAL_ETA] = 0.0;
    return stat;
  }
  factorize_hamiltonian(L, nx_, q, r, w, sto_next);
  memset(r->W, 0, sizeof(double) * nu); /* :109 */
  if (ns > 0) {
    /* :110-123 */
    gemv(0, nu, nu, -1.0, w->Ginv, nu, r->psi_u, 0.0, r->T);
    gemv(1, ns, nu, -1.0, w->SinvDGinv, ns, q->Phit, 1.0, r->T);
    if (sto_next) gemv(0, nu, nu, -1.0, w->Ginv, nu, r->phi_u, 0.0, r->W);
    memcpy(r->mt, q->Phit, sizeof(double) * ns);
    llt_solve(w->Ls, ns, ns, r->mt, 1, ns);
    gemv(0, ns, nu, -1.0, w->SinvDGinv, ns, r->psi_u, 1.0, r->mt);
    if (sto_next)
      gemv(0, ns, nu, -1.0, w->SinvDGinv, ns, r->phi_u, 0.0, r->mt_next);
    else
      memset(r->mt_next, 0, sizeof(double) * ns);
  } else {
    /* :125-130 */
    memcpy(r->T, r->psi_u, sizeof(double) * nu);
    llt_solve(w->Lg, nu, nu, r->T, 1, nu);
    for (int i = 0; i < nu; ++i) r->T[i] = -r->T[i];
    if (sto_next) {
      memcpy(r->W, r->phi_u, sizeof(double) * nu);
      llt_solve(w->Lg, nu, nu, r->W, 1, nu);
      for (int i = 0; i < nu; ++i) r->W[i] = -r->W[i];
    }
  }
  factorize_sto(L, nx_, q, r, w, sto_next);
  if (ns == 0) return stat;
  /* :136-141 */
  gemv(1, ns, nx, 1.0, r->M, ldn, q->Phit, 1.0, r->Psi);
  r->scal[RTOC_RIC_SCAL_XI] += dot(ns, r->mt, q->Phit);
  if (sto_next) r->scal[RTOC_RIC_SCAL_CHI] += dot(ns, r->mt_next, q->Phit);
  r->scal[RTOC_RIC_SCAL_ETA] += dot(ns, r->m, q->Phit);
  return stat;
}

/* A.2 impact stage: riccati_factorizer.cpp:178-197, brrf.cpp:146-174 */
static void backward_impact_stage(const rtoc_layout* L, const ric_view* nx_, kkt_view* q,
                                  ric_view* r, ric_scratch* w, int sto) {
  const int nx = L->nx;
  factorize_kkt_matrix_impact(L, nx_, q, w);
  factorize_riccati(L, nx_, q, r, w, 0);
  if (sto) {
    memset(r->Psi, 0, sizeof(double) * nx);
    gemv(1, nx, nx, 1.0, q->Fxx, nx, nx_->Phi, 0.0, r->Phi);
    r->scal[RTOC_RIC_SCAL_XI] = 0.0;
    r->scal[RTOC_RIC_SCAL_CHI] = 0.0;
    r->scal[RTOC_RIC_SCAL_RHO] = nx_->scal[RTOC_RIC_SCAL_RHO];
    r->scal[RTOC_RIC_SCAL_ETA] = 0.0;
    r->scal[RTOC_RIC_SCAL_IOTA] =
        nx_->scal[RTOC_RIC_SCAL_IOTA] + dot(nx, nx_->Phi, q->Fx);
  }
}

/* A.3 phase transition: riccati_factorizer.cpp:145-175.
 * STOPolicy {dtsdx, dtsdts, dts0} is written into the record `pol`. */
static void phase_transition(const rtoc_layout* L, const ric_view* r, ric_view* m, ric_view* pol,
                             int sto_next, double max_dts0) {
  const int nx = L->nx;
  const double* sc = r->scal;
  memcpy(m->P, r->P, sizeof(double) * nx * nx);
  memcpy(m->s, r->s, sizeof(double) * nx);
  memset(m->Psi, 0, sizeof(double) * nx);
  memcpy(m->Phi, r->Psi, sizeof(double) * nx);
  m->scal[RTOC_RIC_SCAL_XI] = 0.0;
  m->scal[RTOC_RIC_SCAL_CHI] = 0.0;
  m->scal[RTOC_RIC_SCAL_RHO] = sc[RTOC_RIC_SCAL_XI];
  m->scal[RTOC_RIC_SCAL_ETA] = 0.0;
  m->scal[RTOC_RIC_SCAL_IOTA] = sc[RTOC_RIC_SCAL_ETA];
  if (sto_next) {
    const double xi = sc[RTOC_RIC_SCAL_XI], chi = sc[RTOC_RIC_SCAL_CHI], rho = sc[RTOC_RIC_SCAL_RHO];
    const double eta = sc[RTOC_RIC_SCAL_ETA], iota = sc[RTOC_RIC_SCAL_IOTA];
    const double eps = sqrt(DBL_EPSILON);
    double sgm = xi - 2.0 * chi + rho;
    if ((sgm * max_dts0) < fabs(eta - iota) || sgm < eps) {
      sgm = fabs(sgm) + fabs(eta - iota) / max_dts0;
    }
    const double isg = 1.0 / sgm;
    for (int i = 0; i < nx; ++i) {
      const double d = r->Psi[i] - r->Phi[i];
      pol->dtsdx[i] = -isg * d;
      m->s[i] += isg * d * (eta - iota);
      m->Phi[i] -= isg * d * (xi - chi);
    }
    pol->scal[RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
    pol->scal[RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
    m->scal[RTOC_RIC_SCAL_RHO] = xi - isg * (xi - chi) * (xi - chi);
    m->scal[RTOC_RIC_SCAL_IOTA] = eta - isg * (xi - chi) * (eta - iota);
  }
}

/* ------------------------------------------------------------------------- */
/* A.4  RiccatiRecursion::backwardRiccatiRecursion, riccati_recursion.cpp:32-80 */
/* ------------------------------------------------------------------------- */
/* kkt / ric point at ONE instance (nstages records).  N = nstages-1 (terminal index). */
unsigned orc_riccati_backward(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt,
                              double* ric, double max_dts0) {
  const int N = nstages - 1, nx = L->nx;
  unsigned stat = 0;
  ric_scratch w = scratch_alloc(L);
  double* mrec = (double*)calloc((size_t)L->ric.stride, sizeof(double));
  ric_view m = ric_rec(L, mrec);
  {
    kkt_view q = kkt_at(L, kkt, N);
    ric_view r = ric_at(L, ric, N);
    memcpy(r.P, q.Qxx, sizeof(double) * nx * nx); /* :37 */
    for (int i = 0; i < nx; ++i) r.s[i] = -q.lx[i]; /* :38 */
  }
  for (int i = N - 1; i >= 0; --i) {
    const rtoc_grid* g = &grid[i];
    kkt_view q = kkt_at(L, kkt, i);
    ric_view r = ric_at(L, ric, i);
    ric_view rn = ric_at(L, ric, i + 1);
    if (g->type == RTOC_GRID_IMPACT) {
      if ((i > 0 && grid[i - 1].sto) || g->sto) {
        ric_view pol = ric_at(L, ric, i); /* sto_policy_[i] */
        phase_transition(L, &rn, &m, &pol, g->sto_next, max_dts0);
        backward_impact_stage(L, &m, &q, &r, &w, g->sto);
      } else {
        backward_impact_stage(L, &rn, &q, &r, &w, g->sto);
      }
    } else if (grid[i + 1].type == RTOC_GRID_LIFT) {
      if (g->sto || g->sto_next) {
        ric_view pol = ric_at(L, ric, i + 1); /* sto_policy_[i+1] */
        phase_transition(L, &rn, &m, &pol, g->sto_next, max_dts0);
        stat |= backward_stage(L, &m, &q, &r, &w, g->dims, g->sto, g->sto_next);
      } else {
        stat |= backward_stage(L, &rn, &q, &r, &w, g->dims, g->sto, g->sto_next);
      }
    } else {
      stat |= backward_stage(L, &rn, &q, &r, &w, g->dims, g->sto, g->sto_next);
    }
  }
  if (grid[0].sto) {
    ric_view r0 = ric_at(L, ric, 0);
    ric_view pol = ric_at(L, ric, 0); /* sto_policy_[0] */
    phase_transition(L, &r0, &m, &pol, grid[0].sto_next, max_dts0);
  }
  free(mrec);
  scratch_free(&w);
  return stat;
}

/* ------------------------------------------------------------------------- */
/* A.5  forward, riccati_recursion.cpp:83-131 + riccati_factorizer.cpp:200-277  */
/* ------------------------------------------------------------------------- */
static void fwd_policy_step(const rtoc_layout* L, const kkt_view* q, const ric_view* r, dir_view* d,
                            dir_view* dn, int sto, int sto_next) {
  const int nv = L->dims.nv, nu = L->dims.nu, nx = L->nx;
  const double dts = d->dts[0], dtsn = d->dts[1];
  /* du = K dx + k (+ T (dts_next-dts) - W dts_next)                     :205-212 */
  gemv(1, nx, nu, 1.0, r->K, nx, d->dx, 0.0, d->du);
  for (int i = 0; i < nu; ++i) d->du[i] += r->k[i];
  if (sto) {
    for (int i = 0; i < nu; ++i) d->du[i] += r->T[i] * (dtsn - dts);
    if (sto_next)
      for (int i = 0; i < nu; ++i) d->du[i] -= r->W[i] * dtsn;
  }
  /* dx+ = Fx + Fxx dx ; dv+ += Fvu du (+ fx (dts_next-dts))             :213-218 */
  memcpy(dn->dx, q->Fx, sizeof(double) * nx);
  gemv(0, nx, nx, 1.0, q->Fxx, nx, d->dx, 1.0, dn->dx);
  gemv(0, nv, nu, 1.0, q->Fvu, nv, d->du, 1.0, dn->dx + nv);
  if (sto)
    for (int i = 0; i < nx; ++i) dn->dx[i] += q->fx[i] * (dtsn - dts);
  dn->dts[0] = dts;
  dn->dts[1] = dtsn;
}

static void fwd_impact_step(const rtoc_layout* L, const kkt_view* q, dir_view* d, dir_view* dn) {
  const int nx = L->nx;
  memcpy(dn->dx, q->Fx, sizeof(double) * nx);
  gemv(0, nx, nx, 1.0, q->Fxx, nx, d->dx, 1.0, dn->dx);
  dn->dts[0] = d->dts[0];
  dn->dts[1] = d->dts[1];
}

/* computeSwitchingTimeDirection :234-240 */
static void sto_direction(const rtoc_layout* L, const ric_view* pol, dir_view* d, int sto_prev) {
  d->dts[1] = dot(L->nx, pol->dtsdx, d->dx) + pol->scal[RTOC_RIC_SCAL_DTS0];
  if (sto_prev) d->dts[1] += pol->scal[RTOC_RIC_SCAL_DTSDTS] * d->dts[0];
}

/* computeCostateDirection 4-arg :243-253 ; 3-arg (impact) :256-262 */
static void costate(const rtoc_layout* L, const ric_view* r, dir_view* d, int sto, int sto_next,
                    int impact_form) {
  const int nx = L->nx;
  gemv(0, nx, nx, 1.0, r->P, nx, d->dx, 0.0, d->dlmdgmm);
  for (int i = 0; i < nx; ++i) d->dlmdgmm[i] -= r->s[i];
  if (impact_form) {
    if (sto)
      for (int i = 0; i < nx; ++i) d->dlmdgmm[i] -= r->Phi[i] * d->dts[1];
    return;
  }
  if (sto) {
    for (int i = 0; i < nx; ++i) d->dlmdgmm[i] += r->Psi[i] * (d->dts[1] - d->dts[0]);
    if (sto_next)
      for (int i = 0; i < nx; ++i) d->dlmdgmm[i] -= r->Phi[i] * d->dts[1];
  }
}

/* computeLagrangeMultiplierDirection :265-277 */
static void lagrange(const rtoc_layout* L, const ric_view* r, dir_view* d, int ns, int sto,
                     int sto_next) {
  const int nx = L->nx, ldn = L->dims.ns_max;
  gemv(0, ns, nx, 1.0, r->M, ldn, d->dx, 0.0, d->dxi);
  for (int i = 0; i < ns; ++i) d->dxi[i] += r->m[i];
  if (sto) {
    for (int i = 0; i < ns; ++i) d->dxi[i] += r->mt[i] * (d->dts[1] - d->dts[0]);
    if (sto_next)
      for (int i = 0; i < ns; ++i) d->dxi[i] -= r->mt_next[i] * d->dts[1];
  }
}

/* d[0].dx must be set by the caller (computeInitialStateDirection). */
void orc_riccati_forward(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt,
                         double* ric, double* dir) {
  const int N = nstages - 1;
  dir_view d0 = dir_at(L, dir, 0);
  d0.dts[0] = 0.0;
  d0.dts[1] = 0.0;
  if (grid[0].sto) {
    ric_view pol = ric_at(L, ric, 0);
    sto_direction(L, &pol, &d0, 0);
  }
  for (int i = 0; i < N; ++i) {
    const rtoc_grid* g = &grid[i];
    kkt_view q = kkt_at(L, kkt, i);
    ric_view r = ric_at(L, ric, i);
    dir_view d = dir_at(L, dir, i);
    dir_view dn = dir_at(L, dir, i + 1);
    if (g->type == RTOC_GRID_IMPACT) {
      dir_view dp = dir_at(L, dir, i - 1);
      d.dts[0] = dp.dts[1];
      d.dts[1] = 0.0;
      fwd_impact_step(L, &q, &d, &dn);
      if (g->sto_next) {
        dn.dts[0] = dp.dts[1];
        dn.dts[1] = 0.0;
        ric_view pol = ric_at(L, ric, i);
        sto_direction(L, &pol, &dn, g->sto);
        d.dts[0] = dn.dts[0];
        d.dts[1] = dn.dts[1];
      }
      costate(L, &r, &d, g->sto, 0, 1);
    } else if (g->type == RTOC_GRID_LIFT) {
      dir_view dp = dir_at(L, dir, i - 1);
      d.dts[0] = dp.dts[1];
      d.dts[1] = 0.0;
      if (g->sto_next) {
        ric_view pol = ric_at(L, ric, i);
        sto_direction(L, &pol, &d, g->sto);
      }
      fwd_policy_step(L, &q, &r, &d, &dn, g->sto, g->sto_next);
      costate(L, &r, &d, g->sto, g->sto_next, 0);
    } else {
      fwd_policy_step(L, &q, &r, &d, &dn, g->sto, g->sto_next);
      costate(L, &r, &d, g->sto, g->sto_next, 0);
    }
    if (g->switching_constraint) lagrange(L, &r, &d, g->dims, g->sto, g->sto_next);
  }
  {
    ric_view r = ric_at(L, ric, N);
    dir_view d = dir_at(L, dir, N);
    costate(L, &r, &d, 0, 0, 0);
  }
}

/* ------------------------------------------------------------------------- */
/* A.6  unconstrained (fixed base, no contacts)                                */
/* unconstr_backward_riccati_recursion_factorizer.cpp:27-70,                   */
/* unconstr_riccati_factorizer.cpp:26-59, unconstr_riccati_recursion.cpp:26-48 */
/* The KKT record is reused: Quu slot holds Qaa, lu slot holds la (nu == nv).   */
/* ------------------------------------------------------------------------- */
unsigned orc_unconstr_backward(const rtoc_layout* L, int nstages, double dt, double* kkt,
                               double* ric) {
  const int N = nstages - 1, nv = L->dims.nv, nx = L->nx;
  unsigned stat = 0;
  double* Lg = (double*)calloc((size_t)nv * nv, sizeof(double));
  double* X = (double*)calloc((size_t)nv * nx, sizeof(double));
  double* GK = (double*)calloc((size_t)nv * nx, sizeof(double));
  double* PF = (double*)calloc((size_t)nx, sizeof(double));
  {
    kkt_view q = kkt_at(L, kkt, N);
    ric_view r = ric_at(L, ric, N);
    memcpy(r.P, q.Qxx, sizeof(double) * nx * nx);
    for (int i = 0; i < nx; ++i) r.s[i] = -q.lx[i];
  }
  for (int st = N - 1; st >= 0; --st) {
    kkt_view q = kkt_at(L, kkt, st);
    ric_view r = ric_at(L, ric, st);
    ric_view rn = ric_at(L, ric, st + 1);
    const double* Pn = rn.P;
    double* Qaa = q.Quu;
    double* la = q.lu;
#define PN(i, j) Pn[(i) + (size_t)(j)*nx]
#define QXX(i, j) q.Qxx[(i) + (size_t)(j)*nx]
#define QXU(i, j) q.Qxu[(i) + (size_t)(j)*nx]
    /* factorizeKKTMatrix :27-50 */
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nx; ++i) QXX(i, j) += PN(i, j);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nv; ++i) QXX(nv + i, j) += dt * PN(i, j);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nx; ++i) QXX(i, nv + j) += dt * PN(i, j);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nv; ++i) QXX(nv + i, nv + j) += (dt * dt) * PN(i, j);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nx; ++i) QXU(i, j) += dt * PN(i, nv + j);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nv; ++i) QXU(nv + i, j) += (dt * dt) * PN(i, nv + j);
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < nv; ++i) Qaa[i + (size_t)j * nv] += (dt * dt) * PN(nv + i, nv + j);
    gemv(0, nv, nx, dt, Pn + nv, nx, q.Fx, 1.0, la);
    for (int i = 0; i < nv; ++i) la[i] -= dt * rn.s[nv + i];
    /* LLT, K, k: unconstr_riccati_factorizer.cpp:30-34 */
    memcpy(Lg, Qaa, sizeof(double) * nv * nv);
    if (llt(Lg, nv, nv)) stat |= RTOC_STAT_QUU_NOT_SPD;
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nv; ++i) X[i + (size_t)j * nv] = QXU(j, i);
    llt_solve(Lg, nv, nv, X, nx, nv);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nv; ++i) r.K[(size_t)i * nx + j] = -X[i + (size_t)j * nv];
    memcpy(r.k, la, sizeof(double) * nv);
    llt_solve(Lg, nv, nv, r.k, 1, nv);
    for (int i = 0; i < nv; ++i) r.k[i] = -r.k[i];
    if (has_nan(r.K, nv * nx) || has_nan(r.k, nv)) stat |= RTOC_STAT_NAN;
    /* factorizeRiccatiFactorization :53-70 */
    gemm(0, 1, nv, nx, nv, 1.0, Qaa, nv, r.K, nx, 0.0, GK, nv);
    gemm(0, 0, nx, nx, nv, -1.0, r.K, nx, GK, nv, 1.0, q.Qxx, nx);
    for (int j = 0; j < nx; ++j)
      for (int i = 0; i < nx; ++i) r.P[i + (size_t)j * nx] = 0.5 * (QXX(i, j) + QXX(j, i));
    memcpy(r.s, rn.s, sizeof(double) * nx);
    for (int i = 0; i < nv; ++i) r.s[nv + i] += dt * rn.s[i];
    gemv(0, nx, nx, 1.0, Pn, nx, q.Fx, 0.0, PF);
    for (int i = 0; i < nx; ++i) r.s[i] -= PF[i];
    for (int i = 0; i < nv; ++i) r.s[nv + i] -= dt * PF[i];
    for (int i = 0; i < nx; ++i) r.s[i] -= q.lx[i];
    gemv(0, nx, nv, -1.0, q.Qxu, nx, r.k, 1.0, r.s);
#undef PN
#undef QXX
#undef QXU
  }
  free(Lg); free(X); free(GK); free(PF);
  return stat;
}

void orc_unconstr_forward(const rtoc_layout* L, int nstages, double dt, double* kkt, double* ric,
                          double* dir) {
  const int N = nstages - 1, nv = L->dims.nv, nx = L->nx;
  for (int st = 0; st <= N; ++st) {
    ric_view r = ric_at(L, ric, st);
    dir_view d = dir_at(L, dir, st);
    if (st < N) {
      kkt_view q = kkt_at(L, kkt, st);
      dir_view dn = dir_at(L, dir, st + 1);
      /* da = K dx + k  (stored in the du slot) :40 */
      gemv(1, nx, nv, 1.0, r.K, nx, d.dx, 0.0, d.du);
      for (int i = 0; i < nv; ++i) d.du[i] += r.k[i];
      for (int i = 0; i < nx; ++i) dn.dx[i] = q.Fx[i] + d.dx[i];
      for (int i = 0; i < nv; ++i) dn.dx[i] += dt * d.dx[nv + i];
      for (int i = 0; i < nv; ++i) dn.dx[nv + i] += dt * d.du[i];
    }
    gemv(0, nx, nx, 1.0, r.P, nx, d.dx, 0.0, d.dlmdgmm);
    for (int i = 0; i < nx; ++i) d.dlmdgmm[i] -= r.s[i];
  }
}

/* ------------------------------------------------------------------------- */
/* batch drivers (OpenMP over instances = the CPU baseline of bench.py)        */
/* ------------------------------------------------------------------------- */
void orc_riccati_sweep_batch(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch,
                             double* kkt, double* ric, double* dir, const double* dx0,
                             double max_dts0, unsigned* stat, int do_backward, int do_forward) {
  const size_t ks = (size_t)nstages * L->kkt.stride, rs = (size_t)nstages * L->ric.stride,
               ds = (size_t)nstages * L->dir.stride;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    unsigned st = 0;
    if (do_backward)
      st = orc_riccati_backward(L, grid, nstages, kkt + b * ks, ric + b * rs, max_dts0);
    if (do_forward) {
      if (dx0) memcpy(dir + b * ds + L->dir.off[RTOC_DIR_DX], dx0 + (size_t)b * L->nx,
                      sizeof(double) * L->nx);
      orc_riccati_forward(L, grid, nstages, kkt + b * ks, ric + b * rs, dir + b * ds);
    }
    if (stat) stat[b] = st;
  }
}

void orc_unconstr_sweep_batch(const rtoc_layout* L, int nstages, int batch, double dt, double* kkt,
                              double* ric, double* dir, const double* dx0, unsigned* stat) {
  const size_t ks = (size_t)nstages * L->kkt.stride, rs = (size_t)nstages * L->ric.stride,
               ds = (size_t)nstages * L->dir.stride;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    unsigned st = orc_unconstr_backward(L, nstages, dt, kkt + b * ks, ric + b * rs);
    if (dx0) memcpy(dir + b * ds + L->dir.off[RTOC_DIR_DX], dx0 + (size_t)b * L->nx,
                    sizeof(double) * L->nx);
    orc_unconstr_forward(L, nstages, dt, kkt + b * ks, ric + b * rs, dir + b * ds);
    if (stat) stat[b] = st;
  }
}

void orc_layout(const rtoc_dims* d, rtoc_layout* L) { rtoc_compute_layout(d, L); }

/* ------------------------------------------------------------------------- */
/* single-stage hooks for tests/test_oracle_reference_identities.py             */
/* (the per-stage entry points of RiccatiFactorizer, riccati_factorizer.hpp)     */
/* ------------------------------------------------------------------------- */
unsigned orc_stage_backward(const rtoc_layout* L, double* kkt_rec, double* ric_next_rec,
                            double* ric_rec_out, int ns, int sto, int sto_next) {
  ric_scratch w = scratch_alloc(L);
  kkt_view q = kkt_at(L, kkt_rec, 0);
  ric_view rn = ric_rec(L, ric_next_rec);
  ric_view r = ric_rec(L, ric_rec_out);
  unsigned st = backward_stage(L, &rn, &q, &r, &w, ns, sto, sto_next);
  scratch_free(&w);
  return st;
}

void orc_stage_backward_impact(const rtoc_layout* L, double* kkt_rec, double* ric_next_rec,
                               double* ric_rec_out, int sto) {
  ric_scratch w = scratch_alloc(L);
  kkt_view q = kkt_at(L, kkt_rec, 0);
  ric_view rn = ric_rec(L, ric_next_rec);
  ric_view r = ric_rec(L, ric_rec_out);
  backward_impact_stage(L, &rn, &q, &r, &w, sto);
  scratch_free(&w);
}

void orc_stage_phase_transition(const rtoc_layout* L, double* ric_rec_in, double* ric_m_rec,
                                double* policy_rec, int sto_next, double max_dts0) {
  ric_view r = ric_rec(L, ric_rec_in);
  ric_view m = ric_rec(L, ric_m_rec);
  ric_view pol = ric_rec(L, policy_rec);
  phase_transition(L, &r, &m, &pol, sto_next, max_dts0);
}

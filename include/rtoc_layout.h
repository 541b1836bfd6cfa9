/*
 * rtoc_layout.h -- problem dimensions, grid description and the packed HBM
 * record layout shared by the C ABI (rtoc.h), the HIP kernels, the C++ host
 * mirror and the test oracle.
 *
 * One OCP instance = `stages` records per buffer; one record = all fields of
 * one grid point, every field padded to a multiple of 8 doubles (64 B) so that
 * a wavefront streams a record with fully coalesced 16 B/lane loads.
 *
 * The fields restate the member lists of the reference containers
 *   SplitKKTMatrix            include/robotoc/core/split_kkt_matrix.hpp:18-488
 *   SplitKKTResidual          src/core/split_kkt_residual.cpp:7-20
 *   SplitRiccatiFactorization include/robotoc/riccati/split_riccati_factorization.hpp:15-227
 *   LQRPolicy / STOPolicy     include/robotoc/riccati/lqr_policy.hpp:16, sto_policy.hpp:16
 *   SplitDirection            src/core/split_direction.cpp:7-22
 *   ContactDynamicsData       include/robotoc/dynamics/contact_dynamics_data.hpp
 * Variable-size blocks (dimf, dims) use the reference's "max-size backing,
 * top-left active view" rule, i.e. the leading dimension is the max size.
 */
#ifndef RTOC_LAYOUT_H_
#define RTOC_LAYOUT_H_

#if defined(__HIPCC__)
#define RTOC_HD __host__ __device__
#else
#define RTOC_HD
#endif
/* the layout functions are usable in C++ constant expressions (kernels bake the field offsets of
 * their robot into immediates); plain inline functions in C */
#if defined(__cplusplus) && __cplusplus >= 201402L
#define RTOC_CONSTEXPR constexpr
#else
#define RTOC_CONSTEXPR
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* GridType, include/robotoc/ocp/grid_info.hpp:13-18 */
#define RTOC_GRID_INTERMEDIATE 0
#define RTOC_GRID_IMPACT 1
#define RTOC_GRID_LIFT 2
#define RTOC_GRID_TERMINAL 3

typedef struct rtoc_dims {
  int nv;     /* Robot::dimv()                                  */
  int nu;     /* Robot::dimu()                                  */
  int np;     /* Robot::dim_passive() (6 floating base, else 0) */
  int nf_max; /* Robot::max_dimf()                              */
  int ns_max; /* max switching-constraint dimension (= nf_max)  */
  int nc_max; /* max number of PDIPM inequality rows per stage  */
} rtoc_dims;

/* The subset of GridInfo (grid_info.hpp:24-93) the hot path reads. */
typedef struct rtoc_grid {
  int type;                 /* RTOC_GRID_*                                        */
  int sto;                  /* GridInfo::sto                                       */
  int sto_next;             /* GridInfo::sto_next                                  */
  int switching_constraint; /* GridInfo::switching_constraint                      */
  int dimf;                 /* ContactStatus::dimf() (ImpactStatus on Impact grids) */
  int dims;                 /* SplitKKTMatrix::dims(), 0 unless switching_constraint */
  int num_grids_in_phase;   /* GridInfo::num_grids_in_phase                        */
  int time_stage;           /* constraint mask level: 0,1,>=2; -1 on impact grids   */
  double dt;                /* GridInfo::dt                                        */
} rtoc_grid;

/* ---- KKT record: condensed SplitKKTMatrix + SplitKKTResidual ---------------- */
enum {
  RTOC_KKT_FXX = 0, /* nx*nx  Fxx = [[Fqq Fqv];[Fvq Fvv]]                */
  RTOC_KKT_FVU,     /* nv*nu                                             */
  RTOC_KKT_QXX,     /* nx*nx                                             */
  RTOC_KKT_QXU,     /* nx*nu                                             */
  RTOC_KKT_QUU,     /* nu*nu                                             */
  RTOC_KKT_FX,      /* nx     SplitKKTResidual::Fx                       */
  RTOC_KKT_LX,      /* nx                                                */
  RTOC_KKT_LU,      /* nu                                                */
  RTOC_KKT_FFX,     /* nx     SplitKKTMatrix::fx (STO)                   */
  RTOC_KKT_HX,      /* nx                                                */
  RTOC_KKT_HU,      /* nu                                                */
  RTOC_KKT_SCAL,    /* 8      [Qtt, Qtt_prev, h, 0...]                   */
  RTOC_KKT_PHIX,    /* ns_max*nx (ld ns_max)                             */
  RTOC_KKT_PHIU,    /* ns_max*nu                                         */
  RTOC_KKT_PHIT,    /* ns_max                                            */
  RTOC_KKT_PRES,    /* ns_max SplitKKTResidual::P()                      */
  RTOC_KKT_NFIELDS
};
#define RTOC_KKT_SCAL_QTT 0
#define RTOC_KKT_SCAL_QTT_PREV 1
#define RTOC_KKT_SCAL_H 2

/* ---- Riccati record: SplitRiccatiFactorization + LQRPolicy + STOPolicy ------ */
enum {
  RTOC_RIC_P = 0, /* nx*nx                                   */
  RTOC_RIC_S,     /* nx                                      */
  RTOC_RIC_PSI,   /* nx  Psi                                 */
  RTOC_RIC_PHI,   /* nx  Phi                                 */
  RTOC_RIC_PSIX,  /* nx  psi_x                               */
  RTOC_RIC_PHIX,  /* nx  phi_x                               */
  RTOC_RIC_PSIU,  /* nu  psi_u                               */
  RTOC_RIC_PHIU,  /* nu  phi_u                               */
  RTOC_RIC_SCAL,  /* 8   [xi,chi,rho,eta,iota,dtsdts,dts0,0] */
  RTOC_RIC_K,     /* nu*nx ROW-major (lqr_policy.hpp:18-19)  */
  RTOC_RIC_KV,    /* nu  LQRPolicy::k                        */
  RTOC_RIC_T,     /* nu                                      */
  RTOC_RIC_W,     /* nu                                      */
  RTOC_RIC_M,     /* ns_max*nx (ld ns_max)                   */
  RTOC_RIC_MV,    /* ns_max  m                               */
  RTOC_RIC_MT,    /* ns_max  mt                              */
  RTOC_RIC_MTN,   /* ns_max  mt_next                         */
  RTOC_RIC_DTSDX, /* nx  STOPolicy::dtsdx                    */
  RTOC_RIC_NFIELDS
};
#define RTOC_RIC_SCAL_XI 0
#define RTOC_RIC_SCAL_CHI 1
#define RTOC_RIC_SCAL_RHO 2
#define RTOC_RIC_SCAL_ETA 3
#define RTOC_RIC_SCAL_IOTA 4
#define RTOC_RIC_SCAL_DTSDTS 5
#define RTOC_RIC_SCAL_DTS0 6

/* ---- direction record: SplitDirection --------------------------------------- */
enum {
  RTOC_DIR_DX = 0,  /* nx                      */
  RTOC_DIR_DU,      /* nu                      */
  RTOC_DIR_DLMDGMM, /* nx                      */
  RTOC_DIR_DXI,     /* ns_max                  */
  RTOC_DIR_DTS,     /* 8 [dts, dts_next, 0...] */
  RTOC_DIR_DAF,     /* nv+nf_max               */
  RTOC_DIR_DBETAMU, /* nv+nf_max               */
  RTOC_DIR_DNUP,    /* 8 dnu_passive (np<=6)   */
  RTOC_DIR_NFIELDS
};

/* ---- contact-dynamics record: ContactDynamicsData + the pre-condensation
 *      KKT pieces consumed by condenseContactDynamics ------------------------- */
enum {
  /* inputs */
  RTOC_CDD_DIDDA = 0, /* nv*nv        dIDda (dIDddv on impact grids)           */
  RTOC_CDD_DIDCDQV,   /* nvf_max*nx   [dIDdq dIDdv; dCdq dCdv], ld nvf_max      */
  RTOC_CDD_DCDA,      /* nf_max*nv    dCda (dCdv on impact grids), ld nf_max    */
  RTOC_CDD_IDC,       /* nvf_max      [ID_full; C]                              */
  RTOC_CDD_QAA,       /* nv           Qaa.diagonal() (Qdvdv on impact grids)    */
  RTOC_CDD_QFF,       /* nf_max^2     ld nf_max                                 */
  RTOC_CDD_QQF,       /* nv*nf_max    ld nv                                     */
  RTOC_CDD_LA,        /* nv           la (ldv on impact grids)                  */
  RTOC_CDD_LF,        /* nf_max                                                 */
  RTOC_CDD_HA,        /* nv                                                     */
  RTOC_CDD_HF,        /* nf_max                                                 */
  RTOC_CDD_PHIA,      /* ns_max*nv    ld ns_max                                 */
  RTOC_CDD_LUP,       /* 8            lu_passive                                */
  /* outputs kept for expandContactDynamicsPrimal/Dual */
  RTOC_CDD_MJTJINV,   /* nvf_max^2    ld nvf_max                                */
  RTOC_CDD_MJD,       /* nvf_max*nx   MJtJinv_dIDCdqv                           */
  RTOC_CDD_MJIDC,     /* nvf_max      MJtJinv_IDC                               */
  RTOC_CDD_QAFQV,     /* nvf_max*nx                                             */
  RTOC_CDD_QAFU,      /* nvf_max*nv   Qafu_full                                 */
  RTOC_CDD_LAF,       /* nvf_max      [la; lf] condensed                        */
  RTOC_CDD_QXUP,      /* nx*8         Qxu_passive (ld nx, np cols)              */
  RTOC_CDD_QUUPTR,    /* 8*nu         Quu_passive_topRight (ld np)              */
  RTOC_CDD_HAF,       /* nvf_max      [ha; -hf]                                 */
  RTOC_CDD_NFIELDS
};

/* One PDIPM inequality row of a joint-limit component (the eight Joint{Position,Velocity,Acceleration,Torques}
 * {Lower,Upper}Limit classes, e.g. src/constraints/joint_torques_lower_limit.cpp:50-83,
 * joint_acceleration_lower_limit.cpp:50-95): a bound on a single primal variable.  g(z) = sign * z[index] - bound <= 0,
 * i.e. sign = -1 for a lower limit (xmin - x <= 0) and +1 for an upper limit (x - xmax <= 0).
 * RTOC_VAR_A rows act on Qaa.diagonal() / la of the ContactDynamicsData record (RTOC_CDD_QAA, RTOC_CDD_LA) BEFORE the
 * contact-dynamics condensation reads them (contact_dynamics.cpp:68-86), level 0, contact path only; rtoc_condense
 * leaves the updated diagonal in RTOC_CDD_QAA like the reference leaves it in kkt_matrix.Qaa. */
#define RTOC_VAR_Q 0
#define RTOC_VAR_V 1
#define RTOC_VAR_U 2
#define RTOC_VAR_A 3
typedef struct rtoc_box_row {
  int var;   /* RTOC_VAR_Q / _V / _U / _A                                                  */
  int index; /* entry of q / v / a (0..nv-1; joint limits use the tail nu entries) or u (0..nu-1) */
  int sign;  /* -1 lower limit, +1 upper limit                                            */
  int level; /* KinematicsLevel: 0 acceleration (torques), 1 velocity, 2 position;
                active on a grid iff time_stage >= level (src/constraints/constraints_data.cpp:20-45),
                never on impact grids (time_stage = -1) or the terminal grid                */
} rtoc_box_row;

/* ---- constraint record: ConstraintComponentData of all box rows
 *      (include/robotoc/constraints/constraint_component_data.hpp:61-114); each field pad8(nc_max) */
enum {
  RTOC_CON_SLACK = 0,
  RTOC_CON_DUAL,
  RTOC_CON_RESIDUAL,
  RTOC_CON_CMPL,
  RTOC_CON_COND,
  RTOC_CON_DSLACK,
  RTOC_CON_DDUAL,
  RTOC_CON_NFIELDS
};

/* ---- solution record: SplitSolution (include/robotoc/core/split_solution.hpp), the members
 *      SplitSolution::integrate updates (src/core/split_solution.cpp:58-90).  q has nv+1 slots
 *      (floating base: 7 base + joints); on impact grids the A slot holds dv. ---- */
enum {
  RTOC_SOL_Q = 0,
  RTOC_SOL_V,
  RTOC_SOL_A,
  RTOC_SOL_U,
  RTOC_SOL_F,
  RTOC_SOL_LMD,
  RTOC_SOL_GMM,
  RTOC_SOL_BETA,
  RTOC_SOL_MU,
  RTOC_SOL_NUP,
  RTOC_SOL_XI,
  RTOC_SOL_NFIELDS
};

/* ---- RTOC_BUF_SE3 record: StateEquationData::Fqq_inv, Fqq_prev_inv (6x6, column-major) ---- */
#define RTOC_SE3_FQQ_INV 0
#define RTOC_SE3_FQQ_PREV_INV 36
#define RTOC_SE3_STRIDE 72

/* ---- RTOC_BUF_CONE record: per ACTIVE contact k of the grid point (compacted), the Jacobians of
 *      its 5 friction-cone rows (friction_cone.cpp:143-191): dg_dq 5 x nv at k*5*nv, dg_df 5 x 3 at
 *      rtoc_cone_dgdf_off + k*15; both column-major with leading dimension 5 ---- */
static inline RTOC_HD RTOC_CONSTEXPR int rtoc_cone_dgdf_off(int nv, int max_contacts) {
  return (max_contacts * 5 * nv + 7) & ~7;
}
static inline RTOC_HD RTOC_CONSTEXPR int rtoc_cone_stride(int nv, int max_contacts) {
  return rtoc_cone_dgdf_off(nv, max_contacts) + ((max_contacts * 15 + 7) & ~7);
}

/* ---- RTOC_BUF_CONE record when WRENCH cones are set (rtoc_set_wrench_cones): per ACTIVE surface
 *      contact k of the grid point (compacted) the 17 x 6 cone matrix of ContactWrenchCone
 *      (ConstraintComponentData::J[i], contact_wrench_cone.cpp:70-78,282-313) at k*102, column-major
 *      with leading dimension 17 ---- */
#define RTOC_WRENCH_ROWS 17
#define RTOC_FRICTION_ROWS 5
static inline RTOC_HD RTOC_CONSTEXPR int rtoc_wrench_cone_stride(int max_contacts) {
  return (max_contacts * RTOC_WRENCH_ROWS * 6 + 7) & ~7;
}

typedef struct rtoc_record_layout {
  int off[24]; /* field offsets in doubles (indexed by the enums above) */
  int stride;  /* record size in doubles                                */
  int nfields;
} rtoc_record_layout;

typedef struct rtoc_layout {
  rtoc_dims dims;
  int nx;      /* 2*nv       */
  int nvf_max; /* nv+nf_max  */
  rtoc_record_layout kkt, ric, dir, cdd, con, sol;
} rtoc_layout;

static inline RTOC_HD RTOC_CONSTEXPR int rtoc_pad8(int n) { return (n + 7) & ~7; }

static inline RTOC_HD RTOC_CONSTEXPR void rtoc_record_finish(rtoc_record_layout* r, const int* sizes, int n) {
  int o = 0;
  for (int i = 0; i < n; ++i) {
    r->off[i] = o;
    o += rtoc_pad8(sizes[i]);
  }
  r->stride = o;
  r->nfields = n;
}

static inline RTOC_HD RTOC_CONSTEXPR void rtoc_compute_layout(const rtoc_dims* d, rtoc_layout* L) {
  const int nv = d->nv, nu = d->nu, nx = 2 * d->nv;
  const int nf = d->nf_max, ns = d->ns_max, nvf = d->nv + d->nf_max;
  L->dims = *d;
  L->nx = nx;
  L->nvf_max = nvf;
  {
    int s[RTOC_KKT_NFIELDS] = {0};
    s[RTOC_KKT_FXX] = nx * nx;
    s[RTOC_KKT_FVU] = nv * nu;
    s[RTOC_KKT_QXX] = nx * nx;
    s[RTOC_KKT_QXU] = nx * nu;
    s[RTOC_KKT_QUU] = nu * nu;
    s[RTOC_KKT_FX] = nx;
    s[RTOC_KKT_LX] = nx;
    s[RTOC_KKT_LU] = nu;
    s[RTOC_KKT_FFX] = nx;
    s[RTOC_KKT_HX] = nx;
    s[RTOC_KKT_HU] = nu;
    s[RTOC_KKT_SCAL] = 8;
    s[RTOC_KKT_PHIX] = ns * nx;
    s[RTOC_KKT_PHIU] = ns * nu;
    s[RTOC_KKT_PHIT] = ns;
    s[RTOC_KKT_PRES] = ns;
    rtoc_record_finish(&L->kkt, s, RTOC_KKT_NFIELDS);
  }
  {
    int s[RTOC_RIC_NFIELDS] = {0};
    s[RTOC_RIC_P] = nx * nx;
    s[RTOC_RIC_S] = nx;
    s[RTOC_RIC_PSI] = nx;
    s[RTOC_RIC_PHI] = nx;
    s[RTOC_RIC_PSIX] = nx;
    s[RTOC_RIC_PHIX] = nx;
    s[RTOC_RIC_PSIU] = nu;
    s[RTOC_RIC_PHIU] = nu;
    s[RTOC_RIC_SCAL] = 8;
    s[RTOC_RIC_K] = nu * nx;
    s[RTOC_RIC_KV] = nu;
    s[RTOC_RIC_T] = nu;
    s[RTOC_RIC_W] = nu;
    s[RTOC_RIC_M] = ns * nx;
    s[RTOC_RIC_MV] = ns;
    s[RTOC_RIC_MT] = ns;
    s[RTOC_RIC_MTN] = ns;
    s[RTOC_RIC_DTSDX] = nx;
    rtoc_record_finish(&L->ric, s, RTOC_RIC_NFIELDS);
  }
  {
    int s[RTOC_DIR_NFIELDS] = {0};
    s[RTOC_DIR_DX] = nx;
    s[RTOC_DIR_DU] = nu;
    s[RTOC_DIR_DLMDGMM] = nx;
    s[RTOC_DIR_DXI] = ns;
    s[RTOC_DIR_DTS] = 8;
    s[RTOC_DIR_DAF] = nvf;
    s[RTOC_DIR_DBETAMU] = nvf;
    s[RTOC_DIR_DNUP] = 8;
    rtoc_record_finish(&L->dir, s, RTOC_DIR_NFIELDS);
  }
  {
    int s[RTOC_CDD_NFIELDS] = {0};
    s[RTOC_CDD_DIDDA] = nv * nv;
    s[RTOC_CDD_DIDCDQV] = nvf * nx;
    s[RTOC_CDD_DCDA] = nf * nv;
    s[RTOC_CDD_IDC] = nvf;
    s[RTOC_CDD_QAA] = nv;
    s[RTOC_CDD_QFF] = nf * nf;
    s[RTOC_CDD_QQF] = nv * nf;
    s[RTOC_CDD_LA] = nv;
    s[RTOC_CDD_LF] = nf;
    s[RTOC_CDD_HA] = nv;
    s[RTOC_CDD_HF] = nf;
    s[RTOC_CDD_PHIA] = ns * nv;
    s[RTOC_CDD_LUP] = 8;
    s[RTOC_CDD_MJTJINV] = nvf * nvf;
    s[RTOC_CDD_MJD] = nvf * nx;
    s[RTOC_CDD_MJIDC] = nvf;
    s[RTOC_CDD_QAFQV] = nvf * nx;
    s[RTOC_CDD_QAFU] = nvf * nv;
    s[RTOC_CDD_LAF] = nvf;
    s[RTOC_CDD_QXUP] = nx * 8;
    s[RTOC_CDD_QUUPTR] = 8 * nu;
    s[RTOC_CDD_HAF] = nvf;
    rtoc_record_finish(&L->cdd, s, RTOC_CDD_NFIELDS);
  }
  {
    int s[RTOC_CON_NFIELDS] = {0};
    for (int i = 0; i < RTOC_CON_NFIELDS; ++i) s[i] = d->nc_max;
    rtoc_record_finish(&L->con, s, RTOC_CON_NFIELDS);
  }
  {
    int s[RTOC_SOL_NFIELDS] = {0};
    s[RTOC_SOL_Q] = nv + 1;
    s[RTOC_SOL_V] = nv;
    s[RTOC_SOL_A] = nv;
    s[RTOC_SOL_U] = nu;
    s[RTOC_SOL_F] = nf;
    s[RTOC_SOL_LMD] = nv;
    s[RTOC_SOL_GMM] = nv;
    s[RTOC_SOL_BETA] = nv;
    s[RTOC_SOL_MU] = nf;
    s[RTOC_SOL_NUP] = 8;
    s[RTOC_SOL_XI] = ns;
    rtoc_record_finish(&L->sol, s, RTOC_SOL_NFIELDS);
  }
}

#ifdef __cplusplus
}
#endif
#endif /* RTOC_LAYOUT_H_ */

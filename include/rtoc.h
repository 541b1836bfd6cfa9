/*
 * rtoc.h -- C ABI of the MI355X-native Riccati / KKT-condensation engine.
 *
 * This is the drop-in boundary for robotoc's per-iteration KKT hot path.  The
 * reference (mayataka/robotoc) has no FFI of its own: the seam is its C++ class
 * API.  Every entry point below therefore names the reference C++ method it
 * replaces (paths relative to the reference tree):
 *
 *   rtoc_condense              DirectMultipleShooting::evalKKT tail
 *                              (src/ocp/intermediate_stage.cpp:134-148,
 *                               src/ocp/impact_stage.cpp:108-121):
 *                              Constraints::condenseSlackAndDual
 *                              (src/constraints/constraints.cpp:322-357),
 *                              condenseContactDynamics
 *                              (src/dynamics/contact_dynamics.cpp:55-164),
 *                              condenseImpactDynamics
 *                              (src/dynamics/impact_dynamics.cpp:38-80)
 *   rtoc_riccati_backward      RiccatiRecursion::backwardRiccatiRecursion
 *                              (src/riccati/riccati_recursion.cpp:32-80)
 *   rtoc_riccati_forward       RiccatiRecursion::forwardRiccatiRecursion
 *                              (src/riccati/riccati_recursion.cpp:83-131)
 *   rtoc_riccati_sweep         the two calls above back to back, as OCPSolver::updateSolution
 *                              issues them (src/solver/ocp_solver.cpp:120-126), pipelined over
 *                              instance chunks on two HIP streams
 *   rtoc_unconstr_backward /   UnconstrRiccatiRecursion::{backward,forward}RiccatiRecursion
 *   rtoc_unconstr_forward      (src/riccati/unconstr_riccati_recursion.cpp:26-48)
 *   rtoc_expand                DirectMultipleShooting::computeStepSizes + the
 *                              expandDual half of integrateSolution
 *                              (src/ocp/direct_multiple_shooting.cpp:174-241;
 *                               src/dynamics/contact_dynamics.cpp:167-202;
 *                               src/constraints/constraints.cpp:360-458)
 *   rtoc_update                Constraints::updateSlack/updateDual
 *                              (include/robotoc/constraints/constraints_impl.hxx:167-182)
 *
 * Conventions
 *   - plain C, no torch / Eigen types; all matrices are IEEE fp64, column-major
 *     (Eigen default), except LQRPolicy::K which is row-major as in the reference
 *     (include/robotoc/riccati/lqr_policy.hpp:18-19).
 *   - data live in packed, stage-contiguous records: buffer[instance][stage][field].
 *     Field offsets come from rtoc_compute_layout() (rtoc_layout.h); every field
 *     starts on a 64-byte boundary.
 *   - every call returns an int status (RTOC_OK == 0, negative = API misuse);
 *     numerical failures (non-SPD Quu / S, NaN) never throw, they set bits in the
 *     per-instance status words readable with rtoc_status().
 *   - calls are asynchronous on the context's HIP stream; rtoc_download /
 *     rtoc_status / rtoc_sync synchronise.
 *   - the HIP path has NO CPU fallback: if the device or the kernels are not
 *     available rtoc_create fails with RTOC_ERR_NO_DEVICE.
 */
#ifndef RTOC_H_
#define RTOC_H_

#include <stddef.h>
#include <stdint.h>

#include "rtoc_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes -------------------------------------------------------- */
#define RTOC_OK 0
#define RTOC_ERR_BAD_ARG (-1)
#define RTOC_ERR_UNSUPPORTED_DIMS (-2)
#define RTOC_ERR_NO_DEVICE (-3)
#define RTOC_ERR_HIP (-4)
#define RTOC_ERR_NOT_READY (-5)
#define RTOC_ERR_RCCL (-6)
#define RTOC_ERR_IO (-7)

/* ---- per-instance numerical status bits (rtoc_status) --------------------- */
#define RTOC_STAT_QUU_NOT_SPD 0x1u  /* LLT(Quu) hit a non-positive pivot (riccati_factorizer.cpp:49-50) */
#define RTOC_STAT_S_NOT_SPD 0x2u    /* LLT(S) of the switching-constraint Schur complement (:63-64)   */
#define RTOC_STAT_NAN 0x4u          /* NaN/Inf in K, k, M, m (:75-79)                                   */
#define RTOC_STAT_M_NOT_SPD 0x8u    /* LLT(dIDda) or LLT(J Minv J^T) failed in computeMJtJinv           */
#define RTOC_STAT_FXX_UNSTRUCTURED 0x10u /* the structure-exploiting backward kernel met an Fxx whose top half is not
                                     * [a I | c I] + two corners (a bound buffer rewritten behind the runtime's back,
                                     * RTOC_OPT_FXX_STRUCTURE): the instance's factorisation is not to be used; call
                                     * rtoc_check_fxx_structure (or set the option to 1) and repeat the sweep */

/* ---- buffers ---------------------------------------------------------------- */
enum rtoc_buffer {
  RTOC_BUF_KKT = 0,  /* [batch][stages][kkt.stride]  condensed KKT system (SplitKKTMatrix/Residual) */
  RTOC_BUF_RIC = 1,  /* [batch][stages][ric.stride]  SplitRiccatiFactorization + LQRPolicy + STOPolicy */
  RTOC_BUF_DIR = 2,  /* [batch][stages][dir.stride]  SplitDirection */
  RTOC_BUF_CDD = 3,  /* [batch][stages][cdd.stride]  ContactDynamicsData (+ pre-condensation KKT parts) */
  RTOC_BUF_CON = 4,  /* [batch][stages][con.stride]  ConstraintComponentData of the PDIPM rows */
  RTOC_BUF_DX0 = 5,  /* [batch][nx]                  initial state direction d[0].dx */
  RTOC_BUF_STEP = 6, /* [batch][2]                   max primal / dual step sizes */
  RTOC_BUF_SE3 = 7,  /* [batch][stages][RTOC_SE3_STRIDE] floating base: Fqq_inv, Fqq_prev_inv of
                      * StateEquationData (include/robotoc/dynamics/state_equation_data.hpp), 6x6 column-major each */
  RTOC_BUF_CONE = 8, /* [batch][stages][rtoc_cone_stride(nv, max_contacts)] friction-cone Jacobians of the
                      * active contacts (rtoc_layout.h); exists after rtoc_set_friction_cones.  With
                      * rtoc_set_wrench_cones: [batch][stages][rtoc_wrench_cone_stride(max_contacts)] */
  RTOC_BUF_SOL = 9,  /* [batch][stages][sol.stride]  SplitSolution (rtoc_integrate_solution) */
  RTOC_NUM_BUFFERS = 10
};

/* kernel-variant knobs (rtoc_set_option) */
enum rtoc_option {
  RTOC_OPT_WRITEBACK_KKT = 0, /* 1: backward writes the mutated Qxx,Qxu,Quu,lu back (reference in-place semantics) */
  RTOC_OPT_MAX_DTS0 = 1,      /* RiccatiRecursion(ocp, max_dts0) / setRegularization; value = double bits */
  RTOC_OPT_BACKWARD_WAVES = 2, /* waves per OCP instance in the backward kernel (0 = default for the dims) */
  RTOC_OPT_CONTACT_INV_DAMPING = 3, /* RobotModelInfo::contact_inv_damping (robot.hxx:662-664); value = double bits */
  RTOC_OPT_SWEEP_CHUNKS = 4, /* instance chunks of rtoc_riccati_sweep's backward/forward pipeline (1..16, default 1 = plain sequence) */
  RTOC_OPT_CONDENSE_SPLIT = 5, /* rtoc_condense: 1 = MJtJinv (and the cone rows) in a high-occupancy kernel of their own ahead of the
                              * condensation kernel; 0 = ONE kernel per grid point (wave 0 assembles MJtJinv, wave 1 condenses the cone
                              * rows, wave 2 stages the inputs; MJtJinv goes to HBM once, for the expansion only).  Default per robot
                              * shape: 0 where five work items of the one-kernel form fit a CU (quadruped-size shapes: same time,
                              * one launch and ~5 % of the HBM traffic less), 1 otherwise (iCub-size shapes: 3.2 vs 3.6 ms).
                              * rtoc_get_option reads the value in force.  Environment: RTOC_CONDENSE_SPLIT=0|1, read by rtoc_create,
                              * replaces the per-shape default of contexts created afterwards (how tools/gpu_dev.sh runs the
                              * whole GPU suite on both pipelines); rtoc_set_option still overrides it. */
  RTOC_OPT_BACKWARD_SCAN = 6, /* 1: rtoc_riccati_backward (and everything built on it) runs the recursion as a scan
                              * over the horizon -- interval elements of all grid points, ceil(log2(nstages))
                              * combination levels, then all policies at once -- instead of the serial chain
                              * (riccati_recursion.cpp:32-80); rtoc_riccati_forward likewise as a prefix scan of
                              * the closed-loop maps (:83-131).  For FEW instances (one MPC problem): latency of a
                              * sweep, not throughput of a batch.  Same outputs (P, s, K, k, M, m) to <= 1e-8
                              * relative.  Grids with switching-time optimisation: the MATRIX half (P, K, M) is the same scan,
                              * the vector half (s, k, m, Psi, Phi, the STO scalars and policies) one serial pass of four barrier-
                              * separated phases per grid point behind it (riccati_scan_sto.hpp); the forward recursion stays serial there.
                              * 2: automatic -- the scan for batches of at most 8 instances (where it is faster
                              * on MI355X), the serial kernels above.  Needs Quu > 0 of every stage by itself (the
                              * serial recursion only needs Quu + B^T P+ B > 0); a violation sets
                              * RTOC_STAT_QUU_NOT_SPD.  Default 0. */
  RTOC_OPT_CONDENSE_KEEP_QAF = 7, /* 1: rtoc_condense also stores ContactDynamicsData::Qafqv and Qafu_full in the
                              * RTOC_BUF_CDD record.  The reference keeps them as scratch for
                              * expandContactDynamicsDual (contact_dynamics.cpp:190-191); rtoc_expand rebuilds the two
                              * products from Qaa, Qff, Qqf and the primal expansion instead, so by default (0) the
                              * 1.6k doubles per grid point are neither written nor read back. */
  RTOC_OPT_FXX_STRUCTURE = 8, /* How rtoc_riccati_backward treats the top half of Fxx, which linearizeStateEquation /
                              * correctLinearizeStateEquation leave as [a I | c I] plus two dense 6 x 6 floating-base
                              * corners (src/dynamics/state_equation.cpp:52-55,80-82).  0 (default): automatic -- the
                              * records are checked on the device once after they change through this API
                              * (rtoc_upload / rtoc_bind / rtoc_load_stage_dump; rtoc_check_fxx_structure forces it) and
                              * the structure-exploiting kernel is used iff EVERY record has the shape, the dense one
                              * otherwise (same results to round-off).  1: always the dense kernel.  2: the caller
                              * asserts the structure (no check; wrong results if it does not hold).  A host that
                              * rewrites a BOUND buffer in place (rtoc_bind / rtoc_device_ptr: the runtime cannot see the
                              * writes) is covered in mode 0 all the same: the quadruped shapes' register-resident kernel
                              * verifies the structured rows of every record it factorises -- they are in its LDS anyway --
                              * and sets RTOC_STAT_FXX_UNSTRUCTURED on an instance that breaks the shape; the iCub-size
                              * shapes' register-wide kernel never loads those rows, so on a bound buffer the device check
                              * runs again before every backward recursion (one pass over the top halves: ~8 % of that
                              * recursion; rtoc_upload-ed buffers are checked once per upload).  Mode 2 switches both off. */
  RTOC_OPT_GRAPH = 9, /* 1: rtoc_riccati_sweep and rtoc_newton_iteration replay their launch sequence from a captured
                      * hipGraph (captured on the second call after any change of grid, options, buffers, rows or
                      * cones; arguments kkt_tol / tau are part of the key).  For the single-OCP latency path, whose
                      * ~25 small kernels are launch-bound.  The calls stay asynchronous on the context's stream; the
                      * stream must not be capturing already.  Default 0. */
  RTOC_OPT_CONE_JACOBIAN = 14, /* dg/dq of the friction-cone rows evaluated on the device (rtoc_contact_eval_kkt).  0 (default): as
                      * the reference composes it -- the LOCAL-frame angular Jacobian column of the contact frame crossed
                      * with the WORLD-frame force (robot.hxx:247-253, 275-287).  1: w_world x f_W, the derivative of
                      * R_wf(q) f (what finite differences give).  They coincide when the contact frame is world-aligned. */
  RTOC_OPT_LINEARIZE_DOFS_PER_PASS = 15, /* tangent directions (dofs, three lanes each) one pass of rtoc_linearize_contact_dynamics'
                      * walk carries, 1..21; 0 (default): chosen per robot model -- fewer lanes per pass = less LDS per wave = more
                      * waves per CU, against more passes over the bodies the pass's dofs reach (ANYmal: 18 = one pass; iCub nv = 32: 21, nv = 35: 19).
                      * rtoc_get_option reads the value in force (0 before rtoc_set_robot_model). */
  RTOC_OPT_LINEARIZE_FUSED = 13, /* 0 (default): rtoc_linearize_contact_dynamics computes the values of the recursion in a
                      * level-parallel pre-pass (lanes = bodies; 64 doubles per body and grid point of scratch) and the
                      * tangent walk reads them; 1: one kernel, every lane recomputes the values along its walk (no scratch) */
  RTOC_OPT_UNCONSTR_DENSE = 12, /* 0 (default): rtoc_unconstr_backward / _forward run the structured recursion (block adds of P+,
                      * unconstr_backward_riccati_recursion_factorizer.cpp:27-70); 1: the general kernels on materialised A, B
                      * (what RTOC_OPT_BACKWARD_SCAN needs; same results to round-off) */
  RTOC_OPT_IMPACT_CONES = 11, /* 1 (default): the friction / wrench cone rows also act on impact grids (a Constraints object
                      * holding FrictionCone AND ImpactFrictionCone, examples/anymal/run.cpp:173-181); 0: impact grids carry no
                      * cone rows (FrictionCone only, examples/anymal/trot.cpp:134-146) -- condensation, expansion, step
                      * sizes, update and KKT error skip them. */
  RTOC_OPT_BACKWARD_REGISTER = 16, /* 1 (default): rtoc_riccati_backward runs the register-resident kernel (one wavefront per instance,
                      * P+ / s+ in MFMA accumulators, the stage record by LDS-DMA, switching-constraint grid points in factorised
                      * form; riccati_backward_rv.hpp) where it applies: shapes whose stacked operand [P+; PB^T] fills its 16-row
                      * tiles (nv = 18, nu = 12: ANYmal, A1), grids without switching-time optimisation, RTOC_OPT_WRITEBACK_KKT = 0
                      * and the default RTOC_OPT_BACKWARD_WAVES.  On the iCub-size shapes (nx = 64 / 70) its counterpart is the
                      * register-wide kernel (riccati_backward_rw.hpp, nx = 64: one wavefront per instance and SIMD, P+ in 16 accumulator
                      * tiles; riccati_backward_rw2.hpp, nx = 70: two wavefronts per instance, P+ in 25 tiles in each, the stage split
                      * between them by role and by column tiles; the dense rows of a STRUCTURED Fxx staged in LDS; checked on the device
                      * like RTOC_OPT_FXX_STRUCTURE -- on a bound record buffer before every recursion unless that option is 2;
                      * switching-constraint grid points as one-stage launches of the tile-split kernel): with 1 on batches of more
                      * instances than the device has compute units (below that the tile-split kernel's four waves per instance
                      * finish a horizon sooner), with 2 on every batch.  Elsewhere, and with 0, the role-split / tile-split
                      * kernels run.  Same results to fp64 round-off (tests/test_backward_register.py). */
  RTOC_OPT_CONDENSE_REGISTER = 17, /* 1 (default): rtoc_condense runs the register-chained kernel (one wavefront per grid point, the saddle
                      * inverse read once into MFMA accumulators, every product of condenseContactDynamics chained through
                      * register layouts, friction-cone rows condensed inside it; condense_rv.hpp) on the CONTACT grid points of shapes
                      * it is laid out for (nv + nf_max <= 32, 32 < 2 nv <= 46: ANYmal, A1) -- impact grid points,
                      * RTOC_OPT_CONDENSE_SPLIT = 1, RTOC_OPT_CONDENSE_KEEP_QAF = 1 and contexts with WRENCH cone rows run the
                      * role-split kernels.  2: also with wrench cone rows (their own kernel first).  0: never.  Same results to fp64
                      * round-off (tests/test_condense_register.py).  RTOC_CONDENSE_REGISTER=0|1|2 in the environment sets the
                      * default of contexts created afterwards. */
  RTOC_OPT_SWITCHING_TRANSPORT = 10 /* Free-flyer block of Phiq / Phiv / Phia in rtoc_contact_eval_kkt's switching-constraint
                      * rows.  0 (default): as the reference composes it -- it hands pinocchio::dIntegrateTransport the
                      * transposed Jacobian (robot.hxx:69-72, :88-91), which yields Pq dIntegrate^T.  1: the chain rule
                      * Pq dIntegrate (what finite differences of P give).  The two agree to first order in the base
                      * displacement (dt1 + dt2) v + dt1 dt2 a. */
};

typedef struct rtoc_ctx rtoc_ctx;

/* Library / device discovery. rtoc_device_count() < 1 means the HIP path is unusable. */
int rtoc_version(void);
int rtoc_device_count(void);
/* Attainable HBM bandwidth of `device`, measured now with this library's own streaming kernels (16 B per lane, all waves
 * sweeping memory as one front, eight loads in flight per wave): a pure read of `bytes` and a copy of `bytes` (read + write,
 * counted as 2 x bytes), best of five launches each, GB/s.  The denominators the bench's roofline fractions are quoted against
 * beside the 8 TB/s spec (a hipMemcpy-style device copy reaches less than these).  Either pointer may be NULL.
 * `bytes` >= RTOC_BANDWIDTH_PROBE_MIN_BYTES (one trip of every wave of the probe: 512 MiB), else RTOC_ERR_BAD_ARG; it is rounded
 * down to a multiple of that. */
#define RTOC_BANDWIDTH_PROBE_MIN_BYTES ((size_t)512 << 20)
int rtoc_bandwidth_probe(int device, size_t bytes, double* read_gbs, double* copy_gbs);
/* 1 if (nv,nu,np) has a compiled kernel specialisation, else 0. */
int rtoc_dims_supported(const rtoc_dims* dims);

/* Create a context for `batch` independent OCP instances of at most
 * `max_stages` grid points each (= N+1+lifts+2*impacts, time_discretization.cpp:45)
 * on HIP device `device`. Allocates all RTOC_BUF_* buffers in HBM.
 * Mirrors the sizing done in OCPSolver's constructor (src/solver/ocp_solver.cpp:20-24). */
int rtoc_create(const rtoc_dims* dims, int max_stages, int batch, int device, rtoc_ctx** out);
int rtoc_destroy(rtoc_ctx* ctx);
/* Deep copy of a context on its device: dimensions, grid, constraint rows, cone set-up, options and every
 * allocated buffer (device-to-device).  Gives the C++ mirrors the value semantics of the reference's solver
 * objects (RiccatiRecursion / OCPSolver are copyable, riccati_recursion.hpp:40-60, ocp_solver.hpp:62-77).
 * Buffers bound to caller-owned memory (rtoc_bind) are copied into buffers the clone owns. */
int rtoc_clone(rtoc_ctx* ctx, rtoc_ctx** out);

/* Layout actually used by the context (identical to rtoc_compute_layout(dims)). */
int rtoc_get_layout(const rtoc_ctx* ctx, rtoc_layout* out);

/* Grid of the current discretisation: `nstages` = TimeDiscretization::size()
 * (terminal stage included, its type must be RTOC_GRID_TERMINAL). Shared by all
 * instances of the batch. Mirrors RiccatiRecursion::resizeData. */
int rtoc_set_grid(rtoc_ctx* ctx, const rtoc_grid* grid, int nstages);

/* Use a caller-owned HIP stream (hipStream_t passed as void*); NULL = the context's own stream.  (The context owns a second
 * stream for work it runs beside that one -- the chunks of rtoc_riccati_sweep, the setZero of rtoc_contact_eval_kkt --, forked
 * from and joined to this stream by events: callers only ever order against the stream given here.) */
int rtoc_set_stream(rtoc_ctx* ctx, void* hip_stream);
int rtoc_set_option(rtoc_ctx* ctx, int option, int64_t value);
/* the value in force of an integer-valued option (RTOC_ERR_BAD_ARG for the double-valued ones: RTOC_OPT_MAX_DTS0,
 * RTOC_OPT_CONTACT_INV_DAMPING, and for RTOC_OPT_IMPACT_CONES-style numbers this header does not list) */
int rtoc_get_option(rtoc_ctx* ctx, int option, int64_t* value);

/* Host <-> HBM transfers of whole buffers (count in doubles, from the buffer start +offset). */
int rtoc_upload(rtoc_ctx* ctx, int buffer, size_t offset, const double* host, size_t count);
int rtoc_download(rtoc_ctx* ctx, int buffer, size_t offset, double* host, size_t count);
/* Device pointer / element count of a buffer (for zero-copy interop and RCCL). */
void* rtoc_device_ptr(rtoc_ctx* ctx, int buffer);
size_t rtoc_buffer_count(const rtoc_ctx* ctx, int buffer);
/* Replace a buffer by caller-owned device memory of at least rtoc_buffer_count doubles. */
int rtoc_bind(rtoc_ctx* ctx, int buffer, void* device_ptr);

/* Joint-limit (box) inequality rows handled by the PDIPM condensation / expansion; shared by all
 * instances and grids (the stage mask follows rtoc_box_row::level).  nrows <= dims.nc_max.
 * Mirrors Constraints::add(...) of the six joint-limit components (examples/anymal/trot.cpp:134-146). */
int rtoc_set_constraint_rows(rtoc_ctx* ctx, const rtoc_box_row* rows, int nrows);

/* Linearised friction cones (FrictionCone / ImpactFrictionCone, src/constraints/friction_cone.cpp):
 * 5 PDIPM rows per active contact with dense Jacobians (RTOC_BUF_CONE).  Their
 * ConstraintComponentData occupy the last 5*max_contacts rows of the RTOC_BUF_CON record
 * (row nc_max - 5*max_contacts + 5k + j for the k-th ACTIVE contact of the grid point), so
 * nrows + 5*max_contacts <= dims.nc_max.  contact_dim = 3 (point) or 6 (surface contact: the cone
 * acts on the first 3 force components).  max_contacts = 0 switches them off.  Once set,
 * rtoc_condense / rtoc_expand / rtoc_update include these rows. */
int rtoc_set_friction_cones(rtoc_ctx* ctx, int max_contacts, int contact_dim);

/* Contact wrench cones of surface contacts (ContactWrenchCone, src/constraints/contact_wrench_cone.cpp:
 * condenseSlackAndDual :209-238, expandSlackAndDual :241-270): 17 PDIPM rows per active surface contact
 * acting on its 6-d wrench only, g = cone * f.  The 17 x 6 cone matrices are handed over in the
 * RTOC_BUF_CONE record (rtoc_layout.h; rtoc_wrench_cone_matrix fills one), the rows' ConstraintComponentData
 * are the last 17*max_contacts rows of the RTOC_BUF_CON record (row nc_max - 17*max_contacts + 17k + j for
 * the k-th ACTIVE contact), nrows + 17*max_contacts <= dims.nc_max, 6*max_contacts <= dims.nf_max.
 * A context carries either friction or wrench cones: setting one kind switches the other off. */
int rtoc_set_wrench_cones(rtoc_ctx* ctx, int max_contacts);
/* computeCone (contact_wrench_cone.cpp:282-303): the cone matrix of a rectangular sole 2X x 2Y with
 * friction coefficient mu; out: 17 x 6 column-major (ld 17).  Host helper, no device work. */
int rtoc_wrench_cone_matrix(double X, double Y, double mu, double* out);

/* ---- the hot path ---------------------------------------------------------- */
int rtoc_condense(rtoc_ctx* ctx);
int rtoc_riccati_backward(rtoc_ctx* ctx);
int rtoc_riccati_forward(rtoc_ctx* ctx);
/* backward + forward of the whole batch; same results as the two calls in sequence. */
int rtoc_riccati_sweep(rtoc_ctx* ctx);
int rtoc_unconstr_backward(rtoc_ctx* ctx, double dt);
int rtoc_unconstr_forward(rtoc_ctx* ctx, double dt);
/* UnconstrDynamics::condenseUnconstrDynamics on every non-terminal grid point
 * (src/dynamics/unconstr_dynamics.cpp:67-88).  Records: KKT.Quu/lu/Qxu hold Qaa/la/[Qqa;Qva] (the
 * acceleration is the Riccati control); CDD.dIDCdqv = [dID_dq | dID_dv], CDD.dIDda = dID_da,
 * CDD.IDC = ID, CDD.Qaa = diag(Quu) and CDD.la = lu of the torque cost; CDD.MJtJinv = the full Quu (nv x nv), of which
 * only the off-diagonal entries are read, by expandDual (leave it zero for a diagonal cost).  nu == nv, nf_max == 0 only. */
int rtoc_unconstr_condense(rtoc_ctx* ctx);
/* UnconstrDynamics::expandPrimal + expandDual (unconstr_dynamics.cpp:91-104) after
 * rtoc_unconstr_forward: DIR.daf <- da (the Riccati control), DIR.du <- torque direction,
 * DIR.dbetamu <- dbeta. */
int rtoc_unconstr_expand(rtoc_ctx* ctx, double dt);
/* expandPrimal + fraction-to-boundary + expandDual; step sizes land in RTOC_BUF_STEP. */
int rtoc_expand(rtoc_ctx* ctx, double fraction_to_boundary_rule);
/* slack/dual update with the per-instance step sizes in RTOC_BUF_STEP. */
int rtoc_update(rtoc_ctx* ctx);

/* ---- floating-base corrections (need RTOC_BUF_SE3; no-ops of the reference for fixed bases) ----
 * correctLinearizeStateEquation / correctLinearizeImpactStateEquation on every grid point
 * (src/dynamics/state_equation.cpp:68-88, impact_state_equation.cpp:57-72).  rtoc_condense calls it
 * itself after the dynamics condensation once RTOC_BUF_SE3 has been uploaded or bound
 * (IntermediateStage::evalKKT order, intermediate_stage.cpp:134-139). */
int rtoc_correct_state_equation(rtoc_ctx* ctx);
/* correctCostateDirection on every grid point (state_equation.cpp:91-96); rtoc_expand calls it
 * itself after the dual expansion (intermediate_stage.cpp:171-173). */
int rtoc_correct_costate_direction(rtoc_ctx* ctx);
/* computeInitialStateDirection's floating-base part on RTOC_BUF_DX0, in place
 * (state_equation.cpp:99-109): upload dq0 = q0 (-) q, dv0 = v0 - v, then call this once. */
int rtoc_compute_initial_state_direction(rtoc_ctx* ctx);

/* Checks (on the device, ~0.2 ms per 4096 x 47 ANYmal records) whether the top half of every Fxx in RTOC_BUF_KKT has the
 * state-equation structure RTOC_OPT_FXX_STRUCTURE describes, caches the answer for rtoc_riccati_backward and
 * returns it in *structured (may be NULL).  Synchronises the stream. */
int rtoc_check_fxx_structure(rtoc_ctx* ctx, int* structured);

/* Per-instance status words (RTOC_STAT_* bits), synchronises the stream. */
int rtoc_status(rtoc_ctx* ctx, uint32_t* host_flags, int count);
int rtoc_clear_status(rtoc_ctx* ctx);
int rtoc_sync(rtoc_ctx* ctx);

/* Time `reps` back-to-back launches of one phase with HIP events recorded on the
 * context's stream; returns the mean milliseconds per launch in *ms.
 * phase: 0 backward, 1 forward, 2 condense, 3 expand, 4 backward+forward, 5 update,
 * 6 rtoc_newton_iteration(kkt_tol = 0, tau = 0.995), 7 / 8 rtoc_linearize_contact_dynamics(augment_residual = 0 / 1)
 * (rtoc_robot.h). */
int rtoc_time_phase(rtoc_ctx* ctx, int phase, int reps, float* ms);

/* ---- KKT error of every instance (first piece of the on-device Newton loop) ----
 * sqrt of the squared KKT residual summed over the horizon, on the PRE-condensation records, as
 * OCPSolver::KKTError() (src/solver/ocp_solver.cpp:429-431) without its STO term:
 * SplitKKTResidual::KKTError (split_kkt_residual.hxx:90-104) + ContactDynamicsData::KKTError
 * (contact_dynamics_data.hpp:204-206, if RTOC_BUF_CDD exists) + ConstraintComponentData::KKTError of the
 * active box / cone rows (constraint_component_data.hpp:122-124, if rows are set).  host_out: [count<=batch]. */
int rtoc_kkt_error(rtoc_ctx* ctx, double* host_out, int count);

/* SwitchingTimeOptimization::evalKKT downstream of the (host-side, scalar) STO cost and dwell-time constraints
 * (src/sto/switching_time_optimization.cpp:105-137), after rtoc_condense like OCPSolver::updateSolution orders them
 * (ocp_solver.cpp:118-119): for every instance the gradient lt and the diagonal of the Hessian Qtt_ of its
 * `num_events` discrete events (host arrays [batch][num_events], event order = grid order) are scattered into h / Qtt
 * of the grid point after an impact and of the lift grid point, and the STO term of the squared KKT error -- the
 * squared differences of the per-phase Hamiltonian sums across STO-enabled events -- is returned in
 * host_err_sq[count <= batch] (may be NULL).  OCPSolver::KKTError() = sqrt(rtoc_kkt_error^2 + that + the STO
 * constraints' own residual, which stays on the host). */
int rtoc_sto_eval_kkt(rtoc_ctx* ctx, const double* host_lt, const double* host_qtt_diag, int num_events,
                      double* host_err_sq, int count);

/* ---- SwitchingTimeOptimization resident on the device (src/sto/switching_time_optimization.cpp, src/sto/sto_constraints.cpp) ----
 * The batch shares the grid STRUCTURE (rtoc_set_grid: event order, grid points per phase -- DiscretizationMethod::PhaseBased,
 * which OCPSolver selects whenever the OCP has an STO problem, ocp_solver.cpp:46-48); every instance owns its event times, hence
 * its own time steps, dwell-time rows and switching-time step.
 * rtoc_sto_set_problem: t0 = the `t` of OCPSolver::updateSolution, T = OCP::T, event_times = ContactSequence::impactTime /
 * liftTime of the `num_events` <= 15 discrete events on the horizon in grid order ([num_events], or [batch][num_events] if
 * per_instance), min_dwell_times[num_events + 1] / barrier_param / fraction_to_boundary_rule = the STOConstraints object
 * (sto_constraints.cpp:12-59).  num_events must equal the number of impact + lift grid points of the grid; num_events == 0
 * switches the STO problem off.  Once set:
 *   - every kernel that reads GridInfo::dt reads the instance's own time step, recomputed from its event times by
 *     rtoc_sto_correct_time_steps = TimeDiscretization::correctTimeSteps (time_discretization.cpp:179-221), which
 *     rtoc_contact_eval_kkt calls first like updateSolution does (ocp_solver.cpp:115-117);
 *   - rtoc_newton_iteration (and rtoc_contact_update_solution) run the STO half of updateSolution: rtoc_sto_eval_kkt_device
 *     after the condensation (sto_.evalKKT :119 -- STO cost terms + regularisation, linearizeConstraints and
 *     condenseSlackAndDual of the dwell-time rows, scatter into h / Qtt, STO term of the KKT error; rtoc_kkt_error's value
 *     becomes OCPSolver::KKTError() = sqrt(dms + sto) :429-431), rtoc_sto_compute_step_sizes after the expansion
 *     (sto_.computeStepSizes + min with the stages' step sizes :128-132), rtoc_sto_integrate_solution (:143: event times,
 *     slack, dual).  The three are also callable on their own (a host that drives the iteration call by call).
 * rtoc_sto_set_regularization: sto_.setRegularization (ocp_solver.cpp:169-176; the host owns the schedule).
 * rtoc_sto_set_cost_terms: gradient / Hessian diagonal of an STOCostFunction evaluated by the host ([batch][num_events] each;
 * NULL, NULL: none -- the reference ships no STO cost component, its examples pass an empty STOCostFunction).
 * rtoc_sto_init_constraints: sto_.initConstraints (STOConstraints::setSlackAndDual, sto_constraints.cpp:148-172).
 * Getters (synchronise): event times [count][num_events], time steps [count][nstages], the dwell-time rows' data
 * [count][6][16] = slack, dual, residual, cmpl, dslack, ddual, and what the last evalKKT scattered (lt, diag Qtt, squared
 * STO KKT term; any pointer may be NULL). */
int rtoc_sto_set_problem(rtoc_ctx* ctx, double t0, double T, const double* event_times, int num_events, int per_instance,
                         const double* min_dwell_times, double barrier_param, double fraction_to_boundary_rule);
int rtoc_sto_set_regularization(rtoc_ctx* ctx, double sto_reg);
int rtoc_sto_set_cost_terms(rtoc_ctx* ctx, const double* host_lt, const double* host_qtt_diag);
int rtoc_sto_init_constraints(rtoc_ctx* ctx);
/* slack / dual of the dwell-time rows from the host instead ([batch][num_events + 1] each, positive): a warm start */
int rtoc_sto_set_slack_dual(rtoc_ctx* ctx, const double* host_slack, const double* host_dual);
int rtoc_sto_correct_time_steps(rtoc_ctx* ctx);
int rtoc_sto_eval_kkt_device(rtoc_ctx* ctx);
int rtoc_sto_compute_step_sizes(rtoc_ctx* ctx);
int rtoc_sto_integrate_solution(rtoc_ctx* ctx);
int rtoc_sto_get_event_times(rtoc_ctx* ctx, double* host_out, int count);
int rtoc_sto_get_time_steps(rtoc_ctx* ctx, double* host_out, int count);
int rtoc_sto_get_constraint_data(rtoc_ctx* ctx, double* host_out, int count);
int rtoc_sto_get_kkt_terms(rtoc_ctx* ctx, double* host_lt, double* host_qtt_diag, double* host_err_sq, int count);

/* LineSearchFilter (src/line_search/line_search_filter.cpp:26-60) of every instance, resident on the device: one call is
 * the accept test of LineSearch::lineSearchFilterMethod (src/line_search/line_search.cpp:63-83) for one trial step of
 * the whole batch.  For instance b with trial pair (cost[b], violation[b]) -- DirectMultipleShooting::getEval() of the
 * trial iterate: cost + cost_barrier, primal_feasibility -- isAccepted() is evaluated against its filter; accepted pairs
 * are augment()ed (dominated entries erased, order kept), and accepted[b] is set to 1, else 0.  An empty filter accepts
 * everything, so the first call seeds the filters with the current iterates (:58-62).  mask[b] == 0 skips the instance
 * (its step was accepted earlier in the backtracking loop); mask may be NULL.  Host arrays of `count` <= batch entries.
 * RTOC_LINE_SEARCH_FILTER_CAPACITY pairs per instance; an instance whose filter is full keeps its newest entries. */
#define RTOC_LINE_SEARCH_FILTER_CAPACITY 32
int rtoc_line_search_filter(rtoc_ctx* ctx, const double* cost, const double* violation, const int* mask, int count,
                            double cost_reduction_rate, double constraint_violation_reduction_rate, int* accepted);
/* LineSearch::clearHistory (line_search.cpp:52-54) for every instance. */
int rtoc_line_search_clear(rtoc_ctx* ctx);

/* SplitSolution::integrate (src/core/split_solution.cpp:58-90) on every grid point, as the
 * updatePrimal half of DirectMultipleShooting::integrateSolution (direct_multiple_shooting.cpp:212-241)
 * does after rtoc_expand: RTOC_BUF_SOL += primal step (RTOC_BUF_STEP) x RTOC_BUF_DIR for v, a (dv on
 * impact grids), u, lmd, gmm, beta, nu_passive, f, mu, xi, and q on the configuration manifold
 * (Robot::integrateConfiguration): joints additively, the 7 entries [x y z qx qy qz qw] of a free-flyer base
 * (dims.np == 6) by the SE(3) exponential of step x dq[0..5], quaternion re-normalised. */
int rtoc_integrate_solution(rtoc_ctx* ctx);

/* One Newton / SQP iteration of the whole batch as ONE launch sequence without host synchronisation
 * (OCPSolver::updateSolution, src/solver/ocp_solver.cpp:111-145, downstream of the linearisation):
 * KKT error of the freshly linearised records -> rtoc_condense -> rtoc_riccati_sweep -> rtoc_expand(tau)
 * (directions + fraction-to-boundary step sizes, resident in RTOC_BUF_STEP) -> instances whose
 * KKT error (rtoc_kkt_error, i.e. OCPSolver::KKTError()) < kkt_tol get step sizes 0 and keep their iterate (the convergence test of
 * ocp_solver.cpp:200,206 per instance, on the device) -> rtoc_update -> rtoc_integrate_solution (if
 * RTOC_BUF_SOL exists).  Needs dx0 (RTOC_BUF_DX0) like rtoc_riccati_forward.  The caller re-linearises
 * (CPU side) and uploads before the next iteration. */
int rtoc_newton_iteration(rtoc_ctx* ctx, double kkt_tol, double fraction_to_boundary_rule);
/* How often RTOC_OPT_GRAPH has replayed a captured hipGraph (rtoc_riccati_sweep + rtoc_newton_iteration) on this
 * context: diagnostics, and what the tests assert instead of assuming a replay happened. */
int rtoc_graph_replay_count(rtoc_ctx* ctx, unsigned long long* out);
/* Number of instances the last rtoc_newton_iteration found converged (synchronises). */
int rtoc_converged_count(rtoc_ctx* ctx, int* host_count);

/* ---- stage dump / replay ----------------------------------------
 * A self-describing file of everything a context holds at the evalKKT boundary
 * (IntermediateStage::evalKKT outputs, src/ocp/intermediate_stage.cpp:113-148): dims, grid, box rows,
 * cone set-up and the selected buffers, so that stage data recorded on a Pinocchio-equipped machine can
 * drive the kernels elsewhere.  Layout (little endian): rtoc_dump_header, rtoc_grid[nstages],
 * rtoc_box_row[nrows], then for every buffer with count != 0 its first `count` doubles
 * ([batch][nstages][stride], the part the kernels index).  robotoc_amd/replay.py reads / writes the
 * same format in numpy. */
typedef struct rtoc_dump_header {
  char magic[8];             /* "RTOCDMP1" */
  unsigned int version;      /* 1 */
  unsigned int header_bytes; /* sizeof(rtoc_dump_header) */
  rtoc_dims dims;
  int nstages, batch, nrows, cone_contacts, cone_dim, cone_rows /* 0|5 friction, 17 wrench */, reserved[2];
  unsigned long long count[16]; /* doubles stored per RTOC_BUF_* (0: absent) */
} rtoc_dump_header;
/* buffer_mask: bit b set = store RTOC_BUF_b if it exists. */
int rtoc_save_stage_dump(rtoc_ctx* ctx, const char* path, unsigned int buffer_mask);
/* Creates a context sized for the dump (max_stages = nstages), restores grid / rows / cones / buffers. */
int rtoc_load_stage_dump(const char* path, int device, rtoc_ctx** out);

/* Multi-GPU: all-gather the direction buffers of all ranks over RCCL.
 * `nccl_comm` is an ncclComm_t passed as void*; `out` is device memory of
 * world_size * rtoc_buffer_count(ctx, RTOC_BUF_DIR) doubles. */
int rtoc_gather_directions(rtoc_ctx* ctx, void* nccl_comm, double* out);

const char* rtoc_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* RTOC_H_ */

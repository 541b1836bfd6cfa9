/*
 * rtoc_oracle_bench.c -- timed drivers of the CPU oracle for bench.py's `cpu_baseline` leg
 * (TEST INFRASTRUCTURE ONLY, see rtoc_oracle.c for the rules and the parity status).
 *
 * The oracle's backward recursion mutates Qxx, Qxu, Quu, lu in place like the reference
 * (riccati_factorizer_test.cpp:65-66), so a repeated measurement has to start every sweep from fresh
 * records.  Here every OpenMP thread owns a PRIVATE copy of one instance's records (1.5 MB at ANYmal
 * size: cache resident) that it refills from the shared read-only inputs inside the parallel region --
 * the role DirectMultipleShooting::evalKKT's writes play in the reference (every iteration rewrites the
 * stage data, direct_multiple_shooting.cpp:129-159) -- and the time spent in that refill is measured
 * per thread and reported separately, so the baseline can be quoted with and without it.
 *
 *   orc_bench_sweep   RiccatiRecursion::backward + forward of `reps` x `batch` instances
 *                     (src/riccati/riccati_recursion.cpp:32-131)
 *   orc_bench_sqp     the hot path of OCPSolver::updateSolution downstream of the linearisation
 *                     (src/solver/ocp_solver.cpp:118-142): condenseSlackAndDual (box + friction-cone rows),
 *                     condenseContact/ImpactDynamics, Riccati sweep, expansions, step sizes, slack/dual update
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rtoc.h"

unsigned orc_riccati_backward(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt, double* ric,
                              double max_dts0);
void orc_riccati_forward(const rtoc_layout* L, const rtoc_grid* grid, int nstages, double* kkt, double* ric,
                         double* dir);
unsigned orc_condense_stage(const rtoc_layout* L, const rtoc_grid* g, double* kkt_rec, double* cdd_rec, double damping);
unsigned orc_condense_impact_stage(const rtoc_layout* L, const rtoc_grid* g, double* kkt_rec, double* cdd_rec,
                                   double damping);
void orc_expand_stage(const rtoc_layout* L, const rtoc_grid* g, double* cdd_rec, double* dir_rec, double* dir_next_rec);
void orc_pdipm_condense_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows, int nrows,
                              double* kkt_rec, double* con_rec, double* cdd_rec);
void orc_pdipm_expand_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows, int nrows,
                            const double* dir_rec, double* con_rec, double tau, double* steps);
void orc_pdipm_update_stage(const rtoc_layout* L, const rtoc_grid* g, const rtoc_box_row* rows, int nrows,
                            double* con_rec, double primal_step, double dual_step);
void orc_cone_condense_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                             const double* cone_rec, double* kkt_rec, double* cdd_rec, double* con_rec);
void orc_cone_expand_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                           const double* cone_rec, const double* dir_rec, double* con_rec, double tau, double* steps);
void orc_cone_update_stage(const rtoc_layout* L, const rtoc_grid* g, int max_contacts, int contact_dim,
                           double* con_rec, double primal_step, double dual_step);

/* out[0] = wall seconds of the parallel region, out[1] = mean seconds per thread spent refilling the private
 * records, out[2] = threads used, out[3] = checksum (keeps the work alive).  Returns OR of the status bits. */
unsigned orc_bench_sweep(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch, const double* kkt,
                         const double* dx0, double max_dts0, int reps, int nthreads, double* out) {
  const size_t ks = (size_t)nstages * L->kkt.stride, rs = (size_t)nstages * L->ric.stride,
               ds = (size_t)nstages * L->dir.stride;
  unsigned status = 0;
  double copy_total = 0.0, checksum = 0.0;
  int used = 1;
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const double t0 = omp_get_wtime();
#pragma omp parallel reduction(| : status) reduction(+ : copy_total, checksum)
  {
    double* k = (double*)malloc(ks * sizeof(double));
    double* r = (double*)calloc(rs, sizeof(double));
    double* d = (double*)calloc(ds, sizeof(double));
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1) collapse(2)
    for (int rep = 0; rep < reps; ++rep)
      for (int b = 0; b < batch; ++b) {
        const double c0 = omp_get_wtime();
        memcpy(k, kkt + b * ks, ks * sizeof(double));
        copy_total += omp_get_wtime() - c0;
        status |= orc_riccati_backward(L, grid, nstages, k, r, max_dts0);
        memcpy(d + L->dir.off[RTOC_DIR_DX], dx0 + (size_t)b * L->nx, sizeof(double) * L->nx);
        orc_riccati_forward(L, grid, nstages, k, r, d);
        checksum += d[(size_t)(nstages - 1) * L->dir.stride + L->dir.off[RTOC_DIR_DX]];
      }
    free(k);
    free(r);
    free(d);
  }
  out[0] = omp_get_wtime() - t0;
  out[1] = copy_total / used;
  out[2] = used;
  out[3] = checksum;
  return status;
}

unsigned orc_bench_sqp(const rtoc_layout* L, const rtoc_grid* grid, int nstages, int batch, const double* kkt,
                       const double* cdd, const double* con, const double* cone, const double* dx0,
                       const rtoc_box_row* rows, int nrows, int max_contacts, int contact_dim, double tau,
                       double max_dts0, int reps, int nthreads, double* out) {
  const size_t ks = (size_t)nstages * L->kkt.stride, rs = (size_t)nstages * L->ric.stride,
               ds = (size_t)nstages * L->dir.stride, cs = (size_t)nstages * L->cdd.stride,
               ns = (size_t)nstages * L->con.stride,
               es = (size_t)nstages * (max_contacts > 0 ? rtoc_cone_stride(L->dims.nv, max_contacts) : 0);
  unsigned status = 0;
  double copy_total = 0.0, checksum = 0.0;
  int used = 1;
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const double t0 = omp_get_wtime();
#pragma omp parallel reduction(| : status) reduction(+ : copy_total, checksum)
  {
    double* k = (double*)malloc(ks * sizeof(double));
    double* c = (double*)malloc(cs * sizeof(double));
    double* n = (double*)malloc(ns * sizeof(double));
    double* r = (double*)calloc(rs, sizeof(double));
    double* d = (double*)calloc(ds, sizeof(double));
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1) collapse(2)
    for (int rep = 0; rep < reps; ++rep)
      for (int b = 0; b < batch; ++b) {
        const double c0 = omp_get_wtime();
        memcpy(k, kkt + b * ks, ks * sizeof(double));
        memcpy(c, cdd + b * cs, cs * sizeof(double));
        memcpy(n, con + b * ns, ns * sizeof(double));
        copy_total += omp_get_wtime() - c0;
        const double* e = cone ? cone + b * es : 0;
        const size_t est = nstages ? es / nstages : 0;
        /* DirectMultipleShooting::evalKKT tail, stage by stage (intermediate_stage.cpp:134-148) */
        for (int i = 0; i < nstages - 1; ++i) {
          double* kr = k + (size_t)i * L->kkt.stride;
          double* cr = c + (size_t)i * L->cdd.stride;
          double* nr = n + (size_t)i * L->con.stride;
          if (nrows > 0) orc_pdipm_condense_stage(L, &grid[i], rows, nrows, kr, nr, cr);
          if (e) orc_cone_condense_stage(L, &grid[i], max_contacts, contact_dim, e + i * est, kr, cr, nr);
          status |= grid[i].type == RTOC_GRID_IMPACT ? orc_condense_impact_stage(L, &grid[i], kr, cr, 0.0)
                                                     : orc_condense_stage(L, &grid[i], kr, cr, 0.0);
        }
        status |= orc_riccati_backward(L, grid, nstages, k, r, max_dts0);
        memcpy(d + L->dir.off[RTOC_DIR_DX], dx0 + (size_t)b * L->nx, sizeof(double) * L->nx);
        orc_riccati_forward(L, grid, nstages, k, r, d);
        double steps[2] = {1.0, 1.0};
        for (int i = 0; i < nstages - 1; ++i) {
          double* cr = c + (size_t)i * L->cdd.stride;
          double* nr = n + (size_t)i * L->con.stride;
          double* dr = d + (size_t)i * L->dir.stride;
          orc_expand_stage(L, &grid[i], cr, dr, dr + L->dir.stride);
          if (nrows > 0) orc_pdipm_expand_stage(L, &grid[i], rows, nrows, dr, nr, tau, steps);
          if (e) orc_cone_expand_stage(L, &grid[i], max_contacts, contact_dim, e + i * est, dr, nr, tau, steps);
        }
        for (int i = 0; i < nstages - 1; ++i) {
          double* nr = n + (size_t)i * L->con.stride;
          if (nrows > 0) orc_pdipm_update_stage(L, &grid[i], rows, nrows, nr, steps[0], steps[1]);
          if (e) orc_cone_update_stage(L, &grid[i], max_contacts, contact_dim, nr, steps[0], steps[1]);
        }
        checksum += steps[0] + steps[1] + d[(size_t)(nstages - 1) * L->dir.stride + L->dir.off[RTOC_DIR_DX]];
      }
    free(k);
    free(c);
    free(n);
    free(r);
    free(d);
  }
  out[0] = omp_get_wtime() - t0;
  out[1] = copy_total / used;
  out[2] = used;
  out[3] = checksum;
  return status;
}

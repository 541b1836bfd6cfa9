// rtoc_capi.hip -- implementation of the C ABI declared in include/rtoc.h.
// Context, HBM buffers, kernel dispatch by problem dimensions.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtoc.h"
#include "../../include/rtoc_robot.h"
#include "kernel_set.hpp"
#include "rigid_body.hpp"
#include "unconstr_constraints.hpp"
#include "state_equation_lin.hpp"
#include "switching_constraint_lin.hpp"
#include "contact_constraints.hpp"
#include "contact_eval_kkt.hpp"
#include "sto.hpp"

using namespace rtoc;

#define HIP_TRY(expr)                       \
  do {                                      \
    hipError_t e_ = (expr);                 \
    if (e_ != hipSuccess) {                 \
      ctx_set_err(e_, __LINE__);            \
      return RTOC_ERR_HIP;                  \
    }                                       \
  } while (0)

static thread_local char g_errbuf[256] = "";
static void ctx_set_err(hipError_t e, int line) {
  snprintf(g_errbuf, sizeof(g_errbuf), "HIP error %d (%s) at rtoc_capi.hip:%d", (int)e,
           hipGetErrorString(e), line);
}

// ---- kernel table: one entry per compiled robot shape (shape_inst.hip; SHAPES in the Makefile) ------------
#define RTOC_SHAPE(nv, nu, ns, nw0, nw1) namespace rtoc { KernelSet rtoc_shape_##nv##_##nu##_##ns(); }
#include "shape_table.inc"
#undef RTOC_SHAPE
static const std::vector<KernelSet>& kernel_table() {
  static std::vector<KernelSet> t = {
#define RTOC_SHAPE(nv, nu, ns, nw0, nw1) rtoc::rtoc_shape_##nv##_##nu##_##ns(),
#include "shape_table.inc"
#undef RTOC_SHAPE
  };
  return t;
}

// Shapes beyond the compiled-in table: librtoc_shape_<nv>_<nu>_<ns>.so next to this library (or in $RTOC_SHAPE_DIR),
// built by `make -C robotoc_amd/csrc plugin SHAPE=nv:nu:ns:nw0:nw1`; with RTOC_SHAPE_JIT=1 rtoc_create builds it itself
// (hipcc + the source directory this library was built from must be present; ~1 min, once).
#include <mutex>
static std::vector<KernelSet>& plugin_table() {
  static std::vector<KernelSet> t;
  return t;
}
static std::string library_dir() {
  Dl_info info;
  if (dladdr((const void*)&library_dir, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
  }
  return ".";
}
static const KernelSet* load_plugin(const rtoc_dims* d) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& k : plugin_table())
    if (k.nv == d->nv && k.nu == d->nu && k.ns == d->ns_max) return &k;
  char name[96];
  snprintf(name, sizeof name, "librtoc_shape_%d_%d_%d.so", d->nv, d->nu, d->ns_max);
  std::vector<std::string> dirs;
  if (const char* e = getenv("RTOC_SHAPE_DIR")) dirs.push_back(e);
  dirs.push_back(library_dir());
  void* h = nullptr;
  for (const auto& dir : dirs)
    if ((h = dlopen((dir + "/" + name).c_str(), RTLD_NOW | RTLD_LOCAL))) break;
#ifdef RTOC_CSRC_DIR
  const char* jit = getenv("RTOC_SHAPE_JIT");
  if (!h && jit && jit[0] == '1') {
    // tile-split wave counts by state dimension, like the compiled-in shapes: one / three waves up to 36, four beyond
    const int nx = 2 * d->nv, nw0 = nx <= 36 ? 1 : 4, nw1 = nx <= 36 ? 3 : (nx > 64 ? 5 : 4);
    char cmd[1024];
    snprintf(cmd, sizeof cmd, "make -s -C '%s' plugin SHAPE=%d:%d:%d:%d:%d >/dev/null 2>&1", RTOC_CSRC_DIR, d->nv, d->nu, d->ns_max, nw0, nw1);
    if (system(cmd) == 0) h = dlopen((std::string(RTOC_CSRC_DIR) + "/../" + name).c_str(), RTLD_NOW | RTLD_LOCAL);
  }
#endif
  if (!h) return nullptr;
  typedef int (*entry_t)(KernelSet*, size_t, size_t);
  entry_t entry = (entry_t)dlsym(h, "rtoc_shape_plugin");
  KernelSet k;
  if (!entry || entry(&k, sizeof(KernelSet), kernel_abi_stamp()) != 0 || k.nv != d->nv || k.nu != d->nu || k.ns != d->ns_max) {
    dlclose(h);
    return nullptr;
  }
  plugin_table().reserve(64);  // handed-out pointers stay valid
  if (plugin_table().size() >= 64) return nullptr;
  plugin_table().push_back(k);
  return &plugin_table().back();
}

static const KernelSet* find_set(const rtoc_dims* d) {
  if (d->nf_max != d->ns_max || d->np != d->nv - d->nu) return nullptr;
  for (const auto& k : kernel_table())
    if (k.nv == d->nv && k.nu == d->nu && k.ns == d->ns_max) return &k;
  return load_plugin(d);
}

static bool model_has_surface_contacts(const rtoc_robot_model& m) {
  for (int k = 0; k < m.ncontacts; ++k)
    if (m.contact_type[k] == RTOC_CONTACT_SURFACE) return true;
  return false;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is per function and process-wide, not per context: two live contexts with
// different models (iCub: 11 tree levels, ANYmal: 4) share it, so it only ever grows (the launch passes its own size)
static hipError_t set_linearize_lds(const rtoc_robot_model& m, int nlevels, int nbranch, int dpp) {
  static std::mutex mu;
  static int max_bytes_of[64] = {};   // the attribute is per DEVICE: one running maximum for each (the current one: callers hipSetDevice first)
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int& max_bytes = max_bytes_of[dev];
  int bytes = (int)rbd::lin_lds_bytes(nlevels, nbranch, m.njoints, m.ncontacts, m.nv, dpp, false);   // the larger of the two modes
  if (bytes <= max_bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)rbd::linearize_contact_dynamics_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)rbd::linearize_contact_dynamics_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)rbd::linearize_contact_dynamics_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)rbd::linearize_contact_dynamics_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)rbd::rbd_values_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * rbd::VAL_SLOTS * (int)sizeof(double));
  if (e == hipSuccess) max_bytes = bytes;
  return e;
}

// ---- context --------------------------------------------------------------------------
#define RTOC_MAX_CHUNK_EVENTS 16
struct rtoc_ctx {
  rtoc_dims dims;
  rtoc_layout L;
  const KernelSet* ks;
  int max_stages, nstages, batch, device;
  hipStream_t own_stream, stream;
  double* buf[RTOC_NUM_BUFFERS];
  size_t count[RTOC_NUM_BUFFERS];
  bool owned[RTOC_NUM_BUFFERS];
  bool kkt_exposed;    // rtoc_device_ptr(RTOC_BUF_KKT) was handed out: the caller can rewrite the records without the runtime seeing it
  bool capturing;      // inside run_graphed's stream capture (nothing that synchronises may run)
  rtoc_grid* d_grid;
  rtoc_box_row* d_rows;
  rtoc_box_row* h_rows;  // host copies (stage dump)
  rtoc_grid* h_grid;
  int* d_nconv;  // instances found converged by the last rtoc_newton_iteration
  int* d_pair;   // first two rows of every primal entry, packed (int4 per entry)
  int* d_entry;  // CSR over the primal entries (q_0..,v_0..,u_0..): [ne+1] offsets, then [nrows] row ids
  int nrows;
  uint32_t* d_status;
  long long* d_prof;
  int writeback;
  double max_dts0;
  double contact_inv_damping;
  int bwd_variant;
  hipEvent_t ev0, ev1;
  hipStream_t stream2;  // forward half of the pipelined sweep
  hipEvent_t ev_fork, ev_join, ev_chunk[RTOC_MAX_CHUNK_EVENTS];
  int sweep_chunks;
  int condense_split;  // 1: MJtJinv in its own kernel ahead of the condensation
  int keep_qaf;        // RTOC_OPT_CONDENSE_KEEP_QAF
  int fxx_mode;        // RTOC_OPT_FXX_STRUCTURE: 0 auto, 1 dense, 2 caller asserts the structure
  int bwd_register;    // RTOC_OPT_BACKWARD_REGISTER: the register-resident backward kernel where it applies
  int num_cus;         // compute units of the device (the register-wide iCub kernel runs where the batch fills them)
  int cond_register;   // RTOC_OPT_CONDENSE_REGISTER: the register-chained condensation of the contact grid points where it applies
  int* d_stage_list;   // [max_stages] grid points 0 .. nstages - 2: the contact ones first (n_stage_contact), then the impact ones
  int n_stage_contact, n_stage_impact;
  int fxx_state;       // auto mode cache: 0 unknown (re-check before the next backward recursion), 1 every Fxx structured, 2 not
  int fxx_last;        // the last check's answer (1 / 2; 0 never checked): the kernel choice baked into captured graphs
  unsigned long long graph_replays;  // hipGraphLaunch count of RTOC_OPT_GRAPH (rtoc_graph_replay_count)
  int* d_fxx_flag;
  double* d_sto;       // rtoc_sto_eval_kkt staging: lt, diag(Qtt), squared error
  // RTOC_OPT_GRAPH: launch sequences replayed from captured hipGraphs
  int use_graph;
  int exact_transport;  // RTOC_OPT_SWITCHING_TRANSPORT
  int unconstr_dense;   // RTOC_OPT_UNCONSTR_DENSE
  int exact_cone_jacobian;  // RTOC_OPT_CONE_JACOBIAN
  int impact_cones;     // RTOC_OPT_IMPACT_CONES (default 1, rtoc_create)
  double* d_mu;         // rtoc_set_friction_coefficients
  int n_mu;             // how many of its RTOC_MAX_CONTACTS entries the caller set
  double* d_wcone;      // rtoc_set_wrench_cone_params: [RTOC_MAX_CONTACTS][17 x 6]
  double *d_vals, *d_vals2;  // rbd_values_kernel -> linearize_contact_dynamics_kernel<.., PRE>: [batch * max_stages][njoints][64]
  size_t vals_cap;
  int vals_fresh;       // the values in d_vals belong to the iterate in RTOC_BUF_SOL (consumed by the next launch_linearize)
  int linearize_fused;  // RTOC_OPT_LINEARIZE_FUSED
  int lin_dpp;          // RTOC_OPT_LINEARIZE_DOFS_PER_PASS (0 = per model)
  unsigned long long epoch;  // bumped by everything that changes a launch parameter baked into a captured graph
  struct GraphSlot {
    hipGraphExec_t exec;
    unsigned long long epoch, warm_epoch;
    double p0, p1;
    bool warm;
  } g_sweep, g_newton;
  size_t sto_cap;
  int cone_contacts, cone_dim;  // friction / wrench cones: max contacts (0 = off), force components per contact
  int cone_rows;                // PDIPM rows per contact: 5 friction cone, 17 contact wrench cone
  double* d_kkterr;             // [batch]
  int backward_scan;            // RTOC_OPT_BACKWARD_SCAN
  double* d_scan[3];            // element ping-pong buffers, value records (allocated on first use)
  double* d_scan_sto;           // riccati_scan_sto.hpp: per grid point At, P+ Fx, P+ fx, factors of G
  // rigid-body model (rtoc_set_robot_model) and contact schedule (rtoc_set_contact_schedule)
  rbd::DevModel* d_model;
  rbd::DevModel* h_model;
  unsigned* d_active;
  double* d_cpos;
  double* d_crot;
  bool has_cpos, has_crot;
  // rtoc_line_search_filter: filters [batch][CAP][2], sizes [batch], staging (cost, violation | mask, accepted)
  double* d_cost;      // rtoc_set_configuration_cost: 9 nv doubles
  double* d_bounds;    // rtoc_set_constraint_bounds: [nrows]
  double barrier, ftb_rule;
  double* d_x0;        // rtoc_set_initial_state: [batch][2 nv]
  double* d_filter;
  int* d_nfilter;
  double* d_ls_in;
  int* d_ls_flags;
  // switching-time optimisation on the device (rtoc_sto_set_problem; sto.hpp)
  int sto_on, sto_nev;
  double sto_t0, sto_T, sto_barrier, sto_tau, sto_reg;
  double* d_ts;         // [batch][nev] event times of every instance
  double* d_dt;         // [batch][max_stages] time steps of every instance (grid_dt)
  double* d_sto_con;    // [batch][RTOC_STO_CON_STRIDE] dwell-time rows
  double* d_min_dwell;  // [RTOC_STO_MAX_EVENTS + 1]
  double* d_sto_cost;   // [2][batch][nev] STO cost gradient / Hessian diagonal handed over by the host, or nullptr
  double* d_sto_out;    // [2][batch][nev] + [batch]: lt, Qtt diagonal as scattered, squared STO KKT term
  double* d_costval;    // [batch][max_stages] cost values of the last rtoc_contact_eval_kkt (rtoc_contact_eval_ocp)
  // filter line search on the device (rtoc_set_line_search, rtoc_contact_line_search)
  int ls_on;
  double ls_rate, ls_min_step, ls_cost_rate, ls_viol_rate;
  int ls_method;        // 0 LineSearchMethod::Filter, 1 MeritBacktracking (rtoc_set_line_search_method)
  double ls_armijo, ls_margin, ls_eps;
  double* d_ls_merit;   // [batch] penalty parameter + [batch] directional derivative
  double ls_unconstr_dt;   // > 0: the last evalKKT was rtoc_unconstr_eval_kkt(dt) -- trial iterates of the line search are evaluated by it
  double* d_eval;       // [2][2][batch]: (cost + barrier | violation) of the current iterate, of the trial iterate
  double* d_eval_part;  // [batch][max_stages][2]
  double* d_sol_trial;  // trial iterate: SplitSolution records, constraint records, steps
  double* d_con_trial;
  double* d_ls_steps;   // [batch][2] trial steps + [batch] alpha
  int* d_ls_active;     // [batch] active flags + [1] counter
  int ls_trials;        // trial evaluations of the last line search
};

extern "C" {

int rtoc_version(void) { return 100; }

int rtoc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- streaming kernels of rtoc_bandwidth_probe: chunk = (block-wave) + k * (waves in the launch), 1 KB per wave-instruction ----
extern "C++" {
template <int U, bool COPY>
static __global__ __launch_bounds__(256) void stream_probe_kernel(const char* __restrict__ src, char* __restrict__ dst, size_t chunks,
                                                                  double* sink) {
  const size_t nwaves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (size_t c = w; c + (U - 1) * nwaves < chunks; c += U * nwaves) {
    d2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = *reinterpret_cast<const d2*>(src + (c + k * nwaves) * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (COPY) *reinterpret_cast<d2*>(dst + (c + k * nwaves) * 1024 + lane * 16) = v[k];
      else acc += v[k][0] + v[k][1];
    }
  }
  if (!COPY && acc == 1234.5) sink[w] = acc;
}
}  // extern "C++"

int rtoc_bandwidth_probe(int device, size_t bytes, double* read_gbs, double* copy_gbs) {
  constexpr int blocks = 256 * 64;                             // 64 workgroups of 4 waves per CU: 6.2 TB/s read (256 * 8: 5.7)
  constexpr size_t per = (size_t)blocks * 4 * 8;               // chunks consumed per trip of all waves (U = 8)
  static_assert(per * 1024 == RTOC_BANDWIDTH_PROBE_MIN_BYTES, "include/rtoc.h states the probe's minimum size: one trip of every wave");
  if (bytes < per * 1024) return RTOC_ERR_BAD_ARG;             // RTOC_BANDWIDTH_PROBE_MIN_BYTES: one trip of every wave (512 MiB)
  HIP_TRY(hipSetDevice(device));
  const size_t chunks = (bytes / 1024) / per * per;
  char *src = nullptr, *dst = nullptr;
  double* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&src, chunks * 1024);
  if (e == hipSuccess) e = hipMalloc((void**)&dst, chunks * 1024);
  if (e == hipSuccess) e = hipMalloc((void**)&sink, sizeof(double) * blocks * 4);
  if (e == hipSuccess) e = hipMemset(src, 0, chunks * 1024);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  double best[2] = {0.0, 0.0};
  for (int mode = 0; mode < 2 && e == hipSuccess; ++mode)
    for (int rep = 0; rep < 6 && e == hipSuccess; ++rep) {   // the first launch warms up
      (void)hipEventRecord(e0, nullptr);
      if (mode == 0) hipLaunchKernelGGL((stream_probe_kernel<8, false>), dim3(blocks), dim3(256), 0, nullptr, src, dst, chunks, sink);
      else hipLaunchKernelGGL((stream_probe_kernel<8, true>), dim3(blocks), dim3(256), 0, nullptr, src, dst, chunks, sink);
      (void)hipEventRecord(e1, nullptr);
      e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      const double gbs = (mode + 1) * (double)chunks * 1024.0 / (ms * 1e-3) / 1e9;
      if (rep > 0 && gbs > best[mode]) best[mode] = gbs;
    }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  if (sink) (void)hipFree(sink);
  if (e != hipSuccess) {
    ctx_set_err(e, __LINE__);
    return RTOC_ERR_HIP;
  }
  if (read_gbs) *read_gbs = best[0];
  if (copy_gbs) *copy_gbs = best[1];
  return RTOC_OK;
}

int rtoc_dims_supported(const rtoc_dims* dims) { return dims && find_set(dims) ? 1 : 0; }

void rtoc_layout_for_dims(const rtoc_dims* dims, rtoc_layout* out) { rtoc_compute_layout(dims, out); }

const char* rtoc_error_string(int code) {
  switch (code) {
    case RTOC_OK: return "ok";
    case RTOC_ERR_BAD_ARG: return "bad argument";
    case RTOC_ERR_UNSUPPORTED_DIMS: return "no kernel specialisation for these dimensions";
    case RTOC_ERR_NO_DEVICE: return "no HIP device (the HIP path has no CPU fallback)";
    case RTOC_ERR_HIP: return g_errbuf;
    case RTOC_ERR_NOT_READY: return "grid not set";
    case RTOC_ERR_RCCL: return "RCCL error";
    case RTOC_ERR_IO: return "stage dump: file error or malformed file";
    default: return "unknown";
  }
}

int rtoc_destroy(rtoc_ctx* c);
// everything of rtoc_create that can fail after the context object exists; the caller destroys the
// half-built context on failure (rtoc_destroy tolerates members that were never created)
static int create_members(rtoc_ctx* c, const rtoc_dims* dims, const KernelSet* ks, int max_stages, int batch, int device) {
  c->dims = *dims;
  rtoc_compute_layout(dims, &c->L);
  // the backward kernels carry their record offsets as immediates: they must be the ones the host
  // (and the caller, through rtoc_get_layout) uses
  if (memcmp(&ks->kl, &c->L.kkt, sizeof(rtoc_record_layout)) != 0 ||
      memcmp(&ks->rl, &c->L.ric, sizeof(rtoc_record_layout)) != 0 ||
      memcmp(&ks->dl, &c->L.dir, sizeof(rtoc_record_layout)) != 0 ||
      memcmp(&ks->cl, &c->L.cdd, sizeof(rtoc_record_layout)) != 0) {
    return RTOC_ERR_BAD_ARG;
  }
  c->ks = ks;
  c->max_stages = max_stages;
  c->nstages = 0;
  c->batch = batch;
  c->device = device;
  c->max_dts0 = 0.1;  // RiccatiRecursion(ocp, max_dts0 = 0.1), riccati_recursion.hpp:35
  c->bwd_variant = (ks->nvariants >= 3) ? ks->nvariants - 1 : 0;  // role-split kernel where it exists
  c->bwd_register = 1;   // ... and the register-resident kernel wherever it applies (rv_applies, rw_applies)
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    c->num_cus = cus;
  }
  c->cond_register = 1;  // likewise the condensation (cond_rv_applies)
  if (const char* e = getenv("RTOC_CONDENSE_REGISTER")) c->cond_register = (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
  HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  for (int i = 0; i < RTOC_MAX_CHUNK_EVENTS; ++i)
    HIP_TRY(hipEventCreateWithFlags(&c->ev_chunk[i], hipEventDisableTiming));
  c->condense_split = c->ks->cond_fused_default ? 0 : 1;   // per robot shape (rtoc.h: RTOC_OPT_CONDENSE_SPLIT)
  // RTOC_CONDENSE_SPLIT=0|1 in the environment: default of RTOC_OPT_CONDENSE_SPLIT for contexts created afterwards (runs the
  // whole test suite / bench on the other condensation pipeline without touching the callers)
  if (const char* e = getenv("RTOC_CONDENSE_SPLIT")) c->condense_split = (e[0] == '0') ? 0 : 1;
  c->sweep_chunks = 1;  // measured on MI355X: chunked pipelining does not pay (forward waves do not fit next to the backward waves)
  const size_t per = (size_t)batch * max_stages;
  c->count[RTOC_BUF_KKT] = per * c->L.kkt.stride;
  c->count[RTOC_BUF_RIC] = per * c->L.ric.stride;
  c->count[RTOC_BUF_DIR] = per * c->L.dir.stride;
  c->count[RTOC_BUF_CDD] = per * c->L.cdd.stride;
  c->count[RTOC_BUF_CON] = per * (size_t)c->L.con.stride;
  c->count[RTOC_BUF_DX0] = (size_t)batch * c->L.nx;
  c->count[RTOC_BUF_STEP] = (size_t)batch * 2;
  c->count[RTOC_BUF_SE3] = per * RTOC_SE3_STRIDE;
  c->count[RTOC_BUF_CONE] = 0;  // sized by rtoc_set_friction_cones
  c->count[RTOC_BUF_SOL] = per * c->L.sol.stride;
  for (int i = 0; i < RTOC_NUM_BUFFERS; ++i) {
    // the CDD / CON buffers are large; they are allocated lazily on first use (upload / bind / condense)
    c->buf[i] = nullptr;
    c->owned[i] = false;
  }
  const int eager[] = {RTOC_BUF_KKT, RTOC_BUF_RIC, RTOC_BUF_DIR, RTOC_BUF_DX0, RTOC_BUF_STEP};
  for (int i : eager) {
    HIP_TRY(hipMalloc((void**)&c->buf[i], c->count[i] * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->buf[i], 0, c->count[i] * sizeof(double), c->stream));
    c->owned[i] = true;
  }
  HIP_TRY(hipMalloc((void**)&c->d_grid, sizeof(rtoc_grid) * max_stages));
  HIP_TRY(hipMalloc((void**)&c->d_stage_list, sizeof(int) * max_stages));
  HIP_TRY(hipMalloc((void**)&c->d_status, sizeof(uint32_t) * batch));
  HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t) * batch, c->stream));
  for (int v = 0; v < ks->nvariants; ++v)
    HIP_TRY(hipFuncSetAttribute((const void*)ks->bwd[v], hipFuncAttributeMaxDynamicSharedMemorySize,
                                ks->bwd_lds[v]));
  if (ks->bwd_sa)
    HIP_TRY(hipFuncSetAttribute((const void*)ks->bwd_sa, hipFuncAttributeMaxDynamicSharedMemorySize, ks->bwd_lds[3]));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->cond, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ks->cond_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->cond_split, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ks->cond_split_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->mjt, hipFuncAttributeMaxDynamicSharedMemorySize, ks->mjt_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->expd, hipFuncAttributeMaxDynamicSharedMemorySize, ks->expd_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->scan_elt, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ks->scan_elt_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->scan_comb, hipFuncAttributeMaxDynamicSharedMemorySize,
                              ks->scan_comb_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->fscan_elt, hipFuncAttributeMaxDynamicSharedMemorySize, ks->fscan_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->fscan_comb, hipFuncAttributeMaxDynamicSharedMemorySize, ks->fscan_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->sto_prep, hipFuncAttributeMaxDynamicSharedMemorySize, ks->sto_prep_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)ks->sto_vec, hipFuncAttributeMaxDynamicSharedMemorySize, ks->sto_vec_lds));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_create(const rtoc_dims* dims, int max_stages, int batch, int device, rtoc_ctx** out) {
  if (!dims || !out || max_stages < 2 || batch < 1) return RTOC_ERR_BAD_ARG;
  const KernelSet* ks = find_set(dims);
  if (!ks) return RTOC_ERR_UNSUPPORTED_DIMS;
  if (rtoc_device_count() <= device || device < 0) return RTOC_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  rtoc_ctx* c = new (std::nothrow) rtoc_ctx();
  if (!c) return RTOC_ERR_BAD_ARG;
  memset(c, 0, sizeof(*c));
  c->device = device;
  c->impact_cones = 1;
  int rc = create_members(c, dims, ks, max_stages, batch, device);
  if (rc) {
    (void)rtoc_destroy(c);
    return rc;
  }
  *out = c;
  return RTOC_OK;
}

int rtoc_destroy(rtoc_ctx* c) {
  if (!c) return RTOC_ERR_BAD_ARG;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < RTOC_NUM_BUFFERS; ++i)
    if (c->owned[i] && c->buf[i]) (void)hipFree(c->buf[i]);
  if (c->d_grid) (void)hipFree(c->d_grid);
  if (c->d_stage_list) (void)hipFree(c->d_stage_list);
  if (c->d_rows) (void)hipFree(c->d_rows);
  free(c->h_rows);
  free(c->h_grid);
  if (c->d_kkterr) (void)hipFree(c->d_kkterr);
  if (c->d_entry) (void)hipFree(c->d_entry);
  if (c->d_pair) (void)hipFree(c->d_pair);
  if (c->d_nconv) (void)hipFree(c->d_nconv);
  if (c->d_fxx_flag) (void)hipFree(c->d_fxx_flag);
  if (c->d_sto) (void)hipFree(c->d_sto);
  if (c->g_sweep.exec) (void)hipGraphExecDestroy(c->g_sweep.exec);
  if (c->g_newton.exec) (void)hipGraphExecDestroy(c->g_newton.exec);
  if (c->d_status) (void)hipFree(c->d_status);
  if (c->d_model) (void)hipFree(c->d_model);
  delete c->h_model;
  if (c->d_active) (void)hipFree(c->d_active);
  if (c->d_cost) (void)hipFree(c->d_cost);
  if (c->d_bounds) (void)hipFree(c->d_bounds);
  if (c->d_mu) (void)hipFree(c->d_mu);
  if (c->d_wcone) (void)hipFree(c->d_wcone);
  if (c->d_vals) (void)hipFree(c->d_vals);
  if (c->d_vals2) (void)hipFree(c->d_vals2);
  if (c->d_x0) (void)hipFree(c->d_x0);
  if (c->d_filter) (void)hipFree(c->d_filter);
  if (c->d_nfilter) (void)hipFree(c->d_nfilter);
  if (c->d_ls_in) (void)hipFree(c->d_ls_in);
  if (c->d_ls_flags) (void)hipFree(c->d_ls_flags);
  if (c->d_ts) (void)hipFree(c->d_ts);
  if (c->d_dt) (void)hipFree(c->d_dt);
  if (c->d_sto_con) (void)hipFree(c->d_sto_con);
  if (c->d_min_dwell) (void)hipFree(c->d_min_dwell);
  if (c->d_sto_cost) (void)hipFree(c->d_sto_cost);
  if (c->d_sto_out) (void)hipFree(c->d_sto_out);
  if (c->d_costval) (void)hipFree(c->d_costval);
  if (c->d_eval) (void)hipFree(c->d_eval);
  if (c->d_eval_part) (void)hipFree(c->d_eval_part);
  if (c->d_sol_trial) (void)hipFree(c->d_sol_trial);
  if (c->d_con_trial) (void)hipFree(c->d_con_trial);
  if (c->d_ls_steps) (void)hipFree(c->d_ls_steps);
  if (c->d_ls_merit) (void)hipFree(c->d_ls_merit);
  if (c->d_ls_active) (void)hipFree(c->d_ls_active);
  if (c->d_cpos) (void)hipFree(c->d_cpos);
  if (c->d_crot) (void)hipFree(c->d_crot);
  if (c->d_prof) (void)hipFree(c->d_prof);
  for (int i = 0; i < 3; ++i)
    if (c->d_scan[i]) (void)hipFree(c->d_scan[i]);
  if (c->d_scan_sto) (void)hipFree(c->d_scan_sto);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (int i = 0; i < RTOC_MAX_CHUNK_EVENTS; ++i)
    if (c->ev_chunk[i]) (void)hipEventDestroy(c->ev_chunk[i]);
  delete c;
  return RTOC_OK;
}

int rtoc_set_grid(rtoc_ctx* c, const rtoc_grid* grid, int nstages);
int rtoc_set_constraint_rows(rtoc_ctx* c, const rtoc_box_row* rows, int nrows);
int rtoc_set_friction_cones(rtoc_ctx* c, int max_contacts, int contact_dim);
int rtoc_set_wrench_cones(rtoc_ctx* c, int max_contacts);
int rtoc_set_option(rtoc_ctx* c, int option, int64_t value);

int rtoc_clone(rtoc_ctx* c, rtoc_ctx** out) {
  if (!c || !out) return RTOC_ERR_BAD_ARG;
  rtoc_ctx* n = nullptr;
  int rc = rtoc_create(&c->dims, c->max_stages, c->batch, c->device, &n);
  if (rc) return rc;
  if (c->nstages >= 2 && c->h_grid) rc = rtoc_set_grid(n, c->h_grid, c->nstages);
  if (!rc && c->nrows > 0 && c->h_rows) rc = rtoc_set_constraint_rows(n, c->h_rows, c->nrows);
  if (!rc && c->cone_contacts > 0)
    rc = c->cone_rows == RTOC_WRENCH_ROWS ? rtoc_set_wrench_cones(n, c->cone_contacts)
                                          : rtoc_set_friction_cones(n, c->cone_contacts, c->cone_dim);
  if (!rc) {
    n->writeback = c->writeback;
    n->bwd_register = c->bwd_register;
    n->cond_register = c->cond_register;
    n->max_dts0 = c->max_dts0;
    n->contact_inv_damping = c->contact_inv_damping;
    n->bwd_variant = c->bwd_variant;
    n->sweep_chunks = c->sweep_chunks;
    n->condense_split = c->condense_split;
    n->keep_qaf = c->keep_qaf;
    n->fxx_mode = c->fxx_mode;
    n->use_graph = c->use_graph;
    n->exact_transport = c->exact_transport;
    n->impact_cones = c->impact_cones;
    n->unconstr_dense = c->unconstr_dense;
    n->linearize_fused = c->linearize_fused;
    n->lin_dpp = c->lin_dpp;
    n->exact_cone_jacobian = c->exact_cone_jacobian;
    n->ls_on = c->ls_on, n->ls_rate = c->ls_rate, n->ls_min_step = c->ls_min_step, n->ls_cost_rate = c->ls_cost_rate, n->ls_viol_rate = c->ls_viol_rate;
    n->ls_method = c->ls_method, n->ls_armijo = c->ls_armijo, n->ls_margin = c->ls_margin, n->ls_eps = c->ls_eps;
    if (c->backward_scan) rc = rtoc_set_option(n, RTOC_OPT_BACKWARD_SCAN, c->backward_scan);
  }
  hipError_t e = hipStreamSynchronize(c->stream);
  // the rigid-body model, contact schedule, cost, initial states, constraint bounds and line-search filters (rtoc_robot.h)
  auto dup = [&](void** dst, const void* src, size_t bytes) {
    if (!src || rc || e != hipSuccess) return;
    e = hipMalloc(dst, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyDeviceToDevice, n->stream);
  };
  if (!rc && c->h_model) {
    n->h_model = new (std::nothrow) rbd::DevModel(*c->h_model);
    if (!n->h_model) rc = RTOC_ERR_HIP;
    dup((void**)&n->d_model, c->d_model, sizeof(rbd::DevModel));
    if (!rc && e == hipSuccess)
      e = set_linearize_lds(n->h_model->m, n->h_model->nlevels, n->h_model->nbranch, n->h_model->dpp);
  }
  dup((void**)&n->d_active, c->d_active, sizeof(unsigned) * c->max_stages);
  dup((void**)&n->d_cpos, c->d_cpos, sizeof(double) * c->max_stages * RTOC_MAX_CONTACTS * 3);
  dup((void**)&n->d_crot, c->d_crot, sizeof(double) * c->max_stages * RTOC_MAX_CONTACTS * 9);
  n->has_cpos = c->has_cpos, n->has_crot = c->has_crot;
  dup((void**)&n->d_cost, c->d_cost, sizeof(double) * 12 * (c->dims.nv + 1));
  dup((void**)&n->d_x0, c->d_x0, sizeof(double) * c->batch * (2 * c->dims.nv + (c->dims.np == 6 ? 1 : 0)));
  dup((void**)&n->d_bounds, c->d_bounds, sizeof(double) * c->dims.nc_max);
  dup((void**)&n->d_mu, c->d_mu, sizeof(double) * RTOC_MAX_CONTACTS);
  n->n_mu = c->n_mu;
  dup((void**)&n->d_wcone, c->d_wcone, sizeof(double) * RTOC_MAX_CONTACTS * RTOC_WRENCH_ROWS * 6);
  n->barrier = c->barrier, n->ftb_rule = c->ftb_rule;
  if (c->d_filter) {
    dup((void**)&n->d_filter, c->d_filter, sizeof(double) * 2 * RTOC_LINE_SEARCH_FILTER_CAPACITY * c->batch);
    dup((void**)&n->d_nfilter, c->d_nfilter, sizeof(int) * c->batch);
    dup((void**)&n->d_ls_in, c->d_ls_in, sizeof(double) * 2 * c->batch);
    dup((void**)&n->d_ls_flags, c->d_ls_flags, sizeof(int) * 2 * c->batch);
  }
  if (c->sto_on) {
    n->sto_on = 1, n->sto_nev = c->sto_nev, n->sto_t0 = c->sto_t0, n->sto_T = c->sto_T;
    n->sto_barrier = c->sto_barrier, n->sto_tau = c->sto_tau, n->sto_reg = c->sto_reg;
    const size_t ne = (size_t)c->batch * (c->sto_nev > 0 ? c->sto_nev : 1);
    dup((void**)&n->d_ts, c->d_ts, sizeof(double) * ne);
    dup((void**)&n->d_dt, c->d_dt, sizeof(double) * c->batch * c->max_stages);
    dup((void**)&n->d_sto_con, c->d_sto_con, sizeof(double) * c->batch * RTOC_STO_CON_STRIDE);
    dup((void**)&n->d_min_dwell, c->d_min_dwell, sizeof(double) * (RTOC_STO_MAX_EVENTS + 1));
    dup((void**)&n->d_sto_cost, c->d_sto_cost, sizeof(double) * 2 * ne);
    dup((void**)&n->d_sto_out, c->d_sto_out, sizeof(double) * (2 * ne + c->batch));
  }
  for (int b = 0; !rc && e == hipSuccess && b < RTOC_NUM_BUFFERS; ++b) {
    if (!c->buf[b]) continue;
    if (!n->buf[b]) {
      n->count[b] = c->count[b];
      e = hipMalloc((void**)&n->buf[b], n->count[b] * sizeof(double));
      if (e != hipSuccess) break;
      n->owned[b] = true;
    }
    e = hipMemcpyAsync(n->buf[b], c->buf[b], c->count[b] * sizeof(double), hipMemcpyDeviceToDevice, n->stream);
  }
  if (!rc && e == hipSuccess) e = hipMemcpyAsync(n->d_status, c->d_status, sizeof(uint32_t) * c->batch, hipMemcpyDeviceToDevice, n->stream);
  if (!rc && e == hipSuccess) e = hipStreamSynchronize(n->stream);
  if (rc || e != hipSuccess) {
    if (e != hipSuccess) ctx_set_err(e, __LINE__);
    (void)rtoc_destroy(n);
    return rc ? rc : RTOC_ERR_HIP;
  }
  *out = n;
  return RTOC_OK;
}

int rtoc_get_layout(const rtoc_ctx* c, rtoc_layout* out) {
  if (!c || !out) return RTOC_ERR_BAD_ARG;
  *out = c->L;
  return RTOC_OK;
}

int rtoc_set_grid(rtoc_ctx* c, const rtoc_grid* grid, int nstages) {
  if (!c || !grid || nstages < 2 || nstages > c->max_stages) return RTOC_ERR_BAD_ARG;
  if (grid[nstages - 1].type != RTOC_GRID_TERMINAL) return RTOC_ERR_BAD_ARG;
  for (int i = 0; i < nstages; ++i) {
    const rtoc_grid& g = grid[i];
    if (g.dims < 0 || g.dims > c->dims.ns_max || g.dimf < 0 || g.dimf > c->dims.nf_max)
      return RTOC_ERR_BAD_ARG;
    if (i < nstages - 1 && g.type == RTOC_GRID_TERMINAL) return RTOC_ERR_BAD_ARG;
    if (g.type == RTOC_GRID_IMPACT && (i == 0 || i >= nstages - 2)) return RTOC_ERR_BAD_ARG;
    if (g.type == RTOC_GRID_LIFT && i == 0) return RTOC_ERR_BAD_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->d_grid, grid, sizeof(rtoc_grid) * nstages, hipMemcpyHostToDevice, c->stream));
  {
    // the grid points a condensation launch covers, by kind (condense_rv_kernel takes the contact ones, condense_kernel the impact ones)
    std::vector<int> list;
    for (int i = 0; i + 1 < nstages; ++i)
      if (grid[i].type != RTOC_GRID_IMPACT) list.push_back(i);
    c->n_stage_contact = (int)list.size();
    for (int i = 0; i + 1 < nstages; ++i)
      if (grid[i].type == RTOC_GRID_IMPACT) list.push_back(i);
    c->n_stage_impact = (int)list.size() - c->n_stage_contact;
    HIP_TRY(hipMemcpyAsync(c->d_stage_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // (the pageable source dies with this scope)
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!c->h_grid) c->h_grid = (rtoc_grid*)malloc(sizeof(rtoc_grid) * c->max_stages);
  if (c->h_grid) memcpy(c->h_grid, grid, sizeof(rtoc_grid) * nstages);
  c->nstages = nstages;
  c->fxx_state = 0;
  c->epoch++;
  if (c->sto_on) {  // the event times on the device belong to the previous grid structure unless the events are the same
    int nev = 0;
    for (int i = 0; i + 1 < nstages; ++i)
      if (grid[i].type == RTOC_GRID_IMPACT || grid[i].type == RTOC_GRID_LIFT) ++nev;
    if (nev != c->sto_nev) c->sto_on = 0;   // rtoc_sto_set_problem again
  }
  return RTOC_OK;
}

int rtoc_set_stream(rtoc_ctx* c, void* s) {
  if (!c) return RTOC_ERR_BAD_ARG;
  c->epoch++;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return RTOC_OK;
}

static int ensure_scan_buffers(rtoc_ctx* c);
#define RTOC_SCAN_AUTO_MAX_BATCH 8  // measured on MI355X (profiles/r01_scan_batch_crossover.log): the scan wins up to ~16 ANYmal / ~10 iCub instances

// (re)plans the passes of the tangent walk for the context's model and sends the model to the device
static int apply_linearize_plan(rtoc_ctx* c) {
  rbd::DevModel* h = c->h_model;
  const int old_dpp = h->dpp;
  rbd::plan_passes(h, c->lin_dpp);
  if (rbd::lin_lds_bytes(h->nlevels, h->nbranch, h->m.njoints, h->m.ncontacts, h->m.nv, h->dpp, false) > 160 * 1024) {
    rbd::plan_passes(h, old_dpp);
    return RTOC_ERR_BAD_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->d_model, h, sizeof(rbd::DevModel), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(set_linearize_lds(h->m, h->nlevels, h->nbranch, h->dpp));
  return RTOC_OK;
}

int rtoc_get_option(rtoc_ctx* c, int option, int64_t* value) {
  if (!c || !value) return RTOC_ERR_BAD_ARG;
  switch (option) {
    case RTOC_OPT_WRITEBACK_KKT: *value = c->writeback; return RTOC_OK;
    case RTOC_OPT_SWEEP_CHUNKS: *value = c->sweep_chunks; return RTOC_OK;
    case RTOC_OPT_CONDENSE_SPLIT: *value = c->condense_split; return RTOC_OK;
    case RTOC_OPT_BACKWARD_SCAN: *value = c->backward_scan; return RTOC_OK;
    case RTOC_OPT_CONDENSE_KEEP_QAF: *value = c->keep_qaf; return RTOC_OK;
    case RTOC_OPT_FXX_STRUCTURE: *value = c->fxx_mode; return RTOC_OK;
    case RTOC_OPT_GRAPH: *value = c->use_graph; return RTOC_OK;
    case RTOC_OPT_IMPACT_CONES: *value = c->impact_cones; return RTOC_OK;
    case RTOC_OPT_LINEARIZE_DOFS_PER_PASS: *value = c->h_model ? c->h_model->dpp : 0; return RTOC_OK;
    case RTOC_OPT_BACKWARD_REGISTER: *value = c->bwd_register; return RTOC_OK;
    case RTOC_OPT_CONDENSE_REGISTER: *value = c->cond_register; return RTOC_OK;
    case RTOC_OPT_BACKWARD_WAVES: *value = c->ks->bwd_waves[c->bwd_variant]; return RTOC_OK;
    case RTOC_OPT_SWITCHING_TRANSPORT: *value = c->exact_transport; return RTOC_OK;
    case RTOC_OPT_UNCONSTR_DENSE: *value = c->unconstr_dense; return RTOC_OK;
    case RTOC_OPT_LINEARIZE_FUSED: *value = c->linearize_fused; return RTOC_OK;
    case RTOC_OPT_CONE_JACOBIAN: *value = c->exact_cone_jacobian; return RTOC_OK;
    default: return RTOC_ERR_BAD_ARG;   // the double-valued options (RTOC_OPT_MAX_DTS0, RTOC_OPT_CONTACT_INV_DAMPING)
  }
}

int rtoc_set_option(rtoc_ctx* c, int option, int64_t value) {
  if (!c) return RTOC_ERR_BAD_ARG;
  c->epoch++;
  switch (option) {
    case RTOC_OPT_WRITEBACK_KKT:
      c->writeback = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_MAX_DTS0: {
      double d;
      memcpy(&d, &value, sizeof(d));
      if (!(d > 0.0)) return RTOC_ERR_BAD_ARG;
      c->max_dts0 = d;
      return RTOC_OK;
    }
    case RTOC_OPT_CONE_JACOBIAN:
      c->exact_cone_jacobian = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_LINEARIZE_FUSED:
      c->linearize_fused = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_LINEARIZE_DOFS_PER_PASS:
      if (value < 0 || value > rbd::LIN_MAX_DPP) return RTOC_ERR_BAD_ARG;
      {
        const int before = c->lin_dpp;
        c->lin_dpp = (int)value;
        const int rc = c->h_model ? apply_linearize_plan(c) : RTOC_OK;
        if (rc) c->lin_dpp = before;   // refused (LDS): the plan in force is the old one
        return rc;
      }
    case RTOC_OPT_UNCONSTR_DENSE:
      c->unconstr_dense = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_IMPACT_CONES:
      c->impact_cones = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_SWITCHING_TRANSPORT:
      c->exact_transport = value ? 1 : 0;
      return RTOC_OK;
    case RTOC_OPT_CONTACT_INV_DAMPING: {
      double d;
      memcpy(&d, &value, sizeof(d));
      if (!(d >= 0.0)) return RTOC_ERR_BAD_ARG;
      c->contact_inv_damping = d;
      return RTOC_OK;
    }
    case RTOC_OPT_BACKWARD_WAVES: {
      if (value == 0) {
        c->bwd_variant = (c->ks->nvariants >= 3) ? c->ks->nvariants - 1 : 0;
        return RTOC_OK;
      }
      for (int v = 0; v < c->ks->nvariants; ++v)
        if (c->ks->bwd_waves[v] == (int)value) {
          c->bwd_variant = v;
          return RTOC_OK;
        }
      return RTOC_ERR_BAD_ARG;
    }
    case RTOC_OPT_CONDENSE_SPLIT:
      if (value != 0 && value != 1) return RTOC_ERR_BAD_ARG;
      c->condense_split = (int)value;
      return RTOC_OK;
    case RTOC_OPT_BACKWARD_REGISTER:
      if (value != 0 && value != 1 && value != 2) return RTOC_ERR_BAD_ARG;
      c->bwd_register = (int)value;
      return RTOC_OK;
    case RTOC_OPT_CONDENSE_REGISTER:
      if (value != 0 && value != 1 && value != 2) return RTOC_ERR_BAD_ARG;
      c->cond_register = (int)value;
      return RTOC_OK;
    case RTOC_OPT_GRAPH:
      if (value != 0 && value != 1) return RTOC_ERR_BAD_ARG;
      c->use_graph = (int)value;
      return RTOC_OK;
    case RTOC_OPT_FXX_STRUCTURE:
      if (value < 0 || value > 2) return RTOC_ERR_BAD_ARG;
      c->fxx_mode = (int)value;
      return RTOC_OK;
    case RTOC_OPT_CONDENSE_KEEP_QAF:
      if (value != 0 && value != 1) return RTOC_ERR_BAD_ARG;
      c->keep_qaf = (int)value;
      return RTOC_OK;
    case RTOC_OPT_SWEEP_CHUNKS:
      if (value < 1 || value > RTOC_MAX_CHUNK_EVENTS) return RTOC_ERR_BAD_ARG;
      c->sweep_chunks = (int)value;
      return RTOC_OK;
    case RTOC_OPT_BACKWARD_SCAN:
      if (value < 0 || value > 2) return RTOC_ERR_BAD_ARG;
      c->backward_scan = (int)value;
      // element / value-record buffers now, so that the launches themselves never allocate (stream capture)
      if (value != 0 && !(value == 2 && c->batch > RTOC_SCAN_AUTO_MAX_BATCH)) return ensure_scan_buffers(c);
      return RTOC_OK;
    default:
      return RTOC_ERR_BAD_ARG;
  }
}

static int ensure_buffer(rtoc_ctx* c, int b) {
  if (c->buf[b]) return RTOC_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc((void**)&c->buf[b], c->count[b] * sizeof(double)));
  HIP_TRY(hipMemsetAsync(c->buf[b], 0, c->count[b] * sizeof(double), c->stream));
  c->owned[b] = true;
  c->epoch++;
  return RTOC_OK;
}

// switching-time optimisation on the device (sto.hpp): kernel arguments, one thread per instance
static int sto_count_events(const rtoc_ctx* c) {
  int n = 0;
  for (int i = 0; i + 1 < c->nstages; ++i)
    if (c->h_grid[i].type == RTOC_GRID_IMPACT || c->h_grid[i].type == RTOC_GRID_LIFT) ++n;
  return n;
}

static StoDevArgs sto_args(rtoc_ctx* c) {
  StoDevArgs a;
  memset(&a, 0, sizeof(a));
  const size_t ne = (size_t)c->batch * (c->sto_nev > 0 ? c->sto_nev : 1);
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.grid = c->d_grid;
  a.ts = c->d_ts;
  a.dt_inst = c->d_dt;
  a.con = c->d_sto_con;
  a.min_dwell = c->d_min_dwell;
  a.cost_lt = c->d_sto_cost;
  a.cost_qtt = c->d_sto_cost ? c->d_sto_cost + ne : nullptr;
  a.lt = c->d_sto_out;
  a.qtt = c->d_sto_out + ne;
  a.err = c->d_sto_out + 2 * ne;
  a.kkterr = c->d_kkterr;
  a.steps = c->buf[RTOC_BUF_STEP];
  a.nstages = c->nstages, a.batch = c->batch, a.nev = c->sto_nev;
  a.kkt_stride = c->L.kkt.stride, a.scal_off = c->L.kkt.off[RTOC_KKT_SCAL];
  a.dir_stride = c->L.dir.stride, a.dts_off = c->L.dir.off[RTOC_DIR_DTS];
  a.t0 = c->sto_t0, a.T = c->sto_T, a.barrier = c->sto_barrier, a.tau = c->sto_tau, a.sto_reg = c->sto_reg;
  return a;
}
#define STO_LAUNCH(kernel, c)                                                                                   \
  do {                                                                                                          \
    hipLaunchKernelGGL(kernel, dim3(((c)->batch + 63) / 64), dim3(64), 0, (c)->stream, sto_args(c));            \
    HIP_TRY(hipGetLastError());                                                                                 \
  } while (0)

int rtoc_upload(rtoc_ctx* c, int buffer, size_t offset, const double* host, size_t count) {
  if (!c || buffer < 0 || buffer >= RTOC_NUM_BUFFERS || !host) return RTOC_ERR_BAD_ARG;
  if (count > c->count[buffer] || offset > c->count[buffer] - count) return RTOC_ERR_BAD_ARG;
  int rc = ensure_buffer(c, buffer);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->buf[buffer] + offset, host, count * sizeof(double), hipMemcpyHostToDevice,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (buffer == RTOC_BUF_KKT) c->fxx_state = 0;  // re-checked by the next backward recursion (RTOC_OPT_FXX_STRUCTURE)
  if (buffer == RTOC_BUF_SOL) c->vals_fresh = 0;  // the pre-pass kinematics belong to the previous iterate
  return RTOC_OK;
}

int rtoc_download(rtoc_ctx* c, int buffer, size_t offset, double* host, size_t count) {
  if (!c || buffer < 0 || buffer >= RTOC_NUM_BUFFERS || !host) return RTOC_ERR_BAD_ARG;
  if (count > c->count[buffer] || offset > c->count[buffer] - count || !c->buf[buffer]) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(host, c->buf[buffer] + offset, count * sizeof(double), hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

void* rtoc_device_ptr(rtoc_ctx* c, int buffer) {
  if (!c || buffer < 0 || buffer >= RTOC_NUM_BUFFERS) return nullptr;
  const bool fresh = !c->buf[buffer];
  if (ensure_buffer(c, buffer)) return nullptr;
  // the zero fill of a lazily allocated buffer runs on the context's stream: it must have landed before a
  // caller writes through the pointer on a stream of its own
  if (fresh && hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
  if (buffer == RTOC_BUF_KKT) {
    c->fxx_state = 0;  // the caller may write through the pointer
    c->kkt_exposed = true;
  }
  return c->buf[buffer];
}

size_t rtoc_buffer_count(const rtoc_ctx* c, int buffer) {
  if (!c || buffer < 0 || buffer >= RTOC_NUM_BUFFERS) return 0;
  return c->count[buffer];
}

int rtoc_bind(rtoc_ctx* c, int buffer, void* device_ptr) {
  if (!c || buffer < 0 || buffer >= RTOC_NUM_BUFFERS || !device_ptr) return RTOC_ERR_BAD_ARG;
  if (((uintptr_t)device_ptr & 63) != 0) return RTOC_ERR_BAD_ARG;  // records are 64 B aligned
  HIP_TRY(hipSetDevice(c->device));
  if (c->owned[buffer] && c->buf[buffer]) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(c->buf[buffer]));
  }
  c->buf[buffer] = (double*)device_ptr;
  c->owned[buffer] = false;
  if (buffer == RTOC_BUF_KKT) c->fxx_state = 0;
  c->epoch++;
  return RTOC_OK;
}

// ---- hot path ---------------------------------------------------------------------------
// RTOC_OPT_BACKWARD_SCAN: the scan covers grids without switching-time optimisation; others take the serial kernel
static bool grid_has_sto(const rtoc_ctx* c) {
  for (int i = 0; i < c->nstages; ++i)
    if (c->h_grid[i].sto || c->h_grid[i].sto_next) return true;
  return false;
}
// the backward recursion: every grid (with switching-time optimisation: matrix scan + serial vector pass, riccati_scan_sto.hpp)
static bool scan_applies(const rtoc_ctx* c) {
  if (!c->backward_scan || !c->h_grid) return false;
  if (c->backward_scan == 2 && c->batch > RTOC_SCAN_AUTO_MAX_BATCH) return false;  // auto: latency regime only
  if (c->nstages > SCAN_STO_MAX_STAGES && grid_has_sto(c)) return false;            // the vector pass keeps the grid in LDS
  return true;
}
// the forward recursion as a prefix scan: grids without switching-time optimisation (the dts chain is not a fixed affine map)
static bool forward_scan_applies(const rtoc_ctx* c) { return scan_applies(c) && !grid_has_sto(c); }

// Backward recursion as a horizon scan (riccati_scan.hpp): elements, log2 combination levels, then the
// policies of all grid points at once by the tile-split backward kernel in its one-stage mode.
static int ensure_scan_buffers(rtoc_ctx* c) {
  const KernelSet* ks = c->ks;
  const size_t per = (size_t)c->batch * c->max_stages;
  for (int i = 0; i < 3; ++i)
    if (!c->d_scan[i]) {
      const size_t n = per * (i < 2 ? ks->scan_elt_stride : ks->scan_ps_stride);
      HIP_TRY(hipMalloc((void**)&c->d_scan[i], n * sizeof(double)));
    }
  if (!c->d_scan_sto && grid_has_sto(c)) HIP_TRY(hipMalloc((void**)&c->d_scan_sto, per * ks->sto_scr_stride * sizeof(double)));  // (grids with STO only)
  return RTOC_OK;
}

static int launch_backward_scan(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  const KernelSet* ks = c->ks;
  int rc0 = ensure_scan_buffers(c);
  if (rc0) return rc0;
  const int n = c->nstages, nb = end - first;
  ScanArgs s;
  s.kkt = c->buf[RTOC_BUF_KKT];
  s.grid = c->d_grid;
  s.status = c->d_status;
  s.src = c->d_scan[1];
  s.dst = c->d_scan[0];
  s.ps = c->d_scan[2];
  s.nstages = n;
  s.batch = end;
  s.first = first;
  s.dist = 0;
  hipLaunchKernelGGL(ks->scan_elt, dim3(n, nb), dim3(SCAN_ELT_NT), ks->scan_elt_lds, stream, s);
  int cur = 0;
  for (int d = 1; d < n; d *= 2) {
    s.src = c->d_scan[cur];
    s.dst = c->d_scan[cur ^ 1];
    s.dist = d;
    hipLaunchKernelGGL(ks->scan_comb, dim3(n - d, nb, 2), dim3(ks->scan_comb_threads), ks->scan_comb_lds, stream, s);
    cur ^= 1;
  }
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.kkt_rw = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.grid = c->d_grid;
  a.status = c->d_status;
  a.prof = nullptr;
  a.nstages = n;
  a.batch = end;
  a.first = first;
  a.writeback = c->writeback;
  a.max_dts0 = c->max_dts0;
  a.scan_ps = c->d_scan[2];
  a.scan_ps_stride = ks->scan_ps_stride;
  a.scan_ps_soff = ks->scan_ps_soff;
  const int v = ks->scan_policy_variant;
  const bool sto = grid_has_sto(c);
  StoScanArgs t;
  // Grids with switching-time optimisation: the bundles of the vector pass (everything of the vector recursion that does not depend
  // on the chain) are prepared by n - 1 more workgroups per instance of the SAME launch -- unless the policy workgroups write the
  // mutated Quu, lu back into the KKT records (RTOC_OPT_WRITEBACK_KKT), which the preparation reads: then it runs first, by itself.
  const bool ride = sto && !c->writeback && ks->sto_prep_lds <= ks->bwd_lds[v];
  if (sto) {
    t.kkt = c->buf[RTOC_BUF_KKT], t.ric = c->buf[RTOC_BUF_RIC], t.grid = c->d_grid, t.status = c->d_status;
    t.ps = c->d_scan[2], t.scr = c->d_scan_sto;
    t.nstages = n, t.batch = end, t.first = first, t.max_dts0 = c->max_dts0, t.prof = c->d_prof;
    if (!ride) hipLaunchKernelGGL(ks->sto_prep, dim3(n - 1, nb), dim3(SCAN_STO_PREP_NT), ks->sto_prep_lds, stream, t);
  }
  a.sto_scr = ride ? c->d_scan_sto : nullptr;
  hipLaunchKernelGGL(ks->bwd[v], dim3(nb, ride ? 2 * n - 1 : n), dim3(64 * ks->bwd_waves[v]), ks->bwd_lds[v], stream, a);
  if (sto) hipLaunchKernelGGL(ks->sto_vec, dim3(nb), dim3(ks->sto_vec_threads), ks->sto_vec_lds, stream, t);   // s, k, m, the STO quantities
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// RTOC_OPT_FXX_STRUCTURE: may the structure-exploiting backward kernel run on the resident records?
static int check_fxx(rtoc_ctx* c) {
  if (!c->d_fxx_flag) HIP_TRY(hipMalloc((void**)&c->d_fxx_flag, sizeof(int)));
  HIP_TRY(hipMemsetAsync(c->d_fxx_flag, 0, sizeof(int), c->stream));
  FxxCheckArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.grid = c->d_grid;
  a.flag = c->d_fxx_flag;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nv = c->dims.nv;
  a.np = c->dims.np;
  a.fxx_off = c->L.kkt.off[RTOC_KKT_FXX];
  a.stride = c->L.kkt.stride;
  hipLaunchKernelGGL(fxx_structure_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  int bad = 1;
  HIP_TRY(hipMemcpyAsync(&bad, c->d_fxx_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the kernel choice of the backward recursion is part of a captured graph: a new epoch only when the ANSWER changes, not
  // whenever the records were re-uploaded (the documented loop re-linearises and uploads before every iteration)
  if (c->fxx_last != (bad ? 2 : 1)) c->epoch++;
  c->fxx_state = c->fxx_last = bad ? 2 : 1;
  return RTOC_OK;
}
// does the backward recursion of this context have a structure-exploiting kernel to choose? (quadruped shapes: the structured forms of
// the role-split / register-resident kernels; iCub-size shapes: riccati_backward_rw_kernel, which exists in the structured form only)
static bool rw_configured(const rtoc_ctx* c);
static bool fxx_structured(rtoc_ctx* c) {
  if (!((c->ks->bwd_sa && c->bwd_variant == 3) || rw_configured(c))) return false;  // no structured kernel for this shape / variant: nothing to check
  if (c->fxx_mode == 1) return false;
  if (c->fxx_mode == 2) return true;
  if (c->fxx_state == 0 && check_fxx(c) != RTOC_OK) return false;
  return c->fxx_state == 1;
}

// RTOC_OPT_BACKWARD_REGISTER: the register-resident kernel (riccati_backward_rv.hpp), one launch for the whole horizon.
static bool rv_applies(rtoc_ctx* c) {
  if (!(c->bwd_register && c->ks->bwd_rv && c->h_grid && c->nstages >= 2 && c->nstages <= RV_MAX_STAGES && !c->writeback &&
        c->bwd_variant == ((c->ks->nvariants >= 3) ? c->ks->nvariants - 1 : 0)))   // (an explicit RTOC_OPT_BACKWARD_WAVES keeps its kernel)
    return false;
  // grids with switching-time optimisation: the STO instantiation, which exists in the structured-Fxx form only
  if (grid_has_sto(c)) return c->ks->bwd_rv_sto && fxx_structured(c);
  return true;
}
static int launch_backward_rv(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  const KernelSet* ks = c->ks;
  const int N = c->nstages - 1, nb = end - first;
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.kkt_rw = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.grid = c->d_grid;
  a.status = c->d_status;
  a.nstages = c->nstages;
  a.batch = end;
  a.first = first;
  a.max_dts0 = c->max_dts0;
  a.prof = c->d_prof;
  // one launch for the whole horizon: regular, lift, impact and switching-constraint grid points are all the kernel's own
  a.seg_hi = N - 1;
  a.seg_lo = 0;
#ifdef RTOC_RV_DEBUG_MASK
  if (const char* e = getenv("RTOC_RV_DEBUG")) a.scan_ps_soff = atoi(e);
#endif
  const bool sto = grid_has_sto(c);
  const bwd_fn kern = sto ? ks->bwd_rv_sto : ((ks->bwd_rv_sa && fxx_structured(c)) ? ks->bwd_rv_sa : ks->bwd_rv);   // RTOC_OPT_FXX_STRUCTURE, as for the role-split kernel
  // a bound buffer may have been rewritten since the device check that chose the structured form: the kernel verifies as it goes
  a.check_fxx = (kern != ks->bwd_rv && c->fxx_mode == 0 && (!c->owned[RTOC_BUF_KKT] || c->kkt_exposed)) ? 1 : 0;
  if (N >= 1) hipLaunchKernelGGL(kern, dim3(nb), dim3(64), ks->bwd_rv_lds, stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// RTOC_OPT_BACKWARD_REGISTER on the iCub-size shapes: the register-wide kernel (riccati_backward_rw.hpp; one wave per instance and
// SIMD, structured Fxx).  1 (default): batches that fill the machine -- below one instance per CU the tile-split kernel's four waves
// per instance finish a horizon sooner --, 2: always.  Switching-constraint grid points are one-stage launches of the tile-split
// kernel between the segments (P+ / s+ through the Riccati records).
static bool rw_configured(const rtoc_ctx* c) {
  return c->bwd_register && c->ks->bwd_rw && c->h_grid && c->nstages >= 2 && c->nstages <= RV_MAX_STAGES && !c->writeback && !grid_has_sto(c) &&
         c->bwd_variant == ((c->ks->nvariants >= 3) ? c->ks->nvariants - 1 : 0) && (c->bwd_register >= 2 || c->batch > c->num_cus);
}
static bool rw_applies(rtoc_ctx* c) {
  if (!rw_configured(c)) return false;
  // the kernel never loads the structured rows of Fxx, so it cannot verify them: on a bound buffer (the caller may have rewritten the
  // records since the last check) the device check runs again before every recursion, unless the caller asserts the structure
  if (c->fxx_mode == 0 && (!c->owned[RTOC_BUF_KKT] || c->kkt_exposed) && !c->capturing) c->fxx_state = 0;
  return fxx_structured(c);
}
static int launch_backward_rw(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  const KernelSet* ks = c->ks;
  const int N = c->nstages - 1, nb = end - first;
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.kkt_rw = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.grid = c->d_grid;
  a.status = c->d_status;
  a.nstages = c->nstages;
  a.batch = end;
  a.first = first;
  a.max_dts0 = c->max_dts0;
  a.prof = c->d_prof;
  const int v1 = ks->scan_policy_variant;
  auto constrained = [&](int st) { return c->h_grid[st].type != RTOC_GRID_IMPACT && c->h_grid[st].dims > 0; };
  auto one_stage = [&](int st) {   // tile-split kernel, grid point st only (st == N: the terminal record)
    BwdArgs o = a;
    o.scan_ps = c->buf[RTOC_BUF_RIC] + c->L.ric.off[RTOC_RIC_P];
    o.scan_ps_stride = c->L.ric.stride;
    o.scan_ps_soff = c->L.ric.off[RTOC_RIC_S] - c->L.ric.off[RTOC_RIC_P];
    o.seg_hi = o.seg_lo = st;
    hipLaunchKernelGGL(ks->bwd[v1], dim3(nb, 1), dim3(64 * ks->bwd_waves[v1]), ks->bwd_lds[v1], stream, o);
  };
  if (N == 0 || constrained(N - 1)) one_stage(N);   // nobody else writes the terminal record then
  int hi = N - 1;
  while (hi >= 0) {
    if (constrained(hi)) {
      one_stage(hi);
      --hi;
      continue;
    }
    int lo = hi;
    while (lo > 0 && !constrained(lo - 1)) --lo;
    a.seg_hi = hi;
    a.seg_lo = lo;
    hipLaunchKernelGGL(ks->bwd_rw, dim3(nb), dim3(ks->bwd_rw_threads), ks->bwd_rw_lds, stream, a);
    hi = lo - 1;
  }
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int launch_backward_range(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  if (scan_applies(c)) return launch_backward_scan(c, first, end, stream);
  if (rv_applies(c)) return launch_backward_rv(c, first, end, stream);
  if (rw_applies(c)) return launch_backward_rw(c, first, end, stream);
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.kkt_rw = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.grid = c->d_grid;
  a.status = c->d_status;
  a.prof = c->d_prof;
  a.nstages = c->nstages;
  a.batch = end;
  a.first = first;
  a.writeback = c->writeback;
  a.max_dts0 = c->max_dts0;
  const int v = c->bwd_variant;
  const int ni = c->ks->bwd_inst[v];
  const bwd_fn kern = fxx_structured(c) ? c->ks->bwd_sa : c->ks->bwd[v];
  hipLaunchKernelGGL(kern, dim3((end - first + ni - 1) / ni), dim3(64 * c->ks->bwd_waves[v]), c->ks->bwd_lds[v], stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}
static int launch_backward(rtoc_ctx* c) { return launch_backward_range(c, 0, c->batch, c->stream); }

// Forward recursion as a prefix scan of the closed-loop maps (riccati_scan.hpp): maps of all grid points,
// log2 composition levels (dx of every grid point), then du / dlmdgmm / dxi of all grid points at once.
static int launch_forward_scan(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  const KernelSet* ks = c->ks;
  int rc0 = ensure_scan_buffers(c);
  if (rc0) return rc0;
  const int n = c->nstages, nb = end - first, N = n - 1;
  FwdScanArgs s;
  s.kkt = c->buf[RTOC_BUF_KKT];
  s.ric = c->buf[RTOC_BUF_RIC];
  s.dir = c->buf[RTOC_BUF_DIR];
  s.dx0 = c->buf[RTOC_BUF_DX0];
  s.grid = c->d_grid;
  s.src = c->d_scan[1];
  s.dst = c->d_scan[0];
  s.nstages = n;
  s.batch = end;
  s.first = first;
  s.dist = 0;
  hipLaunchKernelGGL(ks->fscan_elt, dim3(N, nb), dim3(SCAN_FWD_NT), ks->fscan_lds, stream, s);
  int cur = 0;
  for (int d = 1; d < N; d *= 2) {
    s.src = c->d_scan[cur];
    s.dst = c->d_scan[cur ^ 1];
    s.dist = d;
    hipLaunchKernelGGL(ks->fscan_comb, dim3(N - d, nb), dim3(SCAN_FWD_NT), ks->fscan_lds, stream, s);
    cur ^= 1;
  }
  hipLaunchKernelGGL(ks->fscan_fin, dim3(n, nb), dim3(64), 0, stream, s);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int launch_forward_range(rtoc_ctx* c, int first, int end, hipStream_t stream) {
  if (forward_scan_applies(c)) return launch_forward_scan(c, first, end, stream);
  FwdArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.dx0 = c->buf[RTOC_BUF_DX0];
  a.grid = c->d_grid;
  a.nstages = c->nstages;
  a.batch = end;
  a.first = first;
  // (a structured-Fxx form of this kernel -- top half of Fxx not read, 15 % fewer bytes -- was measured at 1.27 vs
  // 1.28 ms: the kernel is bound by its load queue, not by the bytes it requests; not kept)
  hipLaunchKernelGGL(c->ks->fwd, dim3(end - first), dim3(c->ks->fwd_threads), (size_t)((a.nstages + 3) & ~3) * sizeof(int), stream, a);  // grid table in LDS
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}
static int launch_forward(rtoc_ctx* c) { return launch_forward_range(c, 0, c->batch, c->stream); }

// Backward + forward sweep of the whole batch as a two-stream pipeline over instance chunks: the
// forward recursion of chunk i (HBM-bound, a few small waves per CU) runs under the backward
// recursion of chunk i+1 (MFMA / LDS-bound, leaves most of the HBM bandwidth idle).  Results are
// those of rtoc_riccati_backward followed by rtoc_riccati_forward.
static int launch_sweep(rtoc_ctx* c) {
  const int nch = (c->sweep_chunks > 0) ? c->sweep_chunks : 1;
  if (nch == 1 || scan_applies(c)) {  // the scan's element buffers are not chunked
    int rc = launch_backward(c);
    return rc ? rc : launch_forward(c);
  }
  const int per = (((c->batch + nch - 1) / nch) + 3) & ~3;  // whole 4-instance workgroups
  HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  for (int i = 0; i * per < c->batch; ++i) {
    const int first = i * per, end = (first + per < c->batch) ? first + per : c->batch;
    int rc = launch_backward_range(c, first, end, c->stream);
    if (rc) return rc;
    hipEvent_t e = c->ev_chunk[i % RTOC_MAX_CHUNK_EVENTS];
    HIP_TRY(hipEventRecord(e, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream2, e, 0));
    rc = launch_forward_range(c, first, end, c->stream2);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return RTOC_OK;
}

// RTOC_OPT_CONDENSE_REGISTER: the contact grid points by condense_rv_kernel (one wave per work item, products chained through
// registers), the impact grid points by condense_kernel.  Measured per 4096 ANYmal trot instances: 4.80 -> 4.00 ms without rows,
// 5.16 -> 4.55 ms with 72 joint-limit rows and 4 friction cones.
// friction cones of point contacts are condensed INSIDE condense_rv_kernel (their Gram product's tiles go straight into the seeds and
// operands of the condensation); wrench cones need their own kernel ahead of it
static bool cond_rv_fuses_cones(const rtoc_ctx* c) {
  return c->cone_contacts > 0 && c->cone_rows == RTOC_FRICTION_ROWS && c->cone_dim == 3 && c->ks->cond_fuses_cones && c->ks->cond_rv_cones;
}
static bool cond_rv_applies(const rtoc_ctx* c) {
  if (!c->cond_register || !c->ks->cond_rv || c->condense_split || c->keep_qaf) return false;
  if (c->cone_contacts > 0 && !cond_rv_fuses_cones(c) && c->cond_register < 2) return false;
  return c->n_stage_contact + c->n_stage_impact == c->nstages - 1;
}

static int launch_condense(rtoc_ctx* c) {
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  CondArgs a;
  a.stage_list = nullptr;
  a.nlist = 0;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.grid = c->d_grid;
  a.status = c->d_status;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.damping = c->contact_inv_damping;
  a.prof = c->d_prof;
  a.con = (c->nrows > 0) ? c->buf[RTOC_BUF_CON] : nullptr;
  a.rows = c->d_rows;
  a.entry = c->d_entry;
  a.pair = reinterpret_cast<const int4*>(c->d_pair);
  a.nrows = c->nrows;
  a.nl = c->L.con;
  a.kl = c->L.kkt;
  a.cl = c->L.cdd;
  const int nblocks = c->batch * (c->nstages - 1);
  a.cone_rows = 0;
  a.keep_qaf = c->keep_qaf;
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  const bool rv = cond_rv_applies(c);
  if ((rv ? cond_rv_fuses_cones(c) : (c->condense_split || c->ks->cond_fuses_cones)) && c->cone_contacts > 0) {  // the cone rows ride with the MJtJinv kernel / in wave 1 of the fused kernel / inside condense_rv_kernel
    if (!c->buf[RTOC_BUF_CONE] || !c->buf[RTOC_BUF_CON]) return RTOC_ERR_NOT_READY;
    const bool wrench = c->cone_rows == RTOC_WRENCH_ROWS;
    a.cone_rows = c->cone_rows;
    a.cone_con = c->buf[RTOC_BUF_CON];
    a.cone = c->buf[RTOC_BUF_CONE];
    a.cone_contacts = c->cone_contacts;
    a.cone_dim = c->cone_dim;
    a.cone_row0 = c->dims.nc_max - c->cone_rows * c->cone_contacts;
    a.cone_stride = wrench ? rtoc_wrench_cone_stride(c->cone_contacts) : rtoc_cone_stride(c->dims.nv, c->cone_contacts);
    a.cone_dgdf_off = rtoc_cone_dgdf_off(c->dims.nv, c->cone_contacts);
    a.cone_impact = c->impact_cones;
  }
  if (rv) {
    a.stage_list = c->d_stage_list;
    a.nlist = c->n_stage_contact;
#ifdef RTOC_CRV_DEBUG_LDS_PAD   // occupancy experiments (debug builds only): extra dynamic LDS per work item, clamped to what a launch accepts
    static const int lds_pad_env = getenv("RTOC_CRV_LDS_PAD") ? atoi(getenv("RTOC_CRV_LDS_PAD")) : 0;
    const int lds_room = 64 * 1024 - c->ks->cond_rv_lds;
    const int lds_pad = lds_pad_env < 0 ? 0 : (lds_pad_env > lds_room ? lds_room : lds_pad_env);
#else
    constexpr int lds_pad = 0;
#endif
    if (a.nlist > 0) {
      hipLaunchKernelGGL(a.cone_rows ? c->ks->cond_rv : c->ks->cond_rv_nc, dim3(c->batch * a.nlist), dim3(64), c->ks->cond_rv_lds + lds_pad, c->stream, a);
      HIP_TRY(hipGetLastError());
    }
    a.stage_list = c->d_stage_list + c->n_stage_contact;
    a.nlist = c->n_stage_impact;
    if (a.nlist > 0) hipLaunchKernelGGL(c->ks->cond, dim3(c->batch * a.nlist), dim3(c->ks->cond_threads), c->ks->cond_lds, c->stream, a);
  } else if (c->condense_split) {
    hipLaunchKernelGGL(c->ks->mjt, dim3(nblocks), dim3(64), c->ks->mjt_lds, c->stream, a);
    hipLaunchKernelGGL(c->ks->cond_split, dim3(nblocks), dim3(c->ks->cond_threads), c->ks->cond_split_lds, c->stream, a);
  } else {
    hipLaunchKernelGGL(c->ks->cond, dim3(nblocks), dim3(c->ks->cond_threads), c->ks->cond_lds, c->stream, a);
  }
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int launch_expand(rtoc_ctx* c, double tau) {
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  ExpArgs a;
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.grid = c->d_grid;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.cl = c->L.cdd;
  a.dl = c->L.dir;
  a.tau = tau;
  a.con = (c->nrows > 0) ? c->buf[RTOC_BUF_CON] : nullptr;
  a.rows = c->d_rows;
  a.nrows = c->nrows;
  a.nl = c->L.con;
  a.steps = (unsigned long long*)c->buf[RTOC_BUF_STEP];
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  a.prof = c->d_prof;
  hipLaunchKernelGGL(fill_steps_kernel, dim3((2 * c->batch + 255) / 256), dim3(256), 0, c->stream,
                     c->buf[RTOC_BUF_STEP], 2 * c->batch);
  const int nblocks = c->batch * (c->nstages - 1);
  hipLaunchKernelGGL(c->ks->expd, dim3(nblocks), dim3(c->ks->expd_threads), c->ks->expd_lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int launch_cones(rtoc_ctx* c, int phase, double tau) {  // 0 condense, 1 expand, 2 update
  if (!c->buf[RTOC_BUF_CONE] || !c->buf[RTOC_BUF_CON]) return RTOC_ERR_NOT_READY;
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  ConeArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = c->buf[RTOC_BUF_CON];
  a.cone = c->buf[RTOC_BUF_CONE];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.grid = c->d_grid;
  a.steps = (unsigned long long*)c->buf[RTOC_BUF_STEP];
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.max_contacts = c->cone_contacts;
  a.contact_dim = c->cone_dim;
  a.prof = nullptr;
  const bool wrench = c->cone_rows == RTOC_WRENCH_ROWS;
  a.rows_per_contact = c->cone_rows;
  a.row0 = c->dims.nc_max - c->cone_rows * c->cone_contacts;
  a.cone_stride = wrench ? rtoc_wrench_cone_stride(c->cone_contacts) : rtoc_cone_stride(c->dims.nv, c->cone_contacts);
  a.dgdf_off = rtoc_cone_dgdf_off(c->dims.nv, c->cone_contacts);
  a.impact_cones = c->impact_cones;
  a.tau = tau;
  a.kl = c->L.kkt;
  a.cl = c->L.cdd;
  a.nl = c->L.con;
  a.dl = c->L.dir;
  const dim3 grid(c->batch * (c->nstages - 1));
  if (phase == 0)
    hipLaunchKernelGGL(wrench ? c->ks->wcond : c->ks->ccond, grid, dim3(64), 0, c->stream, a);
  else if (phase == 1)
    hipLaunchKernelGGL(wrench ? c->ks->wexp : c->ks->cexp, grid, dim3(64), 0, c->stream, a);
  else
    hipLaunchKernelGGL(cone_update_kernel, grid, dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

#define CHECK_READY(c)                       \
  if (!(c)) return RTOC_ERR_BAD_ARG;         \
  if ((c)->nstages < 2) return RTOC_ERR_NOT_READY; \
  HIP_TRY(hipSetDevice((c)->device));

static int launch_state_correction(rtoc_ctx* c, int mode) {
  if (!c->buf[RTOC_BUF_SE3]) return RTOC_ERR_BAD_ARG;
  SeArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.dx0 = c->buf[RTOC_BUF_DX0];
  a.se3 = c->buf[RTOC_BUF_SE3];
  a.grid = c->d_grid;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.kl = c->L.kkt;
  a.dl = c->L.dir;
  a.nx = c->L.nx;
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  const int nblocks = (mode == 2) ? c->batch : c->batch * c->nstages;
  if (mode == 0)
    hipLaunchKernelGGL(state_correction_kernel<0>, dim3(nblocks), dim3(64), 0, c->stream, a);
  else if (mode == 1)
    hipLaunchKernelGGL(state_correction_kernel<1>, dim3(nblocks), dim3(64), 0, c->stream, a);
  else
    hipLaunchKernelGGL(state_correction_kernel<2>, dim3(nblocks), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_correct_state_equation(rtoc_ctx* c) {
  CHECK_READY(c);
  if (c->dims.np != 6) return RTOC_ERR_BAD_ARG;  // floating base only (hasFloatingBase())
  return launch_state_correction(c, 0);
}

int rtoc_correct_costate_direction(rtoc_ctx* c) {
  CHECK_READY(c);
  if (c->dims.np != 6) return RTOC_ERR_BAD_ARG;
  return launch_state_correction(c, 1);
}

int rtoc_compute_initial_state_direction(rtoc_ctx* c) {
  CHECK_READY(c);
  if (c->dims.np != 6) return RTOC_ERR_BAD_ARG;
  return launch_state_correction(c, 2);
}

int rtoc_condense(rtoc_ctx* c) {
  CHECK_READY(c);
  int rc = RTOC_OK;
  if (c->cone_contacts > 0 && (cond_rv_applies(c) ? !cond_rv_fuses_cones(c) : (!c->condense_split && !c->ks->cond_fuses_cones)))
    rc = launch_cones(c, 0, 0.0);  // Constraints::condenseSlackAndDual first
  if (!rc) rc = launch_condense(c);
  if (!rc && c->buf[RTOC_BUF_SE3] && c->dims.np == 6) rc = launch_state_correction(c, 0);
  return rc;
}

int rtoc_riccati_backward(rtoc_ctx* c) {
  CHECK_READY(c);
  return launch_backward(c);
}

int rtoc_riccati_forward(rtoc_ctx* c) {
  CHECK_READY(c);
  return launch_forward(c);
}

// RTOC_OPT_GRAPH: run `body` (a sequence of kernel launches on c->stream, no allocation, no synchronisation) from a
// captured hipGraph.  The first call at a given configuration epoch runs it plainly (lazy allocations happen there),
// the second captures and instantiates, later calls are one hipGraphLaunch -- a single-OCP Newton iteration is ~25
// small kernels, whose launch gaps are a third of its latency.
extern "C++" {
template <class Body>
static int run_graphed(rtoc_ctx* c, rtoc_ctx::GraphSlot* g, double p0, double p1, Body body) {
  if (!c->use_graph) return body();
  (void)rw_applies(c);      // (bound buffers of the iCub-size shapes: the re-check of every recursion, see rw_applies)
  (void)fxx_structured(c);  // may check the records (synchronises): before, never inside, a capture
  if (g->exec && g->epoch == c->epoch && g->p0 == p0 && g->p1 == p1) {
    HIP_TRY(hipGraphLaunch(g->exec, c->stream));
    c->graph_replays++;
    return RTOC_OK;
  }
  if (!(g->warm && g->warm_epoch == c->epoch)) {
    const int rc = body();
    g->warm = true;
    g->warm_epoch = c->epoch;  // after the body: its lazy allocations bump the epoch
    return rc;
  }
  if (g->exec) {
    (void)hipGraphExecDestroy(g->exec);
    g->exec = nullptr;
  }
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  c->capturing = true;
  const int rc = body();
  c->capturing = false;
  const hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) ctx_set_err(e, __LINE__);
    return rc ? rc : RTOC_ERR_HIP;
  }
  const hipError_t e2 = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) {
    g->exec = nullptr;
    ctx_set_err(e2, __LINE__);
    return RTOC_ERR_HIP;
  }
  g->epoch = c->epoch;
  g->p0 = p0;
  g->p1 = p1;
  HIP_TRY(hipGraphLaunch(g->exec, c->stream));
  c->graph_replays++;
  return RTOC_OK;
}
}  // extern "C++"

int rtoc_graph_replay_count(rtoc_ctx* c, unsigned long long* out) {
  if (!c || !out) return RTOC_ERR_BAD_ARG;
  *out = c->graph_replays;
  return RTOC_OK;
}

int rtoc_riccati_sweep(rtoc_ctx* c) {
  CHECK_READY(c);
  return run_graphed(c, &c->g_sweep, 0.0, 0.0, [&]() { return launch_sweep(c); });
}

static int launch_fill(rtoc_ctx* c, double dt) {
  FillArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.dt = dt;
  a.kl = c->L.kkt;
  hipLaunchKernelGGL(c->ks->fill, dim3(c->batch * c->nstages), dim3(128), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// UnconstrRiccatiRecursion in its structured form (unconstr_riccati.hpp): whenever the shape has the kernels and the
// horizon scan is not asked for (the scan works on the general elements, i.e. on materialised A, B)
static bool unconstr_structured(const rtoc_ctx* c) {
  return c->ks->ubwd != nullptr && c->dims.nf_max == 0 && !scan_applies(c) && !c->unconstr_dense;
}
static int launch_unconstr_riccati(rtoc_ctx* c, double dt, bool forward) {
  int rc = ensure_buffer(c, RTOC_BUF_RIC);
  if (!rc && forward) rc = ensure_buffer(c, RTOC_BUF_DIR);
  if (rc) return rc;
  if (!c->buf[RTOC_BUF_KKT]) return RTOC_ERR_NOT_READY;
  UrArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.kkt_rw = c->buf[RTOC_BUF_KKT];
  a.ric = c->buf[RTOC_BUF_RIC];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.dx0 = c->buf[RTOC_BUF_DX0];
  a.status = c->d_status;
  a.nstages = c->nstages, a.batch = c->batch, a.writeback = c->writeback;
  a.dt = dt;
  a.kl = c->L.kkt, a.rl = c->L.ric, a.dl = c->L.dir;
  hipLaunchKernelGGL(forward ? c->ks->ufwd : c->ks->ubwd, dim3(c->batch), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_unconstr_backward(rtoc_ctx* c, double dt) {
  CHECK_READY(c);
  if (c->dims.nu != c->dims.nv || !(dt > 0.0)) return RTOC_ERR_BAD_ARG;
  if (unconstr_structured(c)) return launch_unconstr_riccati(c, dt, false);
  int rc = launch_fill(c, dt);
  if (rc) return rc;
  return launch_backward(c);
}

static int launch_unconstr_dynamics(rtoc_ctx* c, bool expand, double dt) {
  if (c->dims.nu != c->dims.nv || c->dims.nf_max != 0) return RTOC_ERR_BAD_ARG;
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  UdArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.dt = dt;
  a.kl = c->L.kkt;
  a.cl = c->L.cdd;
  a.dl = c->L.dir;
  hipLaunchKernelGGL(expand ? c->ks->uexp : c->ks->ucond, dim3(c->batch * (c->nstages - 1)), dim3(64), 0,
                     c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_unconstr_condense(rtoc_ctx* c) {
  CHECK_READY(c);
  return launch_unconstr_dynamics(c, false, 1.0);
}

int rtoc_unconstr_expand(rtoc_ctx* c, double dt) {
  CHECK_READY(c);
  if (!(dt > 0.0)) return RTOC_ERR_BAD_ARG;
  return launch_unconstr_dynamics(c, true, dt);
}

int rtoc_unconstr_forward(rtoc_ctx* c, double dt) {
  CHECK_READY(c);
  if (c->dims.nu != c->dims.nv || !(dt > 0.0)) return RTOC_ERR_BAD_ARG;
  if (unconstr_structured(c)) return launch_unconstr_riccati(c, dt, true);
  return launch_forward(c);
}

int rtoc_expand(rtoc_ctx* c, double tau) {
  CHECK_READY(c);
  if (!(tau > 0.0 && tau <= 1.0)) return RTOC_ERR_BAD_ARG;
  int rc = launch_expand(c, tau);
  if (!rc && c->cone_contacts > 0) rc = launch_cones(c, 1, tau);
  if (!rc && c->buf[RTOC_BUF_SE3] && c->dims.np == 6) rc = launch_state_correction(c, 1);
  return rc;
}

int rtoc_update(rtoc_ctx* c) {
  CHECK_READY(c);
  if (c->cone_contacts > 0) {
    int rc = launch_cones(c, 2, 0.0);
    if (rc) return rc;
  }
  if (c->nrows == 0) return RTOC_OK;
  UpdArgs a;
  a.con = c->buf[RTOC_BUF_CON];
  a.rows = c->d_rows;
  a.grid = c->d_grid;
  a.steps = c->buf[RTOC_BUF_STEP];
  a.nrows = c->nrows;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nl = c->L.con;
  hipLaunchKernelGGL(pdipm_update_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int set_cones(rtoc_ctx* c, int max_contacts, int contact_dim, int rows_per_contact, size_t stride) {
  if (!c || max_contacts < 0) return RTOC_ERR_BAD_ARG;
  if (max_contacts == 0) {
    c->cone_contacts = 0;
    c->cone_rows = 0;
    return RTOC_OK;
  }
  if ((contact_dim != 3 && contact_dim != 6) || max_contacts * contact_dim > c->dims.nf_max ||
      c->nrows + rows_per_contact * max_contacts > c->dims.nc_max || rows_per_contact * max_contacts > 64)
    return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  const size_t need = (size_t)c->batch * c->max_stages * stride;
  if (c->buf[RTOC_BUF_CONE] && c->count[RTOC_BUF_CONE] != need) {
    if (c->owned[RTOC_BUF_CONE]) (void)hipFree(c->buf[RTOC_BUF_CONE]);
    c->buf[RTOC_BUF_CONE] = nullptr;
  }
  c->count[RTOC_BUF_CONE] = need;
  int rc = ensure_buffer(c, RTOC_BUF_CONE);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_CON);
  if (rc) return rc;
  c->cone_contacts = max_contacts;
  c->cone_dim = contact_dim;
  c->cone_rows = rows_per_contact;
  c->epoch++;
  return RTOC_OK;
}

int rtoc_set_friction_cones(rtoc_ctx* c, int max_contacts, int contact_dim) {
  if (!c) return RTOC_ERR_BAD_ARG;
  return set_cones(c, max_contacts, contact_dim, RTOC_FRICTION_ROWS, rtoc_cone_stride(c->dims.nv, max_contacts));
}

int rtoc_set_wrench_cones(rtoc_ctx* c, int max_contacts) {
  if (!c) return RTOC_ERR_BAD_ARG;
  return set_cones(c, max_contacts, 6, RTOC_WRENCH_ROWS, rtoc_wrench_cone_stride(max_contacts));
}

int rtoc_wrench_cone_matrix(double X, double Y, double mu, double* out) {
  if (!out || !(X > 0.0) || !(Y > 0.0) || !(mu > 0.0)) return RTOC_ERR_BAD_ARG;  // ctor checks :19-26
  // Rows: unilaterality; four friction-pyramid faces; centre of pressure inside the sole (tau_x, tau_y);
  // eight yaw-torque bounds -- one per sign pattern (sx, sy, sz) of the f_x, f_y and tau_z coefficients.
  const double xymu = (X + Y) * mu;
  double row[RTOC_WRENCH_ROWS][6] = {{0, 0, -1, 0, 0, 0},   {-1, 0, -mu, 0, 0, 0}, {1, 0, -mu, 0, 0, 0},
                                     {0, -1, -mu, 0, 0, 0}, {0, 1, -mu, 0, 0, 0},  {0, 0, -Y, -1, 0, 0},
                                     {0, 0, -Y, 1, 0, 0},   {0, 0, -X, 0, -1, 0},  {0, 0, -X, 0, 1, 0}};
  // (sign of Y f_x, sign of X f_y) for rows 9..12; rows 13..16 mirror them with tau_z = +1
  static const int sg[4][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
  for (int i = 0; i < 4; ++i) {
    double* lo = row[9 + i];
    double* hi = row[13 + i];
    lo[0] = sg[i][0] * Y;  lo[1] = sg[i][1] * X;  lo[2] = -xymu;
    lo[3] = -sg[i][0] * mu; lo[4] = -sg[i][1] * mu; lo[5] = -1;
    hi[0] = -sg[i][0] * Y; hi[1] = -sg[i][1] * X; hi[2] = -xymu;
    hi[3] = -sg[i][0] * mu; hi[4] = -sg[i][1] * mu; hi[5] = 1;
  }
  for (int j = 0; j < RTOC_WRENCH_ROWS; ++j)
    for (int m = 0; m < 6; ++m) out[j + RTOC_WRENCH_ROWS * m] = row[j][m];
  return RTOC_OK;
}

int rtoc_set_constraint_rows(rtoc_ctx* c, const rtoc_box_row* rows, int nrows) {
  if (!c || nrows < 0 || nrows + c->cone_rows * c->cone_contacts > c->dims.nc_max || (nrows > 0 && !rows))
    return RTOC_ERR_BAD_ARG;
  for (int r = 0; r < nrows; ++r) {
    const rtoc_box_row& w = rows[r];
    const int lim = (w.var == RTOC_VAR_U) ? c->dims.nu : c->dims.nv;
    if (w.var < 0 || w.var > RTOC_VAR_A || w.index < 0 || w.index >= lim || (w.sign != 1 && w.sign != -1) ||
        w.level < 0 || w.level > 2)
      return RTOC_ERR_BAD_ARG;
    // acceleration limits are acceleration-level rows of the contact path (the unconstrained path has no `a` beside its control)
    if (w.var == RTOC_VAR_A && (w.level != 0 || c->dims.nf_max == 0)) return RTOC_ERR_BAD_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (nrows > 0) {
    int rc = ensure_buffer(c, RTOC_BUF_CON);
    if (rc) return rc;
    if (!c->d_rows) HIP_TRY(hipMalloc((void**)&c->d_rows, sizeof(rtoc_box_row) * c->dims.nc_max));
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, sizeof(rtoc_box_row) * nrows, hipMemcpyHostToDevice, c->stream));
    // rows grouped by the primal entry they act on (ascending row index inside a group, i.e. the
    // order in which the reference's components touch that entry)
    // primal entries: q (nv), v (nv), u (nu), a (nv)
    const int nv = c->dims.nv, ne = 3 * nv + c->dims.nu;
    std::vector<int> csr(ne + 1 + nrows, 0);
    auto entry_of = [&](const rtoc_box_row& w) {
      return w.var == RTOC_VAR_Q ? w.index : (w.var == RTOC_VAR_V ? nv + w.index : (w.var == RTOC_VAR_U ? 2 * nv + w.index : 2 * nv + c->dims.nu + w.index));
    };
    for (int r = 0; r < nrows; ++r) csr[entry_of(rows[r]) + 1]++;
    for (int e = 0; e < ne; ++e) csr[e + 1] += csr[e];
    std::vector<int> fill(csr.begin(), csr.begin() + ne);
    for (int r = 0; r < nrows; ++r) csr[ne + 1 + fill[entry_of(rows[r])]++] = r;
    if (!c->d_entry)
      HIP_TRY(hipMalloc((void**)&c->d_entry, sizeof(int) * (ne + 1 + c->dims.nc_max)));
    HIP_TRY(hipMemcpyAsync(c->d_entry, csr.data(), sizeof(int) * csr.size(), hipMemcpyHostToDevice, c->stream));
    // the first two rows of every entry, packed (condense.hpp)
    std::vector<int> pair(4 * (size_t)ne, -1);
    for (int e = 0; e < ne; ++e)
      for (int k = 0; k < 2 && csr[e] + k < csr[e + 1]; ++k) {
        const int r = csr[ne + 1 + csr[e] + k];
        pair[4 * e + k] = r;
        pair[4 * e + 2 + k] = (rows[r].sign & 0xff) | (rows[r].level << 8);
      }
    if (!c->d_pair) HIP_TRY(hipMalloc((void**)&c->d_pair, sizeof(int) * 4 * ne));
    HIP_TRY(hipMemcpyAsync(c->d_pair, pair.data(), sizeof(int) * pair.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (nrows > 0) {
    if (!c->h_rows) c->h_rows = (rtoc_box_row*)malloc(sizeof(rtoc_box_row) * c->dims.nc_max);
    if (c->h_rows) memcpy(c->h_rows, rows, sizeof(rtoc_box_row) * nrows);
  }
  c->nrows = nrows;
  c->epoch++;
  return RTOC_OK;
}

int rtoc_check_fxx_structure(rtoc_ctx* c, int* structured) {
  CHECK_READY(c);
  int rc = check_fxx(c);
  if (rc) return rc;
  if (structured) *structured = c->fxx_state == 1;
  return RTOC_OK;
}

int rtoc_status(rtoc_ctx* c, uint32_t* host_flags, int count) {
  if (!c || !host_flags || count < 0 || count > c->batch) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(host_flags, c->d_status, sizeof(uint32_t) * count, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_clear_status(rtoc_ctx* c) {
  if (!c) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t) * c->batch, c->stream));
  return RTOC_OK;
}

// Tuning aid (not part of the drop-in surface): attach a device buffer of nstages*16 int64 that
// block 0 of the backward kernel fills with phase cycle stamps; nullptr detaches.
int rtoc_debug_profile(rtoc_ctx* c, long long* host_out) {
  if (!c) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = (size_t)c->max_stages * 32;
  if (!c->d_prof) {
    HIP_TRY(hipMalloc((void**)&c->d_prof, n * sizeof(long long)));
    HIP_TRY(hipMemsetAsync(c->d_prof, 0, n * sizeof(long long), c->stream));
    return RTOC_OK;
  }
  if (host_out) {
    HIP_TRY(hipMemcpyAsync(host_out, c->d_prof, n * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return RTOC_OK;
}

int rtoc_sync(rtoc_ctx* c) {
  if (!c) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_newton_iteration(rtoc_ctx* c, double kkt_tol, double tau);
int rtoc_linearize_contact_dynamics(rtoc_ctx* c, int augment_residual);
int rtoc_time_phase(rtoc_ctx* c, int phase, int reps, float* ms) {
  CHECK_READY(c);
  if (!ms || reps < 1 || phase < 0 || phase > 8) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < reps; ++r) {
    int rc = RTOC_OK;
    switch (phase) {
      case 0: rc = launch_backward(c); break;
      case 1: rc = launch_forward(c); break;
      case 2: rc = rtoc_condense(c); break;      // incl. cone rows / state-equation correction if set
      case 3: rc = rtoc_expand(c, 0.995); break;
      case 4: rc = launch_sweep(c); break;
      case 5: rc = rtoc_update(c); break;
      case 6: rc = rtoc_newton_iteration(c, 0.0, 0.995); break;  // the whole iteration as one launch sequence
      case 7: rc = rtoc_linearize_contact_dynamics(c, 0); break;
      case 8: rc = rtoc_linearize_contact_dynamics(c, 1); break;
    }
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
  *ms = t / reps;
  return RTOC_OK;
}

// ---- SplitSolution::integrate ---------------------------------------------------------------
int rtoc_integrate_solution(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  IntArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.steps = c->buf[RTOC_BUF_STEP];
  a.grid = c->d_grid;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nv = c->dims.nv;
  a.nu = c->dims.nu;
  a.np = c->dims.np;
  a.nf_max = c->dims.nf_max;
  a.ns_max = c->dims.ns_max;
  a.sl = c->L.sol;
  a.dl = c->L.dir;
  hipLaunchKernelGGL(integrate_solution_kernel, dim3(c->batch * c->nstages), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// ---- rigid-body linearisation (include/rtoc_robot.h) ------------------------------------------
// The checks of a robot-model table that do not depend on a context, and everything the kernels precompute from it (tree levels,
// packed constants, the tangent walk's storage plan and passes).  *out: a new DevModel (caller deletes) or nullptr.
static int build_dev_model(const rtoc_robot_model* m, int forced_dpp, rbd::DevModel** out, int* max_dimf_out) {
  *out = nullptr;
  if (!m) return RTOC_ERR_BAD_ARG;
  if (m->njoints < 1 || m->njoints > RTOC_MAX_JOINTS || m->ncontacts < 0 || m->ncontacts > RTOC_MAX_CONTACTS) return RTOC_ERR_BAD_ARG;
  int max_dimf = 0;
  for (int k = 0; k < m->ncontacts; ++k) {
    if (m->contact_type[k] != RTOC_CONTACT_POINT && m->contact_type[k] != RTOC_CONTACT_SURFACE) return RTOC_ERR_BAD_ARG;
    if (k > 0 && m->contact_type[k] < m->contact_type[k - 1]) return RTOC_ERR_BAD_ARG;  // points first
    max_dimf += m->contact_type[k] == RTOC_CONTACT_SURFACE ? 6 : 3;
  }
  if (max_dimf_out) *max_dimf_out = max_dimf;
  const bool ff = m->type[0] == RTOC_JOINT_FREE_FLYER;
  if (m->nv < 1 || m->nv > RTOC_MAX_JOINTS + 8 || m->nq != m->nv + (ff ? 1 : 0)) return RTOC_ERR_BAD_ARG;
  rbd::DevModel* h = new (std::nothrow) rbd::DevModel;
  if (!h) return RTOC_ERR_HIP;
  h->m = *m;
  // depth-first order: when joint i is visited, the joint open one level up must be its parent
  int open[RTOC_MAX_JOINTS], iq = 0, iv = 0, nlev = 0;
  for (int k = 0; k < RTOC_MAX_JOINTS; ++k) open[k] = -1;
  bool ok = true;
  for (int i = 0; i < m->njoints && ok; ++i) {
    const int par = m->parent[i];
    ok = par < i && par >= -1 && (m->type[i] == RTOC_JOINT_REVOLUTE || (m->type[i] == RTOC_JOINT_FREE_FLYER && i == 0 && par == -1));
    if (!ok) break;
    const int d = par < 0 ? 0 : h->depth[par] + 1;
    ok = (d == 0 || open[d - 1] == par) && m->idx_q[i] == iq && m->idx_v[i] == iv;
    h->depth[i] = d;
    open[d] = i;
    for (int k = d + 1; k < RTOC_MAX_JOINTS; ++k) open[k] = -1;   // the levels below are closed for good
    nlev = d + 1 > nlev ? d + 1 : nlev;
    iq += m->type[i] == RTOC_JOINT_FREE_FLYER ? 7 : 1;
    iv += m->type[i] == RTOC_JOINT_FREE_FLYER ? 6 : 1;
  }
  ok = ok && iq == m->nq && iv == m->nv;
  for (int k = 0; k < m->ncontacts && ok; ++k) ok = m->contact_parent[k] >= 0 && m->contact_parent[k] < m->njoints;
  if (!ok) {
    delete h;
    return RTOC_ERR_BAD_ARG;
  }
  h->nlevels = nlev;
  rbd::pack_model(h);
  if (forced_dpp) rbd::plan_passes(h, forced_dpp);
  // (the walk's plan word has four bits for a forward-tangent slot: at most 14 branching bodies on a root-to-leaf path)
  if (h->nbranch > 14 || rbd::lin_lds_bytes(nlev, h->nbranch, m->njoints, m->ncontacts, m->nv, h->dpp, false) > 160 * 1024) {
    delete h;
    return RTOC_ERR_BAD_ARG;
  }
  *out = h;
  return RTOC_OK;
}

int rtoc_robot_model_plan(const rtoc_robot_model* m, int forced_dofs_per_pass, rtoc_linearize_plan* plan, unsigned long long* pass_bodies) {
  if (!plan || forced_dofs_per_pass < 0 || forced_dofs_per_pass > rbd::LIN_MAX_DPP) return RTOC_ERR_BAD_ARG;
  rbd::DevModel* h = nullptr;
  const int rc = build_dev_model(m, forced_dofs_per_pass, &h, nullptr);
  if (rc) return rc;
  plan->nlevels = h->nlevels, plan->nbranch = h->nbranch, plan->dofs_per_pass = h->dpp, plan->npass = h->npass;
  plan->lds_bytes = (int)rbd::lin_lds_bytes(h->nlevels, h->nbranch, m->njoints, m->ncontacts, m->nv, h->dpp, true);
  if (pass_bodies)
    for (int p = 0; p < RTOC_MAX_JOINTS + 8; ++p) pass_bodies[p] = p < h->npass ? h->pass_bodies[p] : 0ull;
  delete h;
  return RTOC_OK;
}

int rtoc_set_robot_model(rtoc_ctx* c, const rtoc_robot_model* m) {
  if (!c || !m) return RTOC_ERR_BAD_ARG;
  rbd::DevModel* h = nullptr;
  int max_dimf = 0;
  const int brc = build_dev_model(m, c->lin_dpp, &h, &max_dimf);
  if (brc) return brc;
  const bool ff = m->type[0] == RTOC_JOINT_FREE_FLYER;
  if (m->nv != c->dims.nv || max_dimf > c->dims.nf_max || (ff ? m->nv - 6 : m->nv) != c->dims.nu) {
    delete h;
    return RTOC_ERR_BAD_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_model) HIP_TRY(hipMalloc((void**)&c->d_model, sizeof(rbd::DevModel)));
  delete c->h_model;
  c->h_model = h;
  HIP_TRY(hipMemcpyAsync(c->d_model, h, sizeof(rbd::DevModel), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(set_linearize_lds(*m, h->nlevels, h->nbranch, h->dpp));
  c->epoch++;
  return RTOC_OK;
}

int rtoc_set_contact_schedule(rtoc_ctx* c, const unsigned* active, const double* positions, const double* rotations) {
  CHECK_READY(c);
  if (!c->h_model || !active) return RTOC_ERR_BAD_ARG;
  const rtoc_robot_model& m = c->h_model->m;
  const int nc = m.ncontacts;
  for (int i = 0; i < c->nstages; ++i) {
    if (nc < 32 && (active[i] >> nc) != 0) return RTOC_ERR_BAD_ARG;
    int rows = 0;
    for (int k = 0; k < nc; ++k)
      if ((active[i] >> k) & 1u) rows += m.contact_type[k] == RTOC_CONTACT_SURFACE ? 6 : 3;
    if (i < c->nstages - 1 && rows != c->h_grid[i].dimf) return RTOC_ERR_BAD_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_active) HIP_TRY(hipMalloc((void**)&c->d_active, sizeof(unsigned) * c->max_stages));
  if (!c->d_cpos) HIP_TRY(hipMalloc((void**)&c->d_cpos, sizeof(double) * c->max_stages * RTOC_MAX_CONTACTS * 3));
  if (rotations && !c->d_crot) HIP_TRY(hipMalloc((void**)&c->d_crot, sizeof(double) * c->max_stages * RTOC_MAX_CONTACTS * 9));
  HIP_TRY(hipMemcpyAsync(c->d_active, active, sizeof(unsigned) * c->nstages, hipMemcpyHostToDevice, c->stream));
  if (positions)
    HIP_TRY(hipMemcpyAsync(c->d_cpos, positions, sizeof(double) * c->nstages * nc * 3, hipMemcpyHostToDevice, c->stream));
  if (rotations)
    HIP_TRY(hipMemcpyAsync(c->d_crot, rotations, sizeof(double) * c->nstages * nc * 9, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->has_cpos = positions != nullptr;
  c->has_crot = rotations != nullptr;
  c->epoch++;
  return RTOC_OK;
}

// rbd_values_kernel for the iterate in RTOC_BUF_SOL: the lane-invariant values of the rigid-body recursion per body (and the
// ID rows of RTOC_CDD_IDC), read by the tangent walk and by the friction-cone rows
// scratch of the values pre-pass (hipFree / hipMalloc synchronise the device: callers that fork a stream call this first)
static int ensure_rbd_values(rtoc_ctx* c) {
  if (!c->h_model) return RTOC_ERR_NOT_READY;
  const rtoc_robot_model& m = c->h_model->m;
  bool any_impact = false;
  for (int i = 0; i + 1 < c->nstages; ++i) any_impact = any_impact || c->h_grid[i].type == RTOC_GRID_IMPACT;
  const size_t need = (size_t)c->batch * c->max_stages * m.njoints * rbd::VAL_SLOTS;
  if (c->vals_cap < need) {
    if (c->d_vals) (void)hipFree(c->d_vals);
    if (c->d_vals2) (void)hipFree(c->d_vals2);
    c->d_vals = c->d_vals2 = nullptr;
    c->vals_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->d_vals, need * sizeof(double)));
    c->vals_cap = need;
  }
  if (any_impact && !c->d_vals2) HIP_TRY(hipMalloc((void**)&c->d_vals2, c->vals_cap * sizeof(double)));
  return RTOC_OK;
}

static int launch_rbd_values(rtoc_ctx* c, bool unconstr) {
  if (!c->h_model || !c->d_active || !c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  if (c->nstages < 2) return RTOC_OK;
  const rtoc_robot_model& m = c->h_model->m;
  bool any_impact = false;
  for (int i = 0; i + 1 < c->nstages; ++i) any_impact = any_impact || c->h_grid[i].type == RTOC_GRID_IMPACT;
  rc = ensure_rbd_values(c);
  if (rc) return rc;
  rbd::ValArgs v;
  v.model = c->d_model, v.sol = c->buf[RTOC_BUF_SOL], v.cdd = c->buf[RTOC_BUF_CDD], v.grid = c->d_grid, v.active = c->d_active;
  v.nstages = c->nstages, v.batch = c->batch, v.nv = m.nv, v.njoints = m.njoints, v.ncontacts = m.ncontacts;
  v.nu = m.type[0] == RTOC_JOINT_FREE_FLYER ? m.nv - 6 : m.nv;
  v.nlevels = c->h_model->nlevels, v.unconstr = unconstr ? 1 : 0;
  v.gs = 1;
  while (v.gs < m.njoints) v.gs *= 2;
  v.sol_stride = c->L.sol.stride, v.cdd_stride = c->L.cdd.stride;
  v.o_q = c->L.sol.off[RTOC_SOL_Q], v.o_v = c->L.sol.off[RTOC_SOL_V], v.o_a = c->L.sol.off[RTOC_SOL_A];
  v.o_u = c->L.sol.off[RTOC_SOL_U], v.o_f = c->L.sol.off[RTOC_SOL_F], v.o_idc = c->L.cdd.off[RTOC_CDD_IDC];
  v.gx = m.gravity[0], v.gy = m.gravity[1], v.gz = m.gravity[2];
  v.positions = c->has_cpos ? c->d_cpos : nullptr, v.rotations = c->has_crot ? c->d_crot : nullptr;
  const int G = 64 / v.gs;
  const long long items = (long long)c->batch * (c->nstages - 1);
  const size_t vlds = sizeof(double) * G * m.njoints * rbd::VAL_SLOTS;
  for (int trav = 0; trav < (any_impact ? 2 : 1); ++trav) {
    v.trav = trav;
    v.vals = trav == 0 ? c->d_vals : c->d_vals2;
    v.nsel = 0;
    long long n = items;
    if (trav == 1) {   // the kinematics traversal exists on impact grids only: launch just those (if they fit the list)
      int k = 0;
      for (int i = 0; i + 1 < c->nstages && k <= 16; ++i)
        if (c->h_grid[i].type == RTOC_GRID_IMPACT) {
          if (k < 16) v.sel[k] = i;
          ++k;
        }
      if (k <= 16) v.nsel = k, n = (long long)c->batch * k;
    }
    hipLaunchKernelGGL(rbd::rbd_values_kernel, dim3((unsigned)((n + G - 1) / G)), dim3(64), vlds, c->stream, v);
  }
  HIP_TRY(hipGetLastError());
  c->vals_fresh = 1;
  return RTOC_OK;
}

static int launch_linearize(rtoc_ctx* c, int augment_residual, bool unconstr, double scale) {
  if (!c->h_model || !c->d_active || !c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  if (augment_residual && !c->buf[RTOC_BUF_KKT]) return RTOC_ERR_NOT_READY;
  int rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  rbd::LinArgs a;
  a.model = c->d_model;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.grid = c->d_grid;
  a.active = c->d_active;
  a.positions = c->has_cpos ? c->d_cpos : nullptr;
  a.rotations = c->has_crot ? c->d_crot : nullptr;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.sol_stride = c->L.sol.stride;
  a.cdd_stride = c->L.cdd.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q];
  a.o_v = c->L.sol.off[RTOC_SOL_V];
  a.o_a = c->L.sol.off[RTOC_SOL_A];
  a.o_u = c->L.sol.off[RTOC_SOL_U];
  a.o_f = c->L.sol.off[RTOC_SOL_F];
  a.o_idc = c->L.cdd.off[RTOC_CDD_IDC];
  a.o_didda = c->L.cdd.off[RTOC_CDD_DIDDA];
  a.o_dcda = c->L.cdd.off[RTOC_CDD_DCDA];
  a.o_didcdqv = c->L.cdd.off[RTOC_CDD_DIDCDQV];
  a.ldv = c->dims.nv + c->dims.nf_max;
  a.nf_max = c->dims.nf_max;
  {
    const rtoc_robot_model& m = c->h_model->m;
    a.nlevels = c->h_model->nlevels, a.nbranch = c->h_model->nbranch, a.dpp = c->h_model->dpp;
    a.nv = m.nv, a.nq = m.nq, a.njoints = m.njoints, a.ncontacts = m.ncontacts;
    a.nu = m.type[0] == RTOC_JOINT_FREE_FLYER ? m.nv - 6 : m.nv;
    a.gx = m.gravity[0], a.gy = m.gravity[1], a.gz = m.gravity[2];
  }
  a.kkt = augment_residual ? c->buf[RTOC_BUF_KKT] : nullptr;
  a.kkt_stride = c->L.kkt.stride;
  a.o_lx = c->L.kkt.off[RTOC_KKT_LX];
  a.o_lu = c->L.kkt.off[RTOC_KKT_LU];
  a.o_la = c->L.cdd.off[RTOC_CDD_LA];
  a.o_lf = c->L.cdd.off[RTOC_CDD_LF];
  a.o_lup = c->L.cdd.off[RTOC_CDD_LUP];
  a.o_beta = c->L.sol.off[RTOC_SOL_BETA];
  a.o_mu = c->L.sol.off[RTOC_SOL_MU];
  a.o_nup = c->L.sol.off[RTOC_SOL_NUP];
  a.unconstr = unconstr ? 1 : 0;
  a.scale = scale;
  if (c->nstages < 2) return RTOC_OK;
  const size_t lds = rbd::lin_lds_bytes(c->h_model->nlevels, c->h_model->nbranch, c->h_model->m.njoints, c->h_model->m.ncontacts, c->h_model->m.nv, c->h_model->dpp, !c->linearize_fused);
  const bool surf = model_has_surface_contacts(c->h_model->m);
  a.vals = a.vals2 = nullptr;
  if (!c->linearize_fused) {
    // the values of the recursion first (level-parallel, lanes = bodies), then the tangent walk reads them (rigid_body.hpp)
    if (!c->vals_fresh) {
      const int rv = launch_rbd_values(c, unconstr);
      if (rv) return rv;
    }
    c->vals_fresh = 0;
    a.vals = c->d_vals, a.vals2 = c->d_vals2;
    if (surf)
      hipLaunchKernelGGL((rbd::linearize_contact_dynamics_kernel<true, true>), dim3(c->batch * (c->nstages - 1)), dim3(64), lds, c->stream, a);
    else
      hipLaunchKernelGGL((rbd::linearize_contact_dynamics_kernel<false, true>), dim3(c->batch * (c->nstages - 1)), dim3(64), lds, c->stream, a);
  } else if (surf) {
    hipLaunchKernelGGL(rbd::linearize_contact_dynamics_kernel<true>, dim3(c->batch * (c->nstages - 1)), dim3(64), lds, c->stream, a);
  } else {
    hipLaunchKernelGGL(rbd::linearize_contact_dynamics_kernel<false>, dim3(c->batch * (c->nstages - 1)), dim3(64), lds, c->stream, a);
  }
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_linearize_contact_dynamics(rtoc_ctx* c, int augment_residual) {
  CHECK_READY(c);
  return launch_linearize(c, augment_residual, false, 1.0);
}

// ---- the unconstrained (fixed-base, contact-free) solver iteration closed on the device -------
int rtoc_set_configuration_cost(rtoc_ctx* c, const rtoc_configuration_cost* cost) {
  if (!c || !cost) return RTOC_ERR_BAD_ARG;
  const int nv = c->dims.nv, M = nv + 1;
  if (M > RTOC_MAX_JOINTS || (c->dims.np != 0 && c->dims.np != 6)) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  std::vector<double> h((size_t)12 * M, 0.0);
  const double* src[12] = {cost->q_ref, cost->v_ref, cost->u_ref, cost->q_weight, cost->v_weight, cost->a_weight, cost->u_weight,
                           cost->q_weight_terminal, cost->v_weight_terminal, cost->q_weight_impact, cost->v_weight_impact,
                           cost->dv_weight_impact};
  for (int k = 0; k < 12; ++k) {
    const int n = k == 0 ? nv + (c->dims.np == 6 ? 1 : 0) : (k == 2 || k == 6 ? c->dims.nu : nv);
    for (int i = 0; i < n; ++i) {
      if (k >= 3 && !(src[k][i] >= 0.0)) return RTOC_ERR_BAD_ARG;  // configuration_space_cost.cpp: weights must be non-negative
      h[(size_t)k * M + i] = src[k][i];
    }
  }
  if (!c->d_cost) HIP_TRY(hipMalloc((void**)&c->d_cost, sizeof(double) * 12 * M));
  HIP_TRY(hipMemcpyAsync(c->d_cost, h.data(), sizeof(double) * 12 * M, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_set_initial_state(rtoc_ctx* c, const double* x0, int count) {
  if (!c || !x0 || count != c->batch) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  const int nq = c->dims.nv + (c->dims.np == 6 ? 1 : 0);  // a free-flyer base carries a quaternion
  const size_t n = (size_t)c->batch * (nq + c->dims.nv);
  if (!c->d_x0) HIP_TRY(hipMalloc((void**)&c->d_x0, sizeof(double) * n));
  HIP_TRY(hipMemcpyAsync(c->d_x0, x0, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

// linearizeStateEquation / linearizeImpactStateEquation of every non-terminal grid point (state_equation_lin.hpp)
static int launch_state_equation(rtoc_ctx* c, bool zeroed);
int rtoc_linearize_state_equation(rtoc_ctx* c) { return launch_state_equation(c, false); }
// zeroed: the caller has just zeroed the KKT records (rtoc_contact_eval_kkt) -- only the non-zero entries of the Fxx top
// half are written, and the records are known to have the structure RTOC_OPT_FXX_STRUCTURE's check would find
static int launch_state_equation(rtoc_ctx* c, bool zeroed) {
  CHECK_READY(c);
  if (!c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  if (c->dims.np != 0 && c->dims.np != 6) return RTOC_ERR_BAD_ARG;
  int rc = ensure_buffer(c, RTOC_BUF_KKT);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_DX0);
  if (!rc && c->dims.np == 6) rc = ensure_buffer(c, RTOC_BUF_SE3);
  if (rc) return rc;
  SeLinArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.x0 = c->d_x0;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.se3 = c->dims.np == 6 ? c->buf[RTOC_BUF_SE3] : nullptr;
  a.dx0 = c->d_x0 ? c->buf[RTOC_BUF_DX0] : nullptr;
  a.grid = c->d_grid;
  a.nstages = c->nstages, a.batch = c->batch, a.nv = c->dims.nv, a.floating = c->dims.np == 6;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_v = c->L.sol.off[RTOC_SOL_V], a.o_a = c->L.sol.off[RTOC_SOL_A];
  a.o_lmd = c->L.sol.off[RTOC_SOL_LMD], a.o_gmm = c->L.sol.off[RTOC_SOL_GMM];
  a.o_fxx = c->L.kkt.off[RTOC_KKT_FXX], a.o_fx = c->L.kkt.off[RTOC_KKT_FX], a.o_lx = c->L.kkt.off[RTOC_KKT_LX];
  a.o_hx = c->L.kkt.off[RTOC_KKT_HX], a.o_ffx = c->L.kkt.off[RTOC_KKT_FFX], a.o_scal = c->L.kkt.off[RTOC_KKT_SCAL];
  a.o_la = c->L.cdd.off[RTOC_CDD_LA], a.o_ha = c->L.cdd.off[RTOC_CDD_HA];
  a.zeroed = zeroed ? 1 : 0;
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  hipLaunchKernelGGL(state_equation_lin_kernel, dim3(c->batch * c->nstages), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  if (zeroed && c->fxx_last != 1) c->epoch++;  // the kernel choice of the backward recursion is part of a captured graph
  c->fxx_state = zeroed ? 1 : 0;
  if (zeroed) c->fxx_last = 1;
  return RTOC_OK;
}

int rtoc_set_constraint_bounds(rtoc_ctx* c, const double* bounds, int nrows, double barrier_param, double fraction_to_boundary_rule) {
  if (!c || !bounds || nrows != c->nrows || nrows <= 0) return RTOC_ERR_BAD_ARG;
  if (!(barrier_param > 0.0) || !(fraction_to_boundary_rule > 0.0) || !(fraction_to_boundary_rule < 1.0)) return RTOC_ERR_BAD_ARG;  // constraints.cpp setters
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_bounds) HIP_TRY(hipMalloc((void**)&c->d_bounds, sizeof(double) * c->dims.nc_max));
  HIP_TRY(hipMemcpyAsync(c->d_bounds, bounds, sizeof(double) * nrows, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->barrier = barrier_param;
  c->ftb_rule = fraction_to_boundary_rule;
  return RTOC_OK;
}

static int launch_ubox(rtoc_ctx* c, int mode, bool contact = false) {
  UboxArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = c->buf[RTOC_BUF_CON];
  a.dir = c->buf[RTOC_BUF_DIR];
  a.rows = c->d_rows;
  a.entry = c->d_entry;
  a.bounds = c->d_bounds;
  a.grid = c->d_grid;
  a.steps = (unsigned long long*)c->buf[RTOC_BUF_STEP];
  a.nstages = c->nstages, a.batch = c->batch, a.nrows = c->nrows, a.nv = c->dims.nv, a.nu = c->dims.nu, a.mode = mode;
  a.barrier = c->barrier, a.tau = c->ftb_rule;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride;
  a.con_stride = c->L.con.stride, a.dir_stride = c->L.dir.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_v = c->L.sol.off[RTOC_SOL_V], a.o_u = c->L.sol.off[RTOC_SOL_U], a.o_a = c->L.sol.off[RTOC_SOL_A];
  a.o_qxx = c->L.kkt.off[RTOC_KKT_QXX], a.o_lx = c->L.kkt.off[RTOC_KKT_LX];
  a.o_qaa = c->L.cdd.off[RTOC_CDD_QAA], a.o_la = c->L.cdd.off[RTOC_CDD_LA];
  a.o_dx = c->L.dir.off[RTOC_DIR_DX], a.o_du = c->L.dir.off[RTOC_DIR_DU];
  a.nl = c->L.con;
  a.contact = contact ? 1 : 0;
  a.q_shift = (contact && c->dims.np == 6) ? 1 : 0;
  a.o_lu = c->L.kkt.off[RTOC_KKT_LU];
  hipLaunchKernelGGL(unconstr_box_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}
static bool ubox_on(const rtoc_ctx* c) { return c->nrows > 0 && c->d_bounds != nullptr; }

// UnconstrOCPSolver::initConstraints (unconstr_ocp_solver.cpp:91-93): setSlackAndDual of every row at the current iterate
int rtoc_unconstr_init_constraints(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!ubox_on(c) || !c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  if (c->dims.nu != c->dims.nv || c->dims.nf_max != 0) return RTOC_ERR_BAD_ARG;
  int rc = ensure_buffer(c, RTOC_BUF_CON);
  if (rc) return rc;
  return launch_ubox(c, UBOX_INIT);
}

int rtoc_unconstr_eval_kkt(rtoc_ctx* c, double dt) {
  CHECK_READY(c);
  if (!(dt > 0.0) || c->dims.nu != c->dims.nv || c->dims.nf_max != 0) return RTOC_ERR_BAD_ARG;
  if (!c->h_model || !c->d_cost || !c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  if (c->h_model->m.type[0] == RTOC_JOINT_FREE_FLYER || c->h_model->m.ncontacts != 0) return RTOC_ERR_BAD_ARG;  // unconstr_dynamics.cpp:22-29
  int rc = ensure_buffer(c, RTOC_BUF_KKT);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_DX0);
  if (rc) return rc;
  if (!c->d_active) {  // no contacts: an all-zero schedule
    HIP_TRY(hipMalloc((void**)&c->d_active, sizeof(unsigned) * c->max_stages));
    HIP_TRY(hipMemsetAsync(c->d_active, 0, sizeof(unsigned) * c->max_stages, c->stream));
  }
  rbd::UkArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.cost = c->d_cost;
  a.x0 = c->d_x0;
  a.dx0 = c->buf[RTOC_BUF_DX0];
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nv = c->dims.nv;
  a.dt = dt;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_v = c->L.sol.off[RTOC_SOL_V], a.o_a = c->L.sol.off[RTOC_SOL_A], a.o_u = c->L.sol.off[RTOC_SOL_U];
  a.o_lmd = c->L.sol.off[RTOC_SOL_LMD], a.o_gmm = c->L.sol.off[RTOC_SOL_GMM];
  a.o_qxx = c->L.kkt.off[RTOC_KKT_QXX], a.o_qxu = c->L.kkt.off[RTOC_KKT_QXU], a.o_quu = c->L.kkt.off[RTOC_KKT_QUU];
  a.o_fx = c->L.kkt.off[RTOC_KKT_FX], a.o_lx = c->L.kkt.off[RTOC_KKT_LX], a.o_lu = c->L.kkt.off[RTOC_KKT_LU];
  a.o_qaa = c->L.cdd.off[RTOC_CDD_QAA], a.o_la = c->L.cdd.off[RTOC_CDD_LA], a.o_mj = c->L.cdd.off[RTOC_CDD_MJTJINV];
  c->ls_unconstr_dt = dt;
  if (c->ls_on && !c->d_costval) HIP_TRY(hipMalloc((void**)&c->d_costval, sizeof(double) * c->batch * c->max_stages));
  a.cost_out = c->ls_on ? c->d_costval : nullptr;   // the line search's evalOCP (unconstr_line_search.cpp:56-83)
  hipLaunchKernelGGL(rbd::unconstr_eval_kkt_kernel, dim3(c->batch * c->nstages), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  c->fxx_state = 0;
  rc = launch_linearize(c, 1, true, dt);
  if (!rc && ubox_on(c)) rc = launch_ubox(c, UBOX_LINEARIZE);  // constraints_->linearizeConstraints (unconstr_intermediate_stage.cpp:68-69)
  return rc;
}

// ---- KKT error ------------------------------------------------------------------------------
static int launch_kkt_error(rtoc_ctx* c) {
  if (!c->d_kkterr) HIP_TRY(hipMalloc((void**)&c->d_kkterr, sizeof(double) * c->batch * (size_t)(1 + c->max_stages)));
  KktErrArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = (c->nrows > 0 || c->cone_contacts > 0) ? c->buf[RTOC_BUF_CON] : nullptr;
  a.rows = c->d_rows;
  a.grid = c->d_grid;
  a.out = c->d_kkterr;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nrows = c->nrows;
  a.cone_contacts = c->cone_contacts;
  a.cone_dim = c->cone_dim > 0 ? c->cone_dim : 3;
  a.cone_rows = c->cone_rows;
  a.impact_cones = c->impact_cones;
  a.nc_max = c->dims.nc_max;
  a.nv = c->dims.nv;
  a.nu = c->dims.nu;
  a.np = c->dims.np;
  a.nx = c->L.nx;
  a.kl = c->L.kkt;
  a.cl = c->L.cdd;
  a.nl = c->L.con;
  a.partial = c->d_kkterr + c->batch;
  hipLaunchKernelGGL(kkt_error_kernel, dim3(c->nstages, c->batch), dim3(64), 0, c->stream, a);
  hipLaunchKernelGGL(kkt_error_reduce_kernel, dim3((c->batch + 63) / 64), dim3(64), 0, c->stream, a.partial, c->d_kkterr,
                     c->nstages, c->batch);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_kkt_error(rtoc_ctx* c, double* host_out, int count) {
  CHECK_READY(c);
  if (!host_out || count < 0 || count > c->batch) return RTOC_ERR_BAD_ARG;
  int rc = launch_kkt_error(c);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(host_out, c->d_kkterr, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

// ---- evalKKT / updateSolution of the contact path closed on the device (ConfigurationSpaceCost, no inequality rows) ----
int rtoc_newton_iteration(rtoc_ctx* c, double kkt_tol, double tau);
// ---- inequality rows of the contact path evaluated on the device --------------------------------------------
int rtoc_set_barrier_param(rtoc_ctx* c, double barrier_param, double fraction_to_boundary_rule) {
  if (!c) return RTOC_ERR_BAD_ARG;
  if (!(barrier_param > 0.0) || !(fraction_to_boundary_rule > 0.0) || !(fraction_to_boundary_rule < 1.0)) return RTOC_ERR_BAD_ARG;
  c->barrier = barrier_param;
  c->ftb_rule = fraction_to_boundary_rule;
  c->epoch++;
  return RTOC_OK;
}

int rtoc_set_friction_coefficients(rtoc_ctx* c, const double* mu, int ncontacts) {
  if (!c || !mu || ncontacts < 1 || ncontacts > RTOC_MAX_CONTACTS) return RTOC_ERR_BAD_ARG;
  for (int i = 0; i < ncontacts; ++i)
    if (!(mu[i] > 0.0)) return RTOC_ERR_BAD_ARG;   // ContactStatus::setFrictionCoefficient
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_mu) HIP_TRY(hipMalloc((void**)&c->d_mu, sizeof(double) * RTOC_MAX_CONTACTS));
  double full[RTOC_MAX_CONTACTS] = {0.0};
  memcpy(full, mu, sizeof(double) * ncontacts);
  HIP_TRY(hipMemcpyAsync(c->d_mu, full, sizeof(full), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_mu = ncontacts;
  return RTOC_OK;
}

static bool device_cones_on(const rtoc_ctx* c) { return c->cone_contacts > 0 && c->cone_rows == RTOC_FRICTION_ROWS && c->d_mu != nullptr; }
static bool device_wrench_on(const rtoc_ctx* c) { return c->cone_contacts > 0 && c->cone_rows == RTOC_WRENCH_ROWS && c->d_wcone != nullptr; }

int rtoc_set_wrench_cone_params(rtoc_ctx* c, const double* xy_mu, int ncontacts) {
  if (!c || !xy_mu || ncontacts < 1 || ncontacts > RTOC_MAX_CONTACTS) return RTOC_ERR_BAD_ARG;
  std::vector<double> table((size_t)RTOC_MAX_CONTACTS * RTOC_WRENCH_ROWS * 6, 0.0);
  for (int k = 0; k < ncontacts; ++k) {
    const int rc = rtoc_wrench_cone_matrix(xy_mu[3 * k], xy_mu[3 * k + 1], xy_mu[3 * k + 2], &table[(size_t)k * RTOC_WRENCH_ROWS * 6]);
    if (rc) return rc;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_wcone) HIP_TRY(hipMalloc((void**)&c->d_wcone, sizeof(double) * table.size()));
  HIP_TRY(hipMemcpyAsync(c->d_wcone, table.data(), sizeof(double) * table.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

static int launch_wrench_cones(rtoc_ctx* c, int mode) {
  const rtoc_robot_model& m = c->h_model->m;
  if (m.ncontacts > c->cone_contacts) return RTOC_ERR_BAD_ARG;
  for (int k = 0; k < m.ncontacts; ++k)
    if (m.contact_type[k] != RTOC_CONTACT_SURFACE) return RTOC_ERR_BAD_ARG;
  WcArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = c->buf[RTOC_BUF_CON];
  a.cone = c->buf[RTOC_BUF_CONE];
  a.table = c->d_wcone;
  a.grid = c->d_grid;
  a.active = c->d_active;
  a.nstages = c->nstages, a.batch = c->batch, a.ncontacts = m.ncontacts, a.mode = mode;
  a.row0 = c->dims.nc_max - RTOC_WRENCH_ROWS * c->cone_contacts, a.cone_stride = rtoc_wrench_cone_stride(c->cone_contacts);
  a.impact_cones = c->impact_cones;
  a.barrier = c->barrier;
  a.sol_stride = c->L.sol.stride, a.cdd_stride = c->L.cdd.stride, a.con_stride = c->L.con.stride;
  a.o_f = c->L.sol.off[RTOC_SOL_F], a.o_lf = c->L.cdd.off[RTOC_CDD_LF];
  a.nl = c->L.con;
  hipLaunchKernelGGL(wrench_cone_eval_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

static int launch_contact_cones(rtoc_ctx* c, int mode) {
  const rtoc_robot_model& m = c->h_model->m;
  if (m.ncontacts > c->cone_contacts) return RTOC_ERR_BAD_ARG;
  if (c->n_mu < m.ncontacts) return RTOC_ERR_NOT_READY;  // a friction coefficient for every contact of the model
  for (int k = 0; k < m.ncontacts; ++k)
    if ((m.contact_type[k] == RTOC_CONTACT_SURFACE ? 6 : 3) != c->cone_dim) return RTOC_ERR_BAD_ARG;
  CcArgs a;
  a.model = c->d_model;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = c->buf[RTOC_BUF_CON];
  a.cone = c->buf[RTOC_BUF_CONE];
  a.grid = c->d_grid;
  a.active = c->d_active;
  a.rotations = c->has_crot ? c->d_crot : nullptr;
  a.mu = c->d_mu;
  a.nstages = c->nstages, a.batch = c->batch, a.nv = m.nv, a.njoints = m.njoints, a.ncontacts = m.ncontacts;
  a.nlevels = c->h_model->nlevels, a.mode = mode;
  a.contact_dim = c->cone_dim, a.row0 = c->dims.nc_max - RTOC_FRICTION_ROWS * c->cone_contacts;
  a.cone_stride = rtoc_cone_stride(c->dims.nv, c->cone_contacts), a.dgdf_off = rtoc_cone_dgdf_off(c->dims.nv, c->cone_contacts);
  a.impact_cones = c->impact_cones;
  a.exact_jacobian = c->exact_cone_jacobian;
  a.barrier = c->barrier;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride, a.con_stride = c->L.con.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_f = c->L.sol.off[RTOC_SOL_F], a.o_lx = c->L.kkt.off[RTOC_KKT_LX], a.o_lf = c->L.cdd.off[RTOC_CDD_LF];
  a.nl = c->L.con;
  if (mode == CC_LINEARIZE && c->vals_fresh && c->d_vals) {   // kinematics already there: no tree walk (contact_cone_vals_kernel)
    CvArgs v;
    v.c = a;
    v.vals = c->d_vals;
    hipLaunchKernelGGL(contact_cone_vals_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), 0, c->stream, v);
    HIP_TRY(hipGetLastError());
    return RTOC_OK;
  }
  const size_t lds = cc_lds_bytes(a.nlevels, a.njoints, a.ncontacts);
  HIP_TRY(hipFuncSetAttribute((const void*)contact_cone_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(contact_cone_kernel, dim3(c->batch * (c->nstages - 1)), dim3(64), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// OCPSolver::initConstraints (src/solver/ocp_solver.cpp:92-96 -> DirectMultipleShooting::initConstraints): setSlackAndDual of
// the joint-limit rows (those with bounds on the device) and of the friction-cone rows (those with friction coefficients)
int rtoc_contact_init_constraints(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->buf[RTOC_BUF_SOL] || !(c->barrier > 0.0)) return RTOC_ERR_NOT_READY;
  const bool rows = c->nrows > 0 && c->d_bounds != nullptr, cones = device_cones_on(c), wrench = device_wrench_on(c);
  if (!rows && !cones && !wrench) return RTOC_ERR_NOT_READY;
  if ((cones || wrench) && (!c->h_model || !c->d_active)) return RTOC_ERR_NOT_READY;
  int rc = ensure_buffer(c, RTOC_BUF_CON);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(c->buf[RTOC_BUF_CON], 0, sizeof(double) * c->count[RTOC_BUF_CON], c->stream));
  if (rows) rc = launch_ubox(c, UBOX_INIT, true);
  if (!rc && cones) rc = launch_contact_cones(c, CC_INIT);
  if (!rc && wrench) rc = launch_wrench_cones(c, CC_INIT);
  return rc;
}

// linearizeSwitchingConstraint (src/dynamics/switching_constraint.cpp:26-70) on the grids that carry one
static int launch_switching_constraint(rtoc_ctx* c) {
  const rtoc_robot_model& m = c->h_model->m;
  SwLinArgs a;
  a.model = c->d_model;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.grid = c->d_grid;
  a.active = c->d_active;
  a.positions = c->has_cpos ? c->d_cpos : nullptr;
  a.rotations = c->has_crot ? c->d_crot : nullptr;
  a.nstages = c->nstages, a.batch = c->batch, a.nv = m.nv, a.nq = m.nq, a.njoints = m.njoints, a.ncontacts = m.ncontacts;
  a.nlevels = c->h_model->nlevels, a.floating = m.type[0] == RTOC_JOINT_FREE_FLYER, a.ns_max = c->dims.ns_max;
  a.exact_transport = c->exact_transport;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_v = c->L.sol.off[RTOC_SOL_V], a.o_a = c->L.sol.off[RTOC_SOL_A], a.o_xi = c->L.sol.off[RTOC_SOL_XI];
  a.o_phix = c->L.kkt.off[RTOC_KKT_PHIX], a.o_phit = c->L.kkt.off[RTOC_KKT_PHIT], a.o_pres = c->L.kkt.off[RTOC_KKT_PRES];
  a.o_lx = c->L.kkt.off[RTOC_KKT_LX], a.o_hx = c->L.kkt.off[RTOC_KKT_HX], a.o_scal = c->L.kkt.off[RTOC_KKT_SCAL];
  a.o_phia = c->L.cdd.off[RTOC_CDD_PHIA], a.o_la = c->L.cdd.off[RTOC_CDD_LA], a.o_ha = c->L.cdd.off[RTOC_CDD_HA];
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  for (int i = 0; i < c->nstages; ++i)
    if (c->h_grid[i].switching_constraint && c->h_grid[i].dims > c->dims.ns_max) return RTOC_ERR_BAD_ARG;
  a.nsel = 0;
  int nsw = 0;
  for (int i = 0; i + 1 < c->nstages; ++i)
    if (c->h_grid[i].switching_constraint) {
      if (nsw < 16) a.sel[nsw] = i;
      ++nsw;
    }
  if (nsw <= 16) a.nsel = nsw;
  const int per = a.nsel > 0 ? a.nsel : c->nstages - 1;
  const size_t lds = sw_lds_bytes(a.nlevels, a.njoints, a.ncontacts);
  HIP_TRY(hipFuncSetAttribute((const void*)switching_constraint_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(switching_constraint_lin_kernel, dim3(c->batch * per), dim3(64), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_contact_eval_kkt(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->h_model || !c->d_active || !c->d_cost || !c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
  bool switching = false;
  for (int i = 0; i < c->nstages; ++i) switching = switching || c->h_grid[i].switching_constraint;
  int rc = ensure_buffer(c, RTOC_BUF_KKT);
  if (!rc) rc = ensure_buffer(c, RTOC_BUF_CDD);
  if (rc) return rc;
  c->ls_unconstr_dt = 0.0;
  // PhaseBased discretisation: time_discretization_.correctTimeSteps(contact_sequence_, t) ahead of evalKKT (ocp_solver.cpp:115-117)
  if (c->sto_on) STO_LAUNCH(sto_time_steps_kernel, c);
  if (!c->d_costval) HIP_TRY(hipMalloc((void**)&c->d_costval, sizeof(double) * c->batch * c->max_stages));
  CostArgs a;
  a.sol = c->buf[RTOC_BUF_SOL];
  a.cost = c->d_cost;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.grid = c->d_grid;
  a.nstages = c->nstages, a.batch = c->batch, a.nv = c->dims.nv, a.nu = c->dims.nu;
  a.nf_max = c->dims.nf_max, a.ns_max = c->dims.ns_max, a.floating = c->dims.np == 6;
  a.sol_stride = c->L.sol.stride, a.kkt_stride = c->L.kkt.stride, a.cdd_stride = c->L.cdd.stride;
  a.o_q = c->L.sol.off[RTOC_SOL_Q], a.o_v = c->L.sol.off[RTOC_SOL_V], a.o_a = c->L.sol.off[RTOC_SOL_A], a.o_u = c->L.sol.off[RTOC_SOL_U];
  a.kl = c->L.kkt, a.cl = c->L.cdd;
  a.dt_inst = c->sto_on ? c->d_dt : nullptr;
  a.cost_out = c->d_costval;
  {
    // setZero of the KKT records (+ the constant diagonals of the cost) as one stream on the context's second stream (rtoc_riccati_sweep's), BESIDE
    // the values pre-pass of the rigid-body linearisation (lanes = bodies; writes its scratch and RTOC_CDD_IDC, which nothing
    // here zeroes): the one is bound by HBM writes, the other by latency -- 1.1 ms each per 4096 x 47 grid points, one after the
    // other on one stream.  The cost kernel and everything behind it wait for both.
    InitArgs ia;
    ia.kkt = c->buf[RTOC_BUF_KKT], ia.cost = c->d_cost, ia.grid = c->d_grid, ia.dt_inst = a.dt_inst;
    ia.nstages = c->nstages, ia.batch = c->batch, ia.nv = c->dims.nv, ia.nu = c->dims.nu, ia.floating = a.floating;
    ia.kkt_stride = c->L.kkt.stride, ia.o_qxx = c->L.kkt.off[RTOC_KKT_QXX], ia.o_quu = c->L.kkt.off[RTOC_KKT_QUU], ia.o_fxx = c->L.kkt.off[RTOC_KKT_FXX];
    const long long nrec = (long long)c->batch * c->nstages;
    // four workgroups per CU: half of the wave slots, so that the pre-pass's waves are resident beside them
    // (CUs from the device: four workgroups each)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    const int blocks = (int)(nrec < (long long)cus * 4 ? nrec : (long long)cus * 4);
    // init_records_kernel moves 16-byte pairs that must not straddle a field: record stride, the three fields it writes constants
    // into and the state dimension are even (rtoc_compute_layout pads fields to 64 B; checked here so that a layout change cannot
    // silently misplace the cost diagonals)
    if ((ia.kkt_stride | ia.o_qxx | ia.o_quu | ia.o_fxx) & 1) return RTOC_ERR_BAD_ARG;
    // the (re)allocation of the pre-pass's scratch synchronises the device: ahead of the fork, never under it
    if (!c->linearize_fused) {
      rc = ensure_rbd_values(c);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    hipLaunchKernelGGL(init_records_kernel, dim3(blocks), dim3(256), 0, c->stream2, ia);
    // from here on the second stream is forked: whatever fails below, c->stream is joined to it before this call returns
    hipError_t ej = hipEventRecord(c->ev_join, c->stream2);
    c->vals_fresh = 0;
    if (ej == hipSuccess && !c->linearize_fused) rc = launch_rbd_values(c, false);   // shared by the cone rows and the tangent walk below
    if (ej == hipSuccess) ej = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    else (void)hipStreamSynchronize(c->stream2);   // no event to wait on: drain the fork on the host
    if (ej != hipSuccess) {
      ctx_set_err(ej, __LINE__);
      return RTOC_ERR_HIP;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(contact_cost_kernel, dim3((c->batch * c->nstages + COST_GP - 1) / COST_GP), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  // constraints_->linearizeConstraints (intermediate_stage.cpp:109-110, impact_stage.cpp:95-96) of the rows evaluated here
  if (c->nrows > 0 && c->d_bounds && c->buf[RTOC_BUF_CON]) rc = launch_ubox(c, UBOX_LINEARIZE, true);
  if (!rc && device_cones_on(c) && c->buf[RTOC_BUF_CON]) rc = launch_contact_cones(c, CC_LINEARIZE);
  if (!rc && device_wrench_on(c) && c->buf[RTOC_BUF_CON]) rc = launch_wrench_cones(c, CC_LINEARIZE);
  if (!rc) rc = launch_state_equation(c, true);
  if (!rc) rc = launch_linearize(c, 1, false, 1.0);
  if (!rc && switching) rc = launch_switching_constraint(c);
  if (rc) c->vals_fresh = 0;  // a failed sequence leaves no kinematics a later stand-alone linearisation may reuse
  return rc;
}

// OCPSolver::solve's iteration schedule (ocp_solver.cpp:169-213), shared by the host shells (include/rtoc_robot.h)
int rtoc_solve_loop(const rtoc_solve_options* o, const rtoc_solve_callbacks* cb, rtoc_solve_stats* st) {
  if (!o || !cb || !st || !cb->update_solution || o->max_iter < 0) return RTOC_ERR_BAD_ARG;
  if (o->sto_enabled && (!cb->max_time_step || !cb->mesh_refinement)) return RTOC_ERR_BAD_ARG;
  st->convergence = 0, st->iter = 0, st->num_mesh_refinements = 0;
  int inner_iter = 0;
  for (int iter = 0; iter < o->max_iter; ++iter, ++inner_iter) {
    if (o->sto_enabled && cb->set_sto_regularization) {                                         // :171-177
      const int rc = cb->set_sto_regularization(cb->user, inner_iter < o->initial_sto_reg_iter ? o->initial_sto_reg : 0.0);
      if (rc) return rc;
    }
    double kkt_error = 0.0;
    int rc = cb->update_solution(cb->user, &kkt_error);                                         // :178-180
    if (rc) return rc;
    st->iter = iter + 1;
    if (o->sto_enabled && kkt_error < o->kkt_tol_mesh) {                                        // :181
      double max_dt = 0.0;
      rc = cb->max_time_step(cb->user, &max_dt);
      if (rc) return rc;
      if (max_dt > o->max_dt_mesh) {                                                            // :182-199
        rc = cb->mesh_refinement(cb->user);
        if (rc) return rc;
        inner_iter = 0;   // (the loop header makes it 1 for the next iteration, as in the reference)
        if (st->num_mesh_refinements < RTOC_SOLVE_MAX_REFINEMENTS) st->mesh_refinement_iter[st->num_mesh_refinements] = iter + 1;
        ++st->num_mesh_refinements;
      } else if (kkt_error < o->kkt_tol) {                                                      // :200-204
        st->convergence = 1;
        break;
      }
    } else if (kkt_error < o->kkt_tol) {                                                        // :206-210
      st->convergence = 1;
      break;
    }
  }
  if (!st->convergence) st->iter = o->max_iter;                                                 // :212-214
  return RTOC_OK;
}

int rtoc_contact_update_solution(rtoc_ctx* c, double tau, double* host_kkt_error, int count) {
  CHECK_READY(c);
  if (count < 0 || count > c->batch || (count > 0 && !host_kkt_error)) return RTOC_ERR_BAD_ARG;
  int rc = rtoc_contact_eval_kkt(c);
  if (!rc) rc = rtoc_newton_iteration(c, 0.0, tau);  // KKT error, condensation, sweep, expansion, steps, update, integrate
  if (rc) return rc;
  if (count > 0) {
    HIP_TRY(hipMemcpyAsync(host_kkt_error, c->d_kkterr, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return RTOC_OK;
}

// UnconstrOCPSolver::updateSolution (src/solver/unconstr_ocp_solver.cpp:96-118) of every instance, one launch
// sequence, no host synchronisation unless host_kkt_error is asked for
static int ensure_line_search(rtoc_ctx* c);
static int launch_eval_ocp(rtoc_ctx* c, double* out);

int rtoc_unconstr_update_solution(rtoc_ctx* c, double dt, double* host_kkt_error, int count) {
  CHECK_READY(c);
  if (count < 0 || count > c->batch || (count > 0 && !host_kkt_error)) return RTOC_ERR_BAD_ARG;
  const bool rows = ubox_on(c);
  int rc = rtoc_unconstr_eval_kkt(c, dt);            // dms_.evalKKT up to the condensation, + computeInitialStateDirection
  if (!rc) rc = launch_kkt_error(c);                  // performance_index.kkt_error (pre-condensation, like :74-75)
  if (!rc && c->ls_on) {                              // dms_.getEval() of the iterate: what UnconstrLineSearch::computeStepSize reads first
    rc = ensure_line_search(c);
    if (!rc) rc = launch_eval_ocp(c, c->d_eval);
  }
  if (!rc && rows) rc = launch_ubox(c, UBOX_CONDENSE);  // constraints_->condenseSlackAndDual (:76-77), ahead of the dynamics
  if (!rc) rc = rtoc_unconstr_condense(c);
  if (!rc) rc = rtoc_unconstr_backward(c, dt);
  if (!rc) rc = rtoc_unconstr_forward(c, dt);
  if (!rc) rc = rtoc_unconstr_expand(c, dt);
  if (rc) return rc;
  rc = ensure_buffer(c, RTOC_BUF_STEP);
  if (rc) return rc;
  hipLaunchKernelGGL(fill_steps_kernel, dim3((2 * c->batch + 255) / 256), dim3(256), 0, c->stream, c->buf[RTOC_BUF_STEP], 2 * c->batch);
  HIP_TRY(hipGetLastError());
  if (rows) rc = launch_ubox(c, UBOX_EXPAND);         // expandSlackAndDual + maxSlack/DualStepSize (:80-97)
  // line_search_.computeStepSize (unconstr_ocp_solver.cpp:107-111, unconstr_line_search.cpp:37-67): the filter's backtracking loop of
  // every instance over trial iterates evaluated on the device; the accepted primal steps replace the maximum ones
  if (!rc && c->ls_on) rc = rtoc_contact_line_search(c, nullptr);
  if (!rc && rows) rc = rtoc_update(c);               // updateSlack / updateDual (:106-118)
  if (rc) return rc;
  rc = rtoc_integrate_solution(c);
  if (rc) return rc;
  if (count > 0) {
    HIP_TRY(hipMemcpyAsync(host_kkt_error, c->d_kkterr, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return RTOC_OK;
}

// ---- SwitchingTimeOptimization::evalKKT: scatter + STO KKT-error term ----------------------
int rtoc_sto_eval_kkt(rtoc_ctx* c, const double* host_lt, const double* host_qtt, int nev, double* host_err_sq, int count) {
  CHECK_READY(c);
  if (nev < 0 || nev > 31 || count < 0 || count > c->batch || (nev > 0 && (!host_lt || !host_qtt))) return RTOC_ERR_BAD_ARG;
  const size_t n = (size_t)c->batch * (nev > 0 ? nev : 1);
  if (c->sto_cap < n) {
    if (c->d_sto) (void)hipFree(c->d_sto);
    c->d_sto = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_sto, (2 * n + c->batch) * sizeof(double)));
    c->sto_cap = n;
  }
  double* d_lt = c->d_sto;
  double* d_qtt = c->d_sto + n;
  double* d_err = c->d_sto + 2 * n;
  if (nev > 0) {
    HIP_TRY(hipMemcpyAsync(d_lt, host_lt, (size_t)c->batch * nev * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_qtt, host_qtt, (size_t)c->batch * nev * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  StoArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.grid = c->d_grid;
  a.lt = d_lt;
  a.qtt = d_qtt;
  a.err = d_err;
  a.nstages = c->nstages;
  a.batch = c->batch;
  a.nev = nev;
  a.stride = c->L.kkt.stride;
  a.scal_off = c->L.kkt.off[RTOC_KKT_SCAL];
  hipLaunchKernelGGL(sto_eval_kkt_kernel, dim3((c->batch + 63) / 64), dim3(64), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  if (host_err_sq && count > 0)
    HIP_TRY(hipMemcpyAsync(host_err_sq, d_err, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

// ---- switching-time optimisation resident on the device (sto.hpp) ------------------------------------------------
int rtoc_sto_set_problem(rtoc_ctx* c, double t0, double T, const double* event_times, int num_events, int per_instance,
                         const double* min_dwell_times, double barrier_param, double fraction_to_boundary_rule) {
  CHECK_READY(c);
  if (num_events == 0) {  // no discrete events on this horizon: nothing to optimise (switching_time_optimization.cpp:85-90)
    c->sto_on = 0;
    c->epoch++;
    return RTOC_OK;
  }
  if (num_events < 0 || num_events > RTOC_STO_MAX_EVENTS || !event_times || !min_dwell_times || !(T > 0.0)) return RTOC_ERR_BAD_ARG;
  if (!(barrier_param > 0.0) || !(fraction_to_boundary_rule > 0.0) || !(fraction_to_boundary_rule < 1.0)) return RTOC_ERR_BAD_ARG;  // sto_constraints.cpp:44-59
  if (!c->h_grid || num_events != sto_count_events(c)) return RTOC_ERR_BAD_ARG;
  for (int p = 0; p <= num_events; ++p)
    if (!(min_dwell_times[p] >= 0.0)) return RTOC_ERR_BAD_ARG;  // :38-43
  const size_t ne = (size_t)c->batch * num_events;
  std::vector<double> ts(ne);
  for (int b = 0; b < c->batch; ++b)
    for (int e = 0; e < num_events; ++e) {
      const double te = event_times[(per_instance ? (size_t)b * num_events : 0) + e];
      const double prev = e > 0 ? ts[(size_t)b * num_events + e - 1] : t0;
      if (!(te > prev) || !(te < t0 + T)) return RTOC_ERR_BAD_ARG;  // events ordered, inside the horizon
      ts[(size_t)b * num_events + e] = te;
    }
  if (c->sto_nev != num_events) {
    for (double** p : {&c->d_ts, &c->d_sto_cost, &c->d_sto_out})
      if (*p) (void)hipFree(*p), *p = nullptr;
  }
  if (!c->d_ts) HIP_TRY(hipMalloc((void**)&c->d_ts, sizeof(double) * ne));
  if (!c->d_sto_out) HIP_TRY(hipMalloc((void**)&c->d_sto_out, sizeof(double) * (2 * ne + c->batch)));
  if (!c->d_dt) HIP_TRY(hipMalloc((void**)&c->d_dt, sizeof(double) * c->batch * c->max_stages));
  if (!c->d_sto_con) HIP_TRY(hipMalloc((void**)&c->d_sto_con, sizeof(double) * c->batch * RTOC_STO_CON_STRIDE));
  if (!c->d_min_dwell) HIP_TRY(hipMalloc((void**)&c->d_min_dwell, sizeof(double) * (RTOC_STO_MAX_EVENTS + 1)));
  if (!c->d_kkterr) HIP_TRY(hipMalloc((void**)&c->d_kkterr, sizeof(double) * c->batch * (size_t)(1 + c->max_stages)));
  int rc = ensure_buffer(c, RTOC_BUF_STEP);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_ts, ts.data(), sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_min_dwell, min_dwell_times, sizeof(double) * (num_events + 1), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_sto_con, 0, sizeof(double) * c->batch * RTOC_STO_CON_STRIDE, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->sto_on = 1, c->sto_nev = num_events, c->sto_t0 = t0, c->sto_T = T;
  c->sto_barrier = barrier_param, c->sto_tau = fraction_to_boundary_rule;
  c->epoch++;
  STO_LAUNCH(sto_time_steps_kernel, c);  // the time steps that belong to these event times
  return RTOC_OK;
}

int rtoc_sto_set_regularization(rtoc_ctx* c, double sto_reg) {
  if (!c || !(sto_reg >= 0.0)) return RTOC_ERR_BAD_ARG;
  if (c->sto_reg != sto_reg) c->epoch++;
  c->sto_reg = sto_reg;
  return RTOC_OK;
}

int rtoc_sto_set_cost_terms(rtoc_ctx* c, const double* lt, const double* qtt_diag) {
  CHECK_READY(c);
  if (!c->sto_on || (!lt) != (!qtt_diag)) return RTOC_ERR_BAD_ARG;
  const size_t ne = (size_t)c->batch * c->sto_nev;
  if (!lt) {
    if (c->d_sto_cost) (void)hipFree(c->d_sto_cost);
    c->d_sto_cost = nullptr;
    c->epoch++;
    return RTOC_OK;
  }
  if (!c->d_sto_cost) {
    HIP_TRY(hipMalloc((void**)&c->d_sto_cost, sizeof(double) * 2 * ne));
    c->epoch++;
  }
  HIP_TRY(hipMemcpyAsync(c->d_sto_cost, lt, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_sto_cost + ne, qtt_diag, sizeof(double) * ne, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_sto_init_constraints(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_OK;  // sto_.initConstraints returns when STO is disabled (:47)
  STO_LAUNCH(sto_init_kernel, c);
  return RTOC_OK;
}

// the dwell-time rows' slack / dual handed over by the host ([batch][num_events + 1] each): a warm start, or a test's iterate
int rtoc_sto_set_slack_dual(rtoc_ctx* c, const double* slack, const double* dual) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_ERR_NOT_READY;
  if (!slack || !dual) return RTOC_ERR_BAD_ARG;
  const int np = c->sto_nev + 1, NP = RTOC_STO_MAX_EVENTS + 1;
  std::vector<double> h((size_t)c->batch * RTOC_STO_CON_STRIDE, 0.0);
  for (int b = 0; b < c->batch; ++b)
    for (int p = 0; p < np; ++p) {
      if (!(slack[(size_t)b * np + p] > 0.0) || !(dual[(size_t)b * np + p] > 0.0)) return RTOC_ERR_BAD_ARG;
      h[(size_t)b * RTOC_STO_CON_STRIDE + 0 * NP + p] = slack[(size_t)b * np + p];
      h[(size_t)b * RTOC_STO_CON_STRIDE + 1 * NP + p] = dual[(size_t)b * np + p];
    }
  HIP_TRY(hipMemcpyAsync(c->d_sto_con, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_sto_correct_time_steps(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_OK;
  STO_LAUNCH(sto_time_steps_kernel, c);
  return RTOC_OK;
}

static int sto_download(rtoc_ctx* c, const double* src, size_t per, double* host_out, int count) {
  if (!c->sto_on) return RTOC_ERR_NOT_READY;
  if (!host_out || count < 0 || count > c->batch) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipMemcpyAsync(host_out, src, sizeof(double) * per * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}
int rtoc_sto_get_event_times(rtoc_ctx* c, double* host_out, int count) {
  CHECK_READY(c);
  return sto_download(c, c->d_ts, c->sto_nev, host_out, count);
}
int rtoc_sto_get_time_steps(rtoc_ctx* c, double* host_out, int count) {
  CHECK_READY(c);
  return sto_download(c, c->d_dt, c->nstages, host_out, count);
}
int rtoc_sto_get_constraint_data(rtoc_ctx* c, double* host_out, int count) {
  CHECK_READY(c);
  return sto_download(c, c->d_sto_con, RTOC_STO_CON_STRIDE, host_out, count);
}
int rtoc_sto_get_kkt_terms(rtoc_ctx* c, double* host_lt, double* host_qtt, double* host_err_sq, int count) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_ERR_NOT_READY;
  const size_t ne = (size_t)c->batch * c->sto_nev;
  int rc = RTOC_OK;
  if (host_lt) rc = sto_download(c, c->d_sto_out, c->sto_nev, host_lt, count);
  if (!rc && host_qtt) rc = sto_download(c, c->d_sto_out + ne, c->sto_nev, host_qtt, count);
  if (!rc && host_err_sq) rc = sto_download(c, c->d_sto_out + 2 * ne, 1, host_err_sq, count);
  return rc;
}

// SwitchingTimeOptimization::evalKKT of every instance from the event times on the device (after rtoc_condense, like
// ocp_solver.cpp:118-119); rtoc_kkt_error's result (RTOC's d_kkterr) becomes OCPSolver::KKTError() incl. the STO term
int rtoc_sto_eval_kkt_device(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_OK;
  STO_LAUNCH(sto_eval_kkt_dev_kernel, c);
  return RTOC_OK;
}
int rtoc_sto_compute_step_sizes(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_OK;
  STO_LAUNCH(sto_step_sizes_kernel, c);
  return RTOC_OK;
}
int rtoc_sto_integrate_solution(rtoc_ctx* c) {
  CHECK_READY(c);
  if (!c->sto_on) return RTOC_OK;
  STO_LAUNCH(sto_integrate_kernel, c);
  return RTOC_OK;
}

// ---- filter line search (line_search_filter.cpp), batched ---------------------------------------
static int ensure_filter(rtoc_ctx* c) {
  if (c->d_filter) return RTOC_OK;
  HIP_TRY(hipMalloc((void**)&c->d_filter, sizeof(double) * 2 * RTOC_LINE_SEARCH_FILTER_CAPACITY * c->batch));
  HIP_TRY(hipMalloc((void**)&c->d_nfilter, sizeof(int) * c->batch));
  HIP_TRY(hipMalloc((void**)&c->d_ls_in, sizeof(double) * 2 * c->batch));
  HIP_TRY(hipMalloc((void**)&c->d_ls_flags, sizeof(int) * 2 * c->batch));
  HIP_TRY(hipMemsetAsync(c->d_nfilter, 0, sizeof(int) * c->batch, c->stream));
  return RTOC_OK;
}

int rtoc_line_search_clear(rtoc_ctx* c) {
  if (!c) return RTOC_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_filter(c);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(c->d_nfilter, 0, sizeof(int) * c->batch, c->stream));
  return RTOC_OK;
}

int rtoc_line_search_filter(rtoc_ctx* c, const double* cost, const double* violation, const int* mask, int count,
                            double cost_rate, double viol_rate, int* accepted) {
  if (!c || !cost || !violation || !accepted || count < 0 || count > c->batch) return RTOC_ERR_BAD_ARG;
  if (!(cost_rate > 0.0) || !(viol_rate > 0.0)) return RTOC_ERR_BAD_ARG;  // line_search_filter.cpp:14-19
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_filter(c);
  if (rc) return rc;
  if (count == 0) return RTOC_OK;
  HIP_TRY(hipMemcpyAsync(c->d_ls_in, cost, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_ls_in + c->batch, violation, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  if (mask) HIP_TRY(hipMemcpyAsync(c->d_ls_flags, mask, sizeof(int) * count, hipMemcpyHostToDevice, c->stream));
  FilterArgs a;
  a.filt = c->d_filter;
  a.nfilt = c->d_nfilter;
  a.cost = c->d_ls_in;
  a.viol = c->d_ls_in + c->batch;
  a.mask = mask ? c->d_ls_flags : nullptr;
  a.accepted = c->d_ls_flags + c->batch;
  a.count = count;
  a.cap = RTOC_LINE_SEARCH_FILTER_CAPACITY;
  a.cost_rate = cost_rate;
  a.viol_rate = viol_rate;
  a.seed_empty = 0;
  hipLaunchKernelGGL(line_search_filter_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(accepted, c->d_ls_flags + c->batch, sizeof(int) * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

// ---- DirectMultipleShooting::evalOCP's performance index and the filter line search on the device ----------------------
static int ensure_line_search(rtoc_ctx* c) {
  int rc = ensure_filter(c);
  if (rc) return rc;
  if (!c->d_eval) HIP_TRY(hipMalloc((void**)&c->d_eval, sizeof(double) * 4 * c->batch));
  if (!c->d_eval_part) HIP_TRY(hipMalloc((void**)&c->d_eval_part, sizeof(double) * 2 * c->batch * c->max_stages));
  if (!c->d_ls_steps) HIP_TRY(hipMalloc((void**)&c->d_ls_steps, sizeof(double) * 3 * c->batch));
  if (!c->d_ls_active) HIP_TRY(hipMalloc((void**)&c->d_ls_active, sizeof(int) * (c->batch + 1)));
  if (!c->d_ls_merit) HIP_TRY(hipMalloc((void**)&c->d_ls_merit, sizeof(double) * 2 * c->batch));
  return RTOC_OK;
}

// (cost + cost_barrier | primal_feasibility) of every instance from the records rtoc_contact_eval_kkt has just written
// (pre-condensation) into out[2][batch]
static int launch_eval_ocp(rtoc_ctx* c, double* out) {
  if (!c->d_costval || !c->buf[RTOC_BUF_KKT] || !c->buf[RTOC_BUF_CDD]) return RTOC_ERR_NOT_READY;
  EvalOcpArgs a;
  a.kkt = c->buf[RTOC_BUF_KKT];
  a.cdd = c->buf[RTOC_BUF_CDD];
  a.con = (c->nrows > 0 || c->cone_contacts > 0) ? c->buf[RTOC_BUF_CON] : nullptr;
  a.costval = c->d_costval;
  a.rows = c->d_rows;
  a.grid = c->d_grid;
  a.partial = c->d_eval_part;
  a.nstages = c->nstages, a.batch = c->batch, a.nrows = c->nrows;
  a.cone_contacts = c->cone_contacts, a.cone_dim = c->cone_dim > 0 ? c->cone_dim : 3, a.cone_rows = c->cone_rows;
  a.nc_max = c->dims.nc_max, a.impact_cones = c->impact_cones;
  a.nv = c->dims.nv, a.nx = c->L.nx;
  a.barrier = c->barrier;
  a.kl = c->L.kkt, a.cl = c->L.cdd, a.nl = c->L.con;
  hipLaunchKernelGGL(eval_ocp_kernel, dim3(c->nstages, c->batch), dim3(64), 0, c->stream, a);
  hipLaunchKernelGGL(eval_ocp_reduce_kernel, dim3((c->batch + 63) / 64), dim3(64), 0, c->stream, c->d_eval_part, out, c->nstages, c->batch);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

int rtoc_set_line_search(rtoc_ctx* c, int enable, double step_size_reduction_rate, double min_step_size, double filter_cost_reduction_rate,
                         double filter_constraint_violation_reduction_rate) {
  if (!c) return RTOC_ERR_BAD_ARG;
  if (enable && (!(step_size_reduction_rate > 0.0 && step_size_reduction_rate < 1.0) || !(min_step_size > 0.0) ||
                 !(filter_cost_reduction_rate > 0.0) || !(filter_constraint_violation_reduction_rate > 0.0)))
    return RTOC_ERR_BAD_ARG;
  c->ls_on = enable ? 1 : 0;
  c->ls_rate = step_size_reduction_rate, c->ls_min_step = min_step_size;
  c->ls_cost_rate = filter_cost_reduction_rate, c->ls_viol_rate = filter_constraint_violation_reduction_rate;
  c->epoch++;
  return RTOC_OK;
}

// trial = 0: DirectMultipleShooting::getEval() of the iterate rtoc_contact_eval_kkt has just linearised (records not yet condensed).
// trial = 1: dms_trial_.integratePrimalSolution(step) + evalOCP (line_search.cpp:65-71) at SOL (+) step DIR with the slacks moved
// by step x dslack, step = the primal entry of RTOC_BUF_STEP of every instance; RTOC_BUF_SOL / CON / DIR / STEP keep their
// contents, the KKT / CDD records are overwritten (the next rtoc_contact_eval_kkt rewrites them anyway).
static int eval_ocp_trial(rtoc_ctx* c, const double* steps, double* out) {
  const size_t nsol = c->count[RTOC_BUF_SOL], ncon = c->count[RTOC_BUF_CON];
  const bool has_con = c->buf[RTOC_BUF_CON] != nullptr;
  if (!c->d_sol_trial) HIP_TRY(hipMalloc((void**)&c->d_sol_trial, sizeof(double) * nsol));
  if (has_con && !c->d_con_trial) HIP_TRY(hipMalloc((void**)&c->d_con_trial, sizeof(double) * ncon));
  HIP_TRY(hipMemcpyAsync(c->d_sol_trial, c->buf[RTOC_BUF_SOL], sizeof(double) * nsol, hipMemcpyDeviceToDevice, c->stream));
  if (has_con) HIP_TRY(hipMemcpyAsync(c->d_con_trial, c->buf[RTOC_BUF_CON], sizeof(double) * ncon, hipMemcpyDeviceToDevice, c->stream));
  double* const sol = c->buf[RTOC_BUF_SOL];
  double* const con = c->buf[RTOC_BUF_CON];
  double* const stp = c->buf[RTOC_BUF_STEP];
  c->buf[RTOC_BUF_SOL] = c->d_sol_trial;
  if (has_con) c->buf[RTOC_BUF_CON] = c->d_con_trial;
  c->buf[RTOC_BUF_STEP] = const_cast<double*>(steps);
  int rc = rtoc_update(c);                       // slack += step dslack (dual step 0)
  if (!rc) rc = rtoc_integrate_solution(c);      // SplitSolution::integrate with the trial step
  if (!rc) rc = c->ls_unconstr_dt > 0.0 ? rtoc_unconstr_eval_kkt(c, c->ls_unconstr_dt) : rtoc_contact_eval_kkt(c);   // evalOCP's quantities (and, unused here, the derivatives)
  if (!rc) rc = launch_eval_ocp(c, out);
  c->buf[RTOC_BUF_SOL] = sol, c->buf[RTOC_BUF_CON] = con, c->buf[RTOC_BUF_STEP] = stp;
  c->vals_fresh = 0;
  c->fxx_state = 0;
  return rc;
}

int rtoc_contact_eval_ocp(rtoc_ctx* c, int trial, double* host_cost, double* host_violation, int count) {
  CHECK_READY(c);
  if (count < 0 || count > c->batch || (count > 0 && (!host_cost || !host_violation))) return RTOC_ERR_BAD_ARG;
  int rc = ensure_line_search(c);
  if (rc) return rc;
  double* out = c->d_eval + (trial ? 2 * c->batch : 0);
  rc = trial ? eval_ocp_trial(c, c->buf[RTOC_BUF_STEP], out) : launch_eval_ocp(c, out);
  if (rc) return rc;
  if (count > 0) {
    HIP_TRY(hipMemcpyAsync(host_cost, out, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(host_violation, out + c->batch, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return RTOC_OK;
}

static int launch_filter_device(rtoc_ctx* c, const double* eval, const int* mask, int seed_empty) {
  FilterArgs a;
  a.filt = c->d_filter, a.nfilt = c->d_nfilter;
  a.cost = eval, a.viol = eval + c->batch;
  a.mask = mask;
  a.accepted = c->d_ls_flags + c->batch;
  a.count = c->batch, a.cap = RTOC_LINE_SEARCH_FILTER_CAPACITY;
  a.cost_rate = c->ls_cost_rate, a.viol_rate = c->ls_viol_rate;
  a.seed_empty = seed_empty;
  hipLaunchKernelGGL(line_search_filter_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return RTOC_OK;
}

// LineSearch::computeStepSize, filter method (line_search.cpp:31-83), for every instance: on entry RTOC_BUF_STEP holds the
// maximum primal steps (fraction-to-boundary), d_eval[0] the current iterates' (cost + barrier, violation) -- rtoc_newton_iteration
// evaluates them right after the linearisation; on exit the primal entries of RTOC_BUF_STEP are the accepted steps.
int rtoc_set_line_search_method(rtoc_ctx* c, int method, double armijo_control_rate, double margin_rate, double eps) {
  if (!c) return RTOC_ERR_BAD_ARG;
  if (method != 0 && method != 1) return RTOC_ERR_BAD_ARG;
  if (method == 1 && (!(armijo_control_rate > 0.0) || !(margin_rate >= 0.0) || !(eps > 0.0))) return RTOC_ERR_BAD_ARG;
  c->ls_method = method;
  if (method == 1) c->ls_armijo = armijo_control_rate, c->ls_margin = margin_rate, c->ls_eps = eps;
  c->epoch++;
  return RTOC_OK;
}

int rtoc_line_search_trials(rtoc_ctx* c, int* trials) {
  if (!c || !trials) return RTOC_ERR_BAD_ARG;
  *trials = c->ls_trials;
  return RTOC_OK;
}

int rtoc_line_search_merit_terms(rtoc_ctx* c, double* host_penalty, double* host_directional_derivative, int count) {
  CHECK_READY(c);
  if (count < 0 || count > c->batch || !c->d_ls_merit) return RTOC_ERR_BAD_ARG;
  if (host_penalty) HIP_TRY(hipMemcpyAsync(host_penalty, c->d_ls_merit, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  if (host_directional_derivative)
    HIP_TRY(hipMemcpyAsync(host_directional_derivative, c->d_ls_merit + c->batch, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

int rtoc_contact_line_search(rtoc_ctx* c, int* host_trials) {
  CHECK_READY(c);
  if (!c->ls_on) return RTOC_ERR_NOT_READY;
  int rc = ensure_line_search(c);
  if (rc) return rc;
  // UnconstrLineSearch (src/line_search/unconstr_line_search.cpp) has the filter method only and ignores line_search_method: an
  // unconstrained context takes the filter path whatever rtoc_set_line_search_method said (its SOL records have no beta / mu / xi)
  const bool merit = c->ls_method == 1 && !(c->ls_unconstr_dt > 0.0);
  LsMeritArgs ma;
  if (!merit) {
    rc = launch_filter_device(c, c->d_eval, nullptr, 1);   // an empty filter is seeded with the current iterate (:58-62)
    if (rc) return rc;
  } else {
    // meritBacktrackingLineSearch (:87-109): penalty parameter from the multipliers of the iterate, directional derivative of the
    // merit function from ONE more trial at step eps for every instance
    if (!c->buf[RTOC_BUF_SOL]) return RTOC_ERR_NOT_READY;
    LsPenaltyArgs pa;
    pa.sol = c->buf[RTOC_BUF_SOL], pa.grid = c->d_grid, pa.penalty = c->d_ls_merit;
    pa.nstages = c->nstages, pa.batch = c->batch, pa.nv = c->dims.nv, pa.np = c->dims.np, pa.sl = c->L.sol, pa.margin = c->ls_margin;
    hipLaunchKernelGGL(ls_penalty_kernel, dim3(c->batch), dim3(64), 0, c->stream, pa);
    ma.cur = c->d_eval, ma.trial = c->d_eval + 2 * c->batch, ma.penalty = c->d_ls_merit, ma.dd = c->d_ls_merit + c->batch;
    ma.trial_steps = c->d_ls_steps, ma.alpha = c->d_ls_steps + 2 * c->batch, ma.active = c->d_ls_active, ma.accepted = c->d_ls_flags + c->batch;
    ma.batch = c->batch, ma.eps = c->ls_eps, ma.armijo = c->ls_armijo;
    ma.phase = 0;
    hipLaunchKernelGGL(ls_merit_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, ma);
    rc = eval_ocp_trial(c, c->d_ls_steps, c->d_eval + 2 * c->batch);
    if (rc) return rc;
    ma.phase = 1;
    hipLaunchKernelGGL(ls_merit_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, ma);
    HIP_TRY(hipGetLastError());
  }
  LsArgs a;
  a.steps = c->buf[RTOC_BUF_STEP];
  a.trial_steps = c->d_ls_steps;
  a.alpha = c->d_ls_steps + 2 * c->batch;
  a.active = c->d_ls_active;
  a.accepted = c->d_ls_flags + c->batch;
  a.nactive = c->d_ls_active + c->batch;
  a.batch = c->batch;
  a.rate = c->ls_rate, a.min_step = c->ls_min_step;
  const dim3 grid((c->batch + 255) / 256), block(256);
  HIP_TRY(hipMemsetAsync(a.nactive, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(ls_begin_kernel, grid, block, 0, c->stream, a);
  int nactive = 0, trials = merit ? 1 : 0;   // (the trial at step eps counts as an evaluation)
  HIP_TRY(hipMemcpyAsync(&nactive, a.nactive, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the backtracking of one instance ends once its step falls below min_step_size (line_search.cpp:64-80): at most
  // log(min_step) / log(rate) reductions from a full step; the bound only guards against a loop that never drains
  const int max_trials = (int)ceil(log(c->ls_min_step < 1.0 ? c->ls_min_step : 1.0) / log(c->ls_rate)) + 2;
  while (nactive > 0 && trials < max_trials + (merit ? 1 : 0)) {
    rc = eval_ocp_trial(c, c->d_ls_steps, c->d_eval + 2 * c->batch);
    if (rc) return rc;
    if (!merit) {
      rc = launch_filter_device(c, c->d_eval + 2 * c->batch, c->d_ls_active, 0);   // isAccepted + augment of the active instances
      if (rc) return rc;
    } else {
      ma.phase = 2;   // armijoCondition of the active instances
      hipLaunchKernelGGL(ls_merit_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, ma);
    }
    HIP_TRY(hipMemsetAsync(a.nactive, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(ls_advance_kernel, grid, block, 0, c->stream, a);
    HIP_TRY(hipMemcpyAsync(&nactive, a.nactive, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++trials;
  }
  c->ls_trials = trials;
  if (host_trials) *host_trials = trials;
  return RTOC_OK;
}

// ---- one Newton iteration of the whole batch as a single launch sequence ----------
// steps[b] <- 0 for instances whose KKT error is already below the tolerance: they keep their iterate
// kkterr holds OCPSolver::KKTError() itself (the sqrt, kkt_error.hpp); the reference tests KKTError() < kkt_tol
// (ocp_solver.cpp:200,206)
__global__ void mask_converged_kernel(double* steps, const double* kkterr, int* nconv, double tol, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  if (kkterr[b] < tol) {
    steps[2 * b] = 0.0;
    steps[2 * b + 1] = 0.0;
    atomicAdd(nconv, 1);
  }
}

static int newton_iteration_body(rtoc_ctx* c, double kkt_tol, double tau) {
  HIP_TRY(hipMemsetAsync(c->d_nconv, 0, sizeof(int), c->stream));
  int rc = launch_kkt_error(c);  // on the freshly linearised (pre-condensation) records
  if (!rc && c->ls_on) rc = launch_eval_ocp(c, c->d_eval);   // dms_.getEval(): cost + barrier, violation of the current iterate
  if (!rc) rc = rtoc_condense(c);
  if (!rc && c->sto_on) STO_LAUNCH(sto_eval_kkt_dev_kernel, c);   // sto_.evalKKT (ocp_solver.cpp:119); KKTError() gains the STO term
  if (!rc) rc = launch_sweep(c);
  if (!rc) rc = rtoc_expand(c, tau);  // directions + fraction-to-boundary step sizes, on the device
  if (rc) return rc;
  if (c->sto_on) STO_LAUNCH(sto_step_sizes_kernel, c);             // sto_.computeStepSizes, min with the stages' steps (:128-132)
  hipLaunchKernelGGL(mask_converged_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream,
                     c->buf[RTOC_BUF_STEP], c->d_kkterr, c->d_nconv, kkt_tol, c->batch);
  HIP_TRY(hipGetLastError());
  if (c->ls_on) {   // line_search_.computeStepSize (:133-139): the accepted primal steps replace the maximum ones
    rc = rtoc_contact_line_search(c, nullptr);
    if (rc) return rc;
  }
  rc = rtoc_update(c);
  if (!rc && c->buf[RTOC_BUF_SOL]) rc = rtoc_integrate_solution(c);
  if (!rc && c->sto_on) STO_LAUNCH(sto_integrate_kernel, c);       // sto_.integrateSolution (:143)
  return rc;
}

int rtoc_newton_iteration(rtoc_ctx* c, double kkt_tol, double tau) {
  CHECK_READY(c);
  if (!(kkt_tol >= 0.0) || !(tau > 0.0 && tau <= 1.0)) return RTOC_ERR_BAD_ARG;
  if (!c->d_nconv) HIP_TRY(hipMalloc((void**)&c->d_nconv, sizeof(int)));
  if (c->ls_on) {   // the backtracking loop synchronises with the host: no graph replay
    int rc = ensure_line_search(c);
    return rc ? rc : newton_iteration_body(c, kkt_tol, tau);
  }
  return run_graphed(c, &c->g_newton, kkt_tol, tau, [&]() { return newton_iteration_body(c, kkt_tol, tau); });
}

int rtoc_converged_count(rtoc_ctx* c, int* host_count) {
  CHECK_READY(c);
  if (!host_count) return RTOC_ERR_BAD_ARG;
  if (!c->d_nconv) return RTOC_ERR_NOT_READY;
  HIP_TRY(hipMemcpyAsync(host_count, c->d_nconv, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOC_OK;
}

// ---- stage dump / replay --------------------------------------------------------------------
static size_t dump_count(const rtoc_ctx* c, int b) {  // doubles of buffer b the kernels index
  const size_t per = (size_t)c->batch * c->nstages;
  switch (b) {
    case RTOC_BUF_KKT: return per * c->L.kkt.stride;
    case RTOC_BUF_RIC: return per * c->L.ric.stride;
    case RTOC_BUF_DIR: return per * c->L.dir.stride;
    case RTOC_BUF_CDD: return per * c->L.cdd.stride;
    case RTOC_BUF_CON: return per * c->L.con.stride;
    case RTOC_BUF_DX0: return (size_t)c->batch * c->L.nx;
    case RTOC_BUF_STEP: return (size_t)c->batch * 2;
    case RTOC_BUF_SE3: return per * RTOC_SE3_STRIDE;
    case RTOC_BUF_CONE:
      if (c->cone_contacts <= 0) return 0;
      return per * (c->cone_rows == RTOC_WRENCH_ROWS ? rtoc_wrench_cone_stride(c->cone_contacts)
                                                     : rtoc_cone_stride(c->dims.nv, c->cone_contacts));
    case RTOC_BUF_SOL: return per * c->L.sol.stride;
    default: return 0;
  }
}

int rtoc_save_stage_dump(rtoc_ctx* c, const char* path, unsigned int mask) {
  CHECK_READY(c);
  if (!path || !c->h_grid) return RTOC_ERR_BAD_ARG;
  rtoc_dump_header h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "RTOCDMP1", 8);
  h.version = 1;
  h.header_bytes = (unsigned)sizeof(h);
  h.dims = c->dims;
  h.nstages = c->nstages;
  h.batch = c->batch;
  h.nrows = c->nrows;
  h.cone_contacts = c->cone_contacts;
  h.cone_dim = c->cone_dim;
  h.cone_rows = c->cone_contacts > 0 ? c->cone_rows : 0;
  for (int b = 0; b < RTOC_NUM_BUFFERS; ++b)
    if ((mask >> b & 1u) && c->buf[b]) h.count[b] = dump_count(c, b);
  FILE* f = fopen(path, "wb");
  if (!f) return RTOC_ERR_IO;
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 &&
            fwrite(c->h_grid, sizeof(rtoc_grid), c->nstages, f) == (size_t)c->nstages &&
            (c->nrows == 0 || fwrite(c->h_rows, sizeof(rtoc_box_row), c->nrows, f) == (size_t)c->nrows);
  std::vector<double> stage;
  for (int b = 0; ok && b < RTOC_NUM_BUFFERS; ++b) {
    if (!h.count[b]) continue;
    stage.resize(h.count[b]);
    if (hipMemcpyAsync(stage.data(), c->buf[b], h.count[b] * sizeof(double), hipMemcpyDeviceToHost, c->stream) !=
            hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
      fclose(f);
      return RTOC_ERR_HIP;
    }
    ok = fwrite(stage.data(), sizeof(double), h.count[b], f) == h.count[b];
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? RTOC_OK : RTOC_ERR_IO;
}

int rtoc_load_stage_dump(const char* path, int device, rtoc_ctx** out) {
  if (!path || !out) return RTOC_ERR_BAD_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return RTOC_ERR_IO;
  rtoc_dump_header h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RTOCDMP1", 8) != 0 || h.version != 1 ||
      h.header_bytes != sizeof(h) || h.nstages < 2 || h.batch < 1 || h.nrows < 0 || h.nrows > h.dims.nc_max) {
    fclose(f);
    return RTOC_ERR_IO;
  }
  std::vector<rtoc_grid> grid(h.nstages);
  std::vector<rtoc_box_row> rows(h.nrows > 0 ? h.nrows : 1);
  if (fread(grid.data(), sizeof(rtoc_grid), h.nstages, f) != (size_t)h.nstages ||
      (h.nrows > 0 && fread(rows.data(), sizeof(rtoc_box_row), h.nrows, f) != (size_t)h.nrows)) {
    fclose(f);
    return RTOC_ERR_IO;
  }
  rtoc_ctx* c = nullptr;
  int rc = rtoc_create(&h.dims, h.nstages, h.batch, device, &c);
  if (!rc) rc = rtoc_set_grid(c, grid.data(), h.nstages);
  if (!rc && h.nrows > 0) rc = rtoc_set_constraint_rows(c, rows.data(), h.nrows);
  if (!rc && h.cone_contacts > 0)
    rc = h.cone_rows == RTOC_WRENCH_ROWS ? rtoc_set_wrench_cones(c, h.cone_contacts)
                                         : rtoc_set_friction_cones(c, h.cone_contacts, h.cone_dim);
  std::vector<double> stage;
  for (int b = 0; !rc && b < RTOC_NUM_BUFFERS; ++b) {
    if (!h.count[b]) continue;
    if (h.count[b] != dump_count(c, b)) {
      rc = RTOC_ERR_IO;
      break;
    }
    stage.resize(h.count[b]);
    if (fread(stage.data(), sizeof(double), h.count[b], f) != h.count[b]) {
      rc = RTOC_ERR_IO;
      break;
    }
    rc = rtoc_upload(c, b, 0, stage.data(), h.count[b]);
  }
  fclose(f);
  if (rc) {
    if (c) rtoc_destroy(c);
    return rc;
  }
  *out = c;
  return RTOC_OK;
}

// ---- multi-GPU: RCCL all-gather of the step directions over xGMI -----------------------------
// RCCL is resolved at run time (dlopen) so that the library has no link-time dependency on a
// particular librccl and shares the copy a host process (e.g. torch.distributed) already loaded.
int rtoc_gather_directions(rtoc_ctx* c, void* nccl_comm, double* out) {
  CHECK_READY(c);
  if (!nccl_comm || !out) return RTOC_ERR_BAD_ARG;
  typedef int (*allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
  static allgather_t fn = nullptr;
  if (!fn) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return RTOC_ERR_RCCL;
    fn = (allgather_t)dlsym(h, "ncclAllGather");
    if (!fn) return RTOC_ERR_RCCL;
  }
  const size_t count = (size_t)c->batch * c->nstages * c->L.dir.stride;
  const int nccl_float64 = 8;  // ncclFloat64 / ncclDouble
  const int rc = fn(c->buf[RTOC_BUF_DIR], out, count, nccl_float64, nccl_comm, c->stream);
  return rc == 0 ? RTOC_OK : RTOC_ERR_RCCL;
}

}  // extern "C"

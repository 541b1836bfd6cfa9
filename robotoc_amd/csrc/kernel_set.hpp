// kernel_set.hpp -- the kernels of ONE robot shape <NV, NU, NS> (NF = NS: the reference sizes the switching-
// constraint blocks with max_dimf too, split_kkt_matrix.cpp:7-34) gathered behind function pointers, so that the host
// runtime (rtoc_capi.hip) dispatches by dimensions at run time.  Every shape is compiled in its own translation unit
// (shape_inst.hip, once per entry of the SHAPES list in the Makefile): adding a robot = one entry + make.
#pragma once
#include <cstring>

#include "../../include/rtoc.h"
#include "condense.hpp"
#include "state_equation.hpp"
#include "unconstr_dynamics.hpp"
#include "friction_cone.hpp"
#include "kkt_error.hpp"
#include "integrate_solution.hpp"
#include "riccati_backward.hpp"
#include "riccati_backward_rs.hpp"
#include "riccati_backward_rv.hpp"
#include "riccati_backward_rw.hpp"
#include "riccati_backward_rw2.hpp"
#include "condense_rv.hpp"
#include "riccati_scan.hpp"
#include "riccati_forward.hpp"
#include "unconstr_riccati.hpp"

namespace rtoc {

typedef void (*bwd_fn)(BwdArgs);
typedef void (*fwd_fn)(FwdArgs);
typedef void (*fill_fn)(FillArgs);
typedef void (*ud_fn)(UdArgs);
typedef void (*cone_fn)(ConeArgs);
typedef void (*cond_fn)(CondArgs);
typedef void (*expd_fn)(ExpArgs);
typedef void (*scan_fn)(ScanArgs);
typedef void (*fscan_fn)(FwdScanArgs);
typedef void (*stoscan_fn)(StoScanArgs);
typedef void (*ur_fn)(UrArgs);
// what a shape plugin must agree on with the runtime that loads it: the kernel-set table and every argument block
constexpr size_t kernel_abi_stamp() {
  size_t h = 1469598103934665603ull;
  const size_t parts[] = {sizeof(BwdArgs), sizeof(FwdArgs), sizeof(FillArgs), sizeof(UdArgs), sizeof(ConeArgs), sizeof(CondArgs),
                          sizeof(ExpArgs), sizeof(ScanArgs), sizeof(FwdScanArgs), sizeof(UrArgs), sizeof(StoScanArgs)};
  for (size_t v : parts) h = (h ^ v) * 1099511628211ull;
  return h;
}

struct KernelSet {
  int nv, nu, ns;
  int nvariants;
  bwd_fn bwd[4];
  int bwd_waves[4];   // waves per workgroup
  int bwd_lds[4];
  int bwd_inst[4];    // OCP instances per workgroup
  bwd_fn bwd_sa;      // structured-Fxx form of variant 3 (role-split, 4 instances per workgroup), or nullptr
  bwd_fn bwd_rv;      // register-resident kernel, one wave per instance (riccati_backward_rv.hpp), or nullptr
  bwd_fn bwd_rv_sa;   // ... its structured-Fxx form, or nullptr
  bwd_fn bwd_rv_sto;  // ... the structured form for grids with switching-time optimisation, or nullptr
  int bwd_rv_lds;
  bwd_fn bwd_rw;      // register-wide kernel of the iCub-size shapes, one wave per instance and SIMD (riccati_backward_rw.hpp), or nullptr
  int bwd_rw_lds;
  int bwd_rw_threads;  // 64: riccati_backward_rw_kernel (T = 4), 128: riccati_backward_rw2_kernel (T = 5, two waves per instance)
  rtoc_record_layout kl, rl, dl, cl;  // record layouts the kernels were compiled for
  fwd_fn fwd;
  int fwd_threads;
  fill_fn fill;
  ud_fn ucond, uexp;  // UnconstrDynamics condense / expand
  ur_fn ubwd, ufwd;   // structured unconstrained Riccati recursion (unconstr_riccati.hpp); nullptr unless nu == nv, ns == 0
  cone_fn ccond, cexp;  // friction-cone rows
  cone_fn wcond, wexp;  // contact-wrench-cone rows
  cond_fn cond;
  cond_fn cond_split, mjt;  // split condensation: MJtJinv kernel + the rest
  int mjt_lds;
  int cond_threads, cond_lds, cond_split_lds;
  int cond_fuses_cones;  // the one-kernel condensation condenses the friction / wrench cone rows itself (CondCfg::FUSE)
  int cond_fused_default;  // ... and is the default pipeline of this shape: five of its work items fit the LDS of a CU
  cond_fn cond_rv_nc;      // ... without the cone-row code (contexts without cone rows)
  cond_fn cond_rv;         // register-chained condensation of the contact grid points, one wave per work item (condense_rv.hpp), or nullptr
  int cond_rv_lds;
  int cond_rv_cones;       // ... condenses friction-cone rows of point contacts itself (CrvCfg::CONES)
  expd_fn expd;
  int expd_threads, expd_lds;
  // horizon scan of the backward recursion (riccati_scan.hpp)
  scan_fn scan_elt, scan_comb;
  int scan_elt_lds, scan_comb_lds, scan_comb_threads;
  int scan_elt_stride, scan_ps_stride, scan_ps_soff;  // doubles per element / value record, offset of s
  int scan_policy_variant;                            // tile-split backward kernel used in its one-stage mode
  fscan_fn fscan_elt, fscan_comb, fscan_fin;          // forward recursion as a prefix scan
  int fscan_lds;
  stoscan_fn sto_prep, sto_vec;                       // scan on grids with switching-time optimisation (riccati_scan_sto.hpp)
  int sto_prep_lds, sto_vec_lds, sto_vec_threads, sto_scr_stride;
};

template <int NV, int NU, int NS, int NW0, int NW1>
inline KernelSet make_set() {
  KernelSet k;
  memset(&k, 0, sizeof(k));
  k.nv = NV;
  k.nu = NU;
  k.ns = NS;
  k.nvariants = 2;
  for (int v = 0; v < 4; ++v) k.bwd_inst[v] = 1;
  k.kl = StaticLayout<NV, NU, NS>::make().kkt;
  k.rl = StaticLayout<NV, NU, NS>::make().ric;
  k.bwd[0] = riccati_backward_kernel<NV, NU, NS, NW0>;
  k.bwd_waves[0] = NW0;
  k.bwd_lds[0] = BwdCfg<NV, NU, NS, NW0>::LDS_BYTES;
  k.bwd[1] = riccati_backward_kernel<NV, NU, NS, NW1>;
  k.bwd_waves[1] = NW1;
  k.bwd_lds[1] = BwdCfg<NV, NU, NS, NW1>::LDS_BYTES;
  // role-split kernel: matrix wave + vector wave per instance; needs the state in 4 tiles, the control Hessian and the
  // three free-rider columns of the G product in one 16-column tile
  if constexpr (2 * NV + 1 <= 64 && NU + 3 <= 16) {
    k.bwd[2] = riccati_backward_rs_kernel<NV, NU, NS>;
    k.bwd_waves[2] = 2;
    k.bwd_lds[2] = BwdCfg<NV, NU, NS, 2>::LDS_BYTES;
    k.nvariants = 3;
    // four instances per workgroup, both waves of an instance on one SIMD
    if (4 * BwdCfg<NV, NU, NS, 2>::LDS_BYTES + 64 <= 160 * 1024) {
      k.bwd[3] = riccati_backward_rs4_kernel<NV, NU, NS>;
      k.bwd_waves[3] = 8;
      k.bwd_lds[3] = 4 * BwdCfg<NV, NU, NS, 2>::LDS_BYTES + 64;
      k.bwd_inst[3] = 4;
      k.nvariants = 4;
      if constexpr (NV % 4 == 2 && NV > NU) k.bwd_sa = riccati_backward_rs4_kernel<NV, NU, NS, true>;
    }
  }
  if constexpr (RvCfg<NV, NU>::OK) {
    k.bwd_rv = riccati_backward_rv_kernel<NV, NU, NS, false>;
    if constexpr (NV % 16 == 2 && RvCfg<NV, NU>::T == 3 && NV - NU > 0 && NV - NU <= 8 && (NV - NU) % 2 == 0)
    {
      k.bwd_rv_sa = riccati_backward_rv_kernel<NV, NU, NS, true>;
      k.bwd_rv_sto = riccati_backward_rv_kernel<NV, NU, NS, true, true>;
    }
    k.bwd_rv_lds = rv_lds_bytes<NV, NU, NS>();
  }
  if constexpr (RwCfg<NV, NU>::OK) {
    k.bwd_rw = riccati_backward_rw_kernel<NV, NU, NS>;
    k.bwd_rw_lds = RwLds<NV, NU, NS>::BYTES;
    k.bwd_rw_threads = 64;
  } else if constexpr (RwCfg<NV, NU>::OK2) {
    k.bwd_rw = riccati_backward_rw2_kernel<NV, NU, NS>;
    k.bwd_rw_lds = Rw2Lds<NV, NU, NS>::BYTES;
    k.bwd_rw_threads = 128;
  }
  constexpr int NWF = (2 * NV + NU + 63) / 64;
  if constexpr (NWF == 1)
    k.fwd = riccati_forward_kernel<NV, NU, NS, 1>;
  else
    k.fwd = riccati_forward_mw_kernel<NV, NU, NS, NWF>;
  k.fwd_threads = 64 * NWF;
  k.dl = StaticLayout<NV, NU, NS>::make().dir;
  k.cl = StaticLayout<NV, NU, NS>::make().cdd;
  k.fill = unconstr_fill_kernel<NV>;
  k.ucond = unconstr_condense_kernel<NV>;
  k.uexp = unconstr_expand_kernel<NV>;
  if constexpr (NU == NV && NS == 0 && 2 * NV + 1 <= 64) {
    if constexpr (NV <= 8) k.ubwd = unconstr_riccati_backward_kernel<NV>;
    else k.ubwd = unconstr_riccati_backward_lds_kernel<NV>;
    k.ufwd = unconstr_riccati_forward_kernel<NV>;
  }
  k.ccond = cone_condense_kernel<NV, NS>;
  k.cexp = cone_expand_kernel<NV, NS>;
  k.wcond = wrench_condense_kernel<NV, NS>;
  k.wexp = wrench_expand_kernel<NV, NS>;
  constexpr int NF = NS;  // nf_max == ns_max for all supported robots
  k.cond = condense_kernel<NV, NU, NF, NS>;
  k.cond_split = condense_kernel<NV, NU, NF, NS, true>;
  k.mjt = mjtjinv_kernel<NV, NU, NF, NS>;
  k.mjt_lds = MjCfg<NV, NF>::LDS_BYTES;
  k.cond_threads = CondCfg<NV, NU, NF, NS>::NT;
  k.cond_lds = CondCfg<NV, NU, NF, NS>::LDS_BYTES;
  k.cond_split_lds = CondCfg<NV, NU, NF, NS, true>::LDS_BYTES;
  k.cond_fuses_cones = CondCfg<NV, NU, NF, NS>::FUSE ? 1 : 0;
  k.cond_fused_default = (CondCfg<NV, NU, NF, NS>::FUSE && CondCfg<NV, NU, NF, NS>::ITEMS >= 5) ? 1 : 0;
  k.cond_rv = nullptr;
  k.cond_rv_nc = nullptr;
  k.cond_rv_lds = 0;
  k.cond_rv_cones = 0;
  if constexpr (CrvCfg<NV, NU, NF, NS>::OK) {
    k.cond_rv = condense_rv_kernel<NV, NU, NF, NS>;
    k.cond_rv_nc = condense_rv_kernel<NV, NU, NF, NS, false>;
    k.cond_rv_lds = CrvCfg<NV, NU, NF, NS>::LDS_BYTES;
    k.cond_rv_cones = CrvCfg<NV, NU, NF, NS>::CONES ? 1 : 0;
  }
  // regression guard for the occupancy the quadruped shape is sized for (condense.hpp: five / ten work items per CU)
  static_assert(!(NV == 18 && NU == 12 && NS == 12) ||
                    (CondCfg<NV, NU, NF, NS, true>::ITEMS >= 5 && CondCfg<NV, NU, NF, NS, true>::MIN_WAVES == 4 && MjCfg<NV, NF>::ITEMS >= 10),
                "LDS carve of the split condensation grew past its granule budget");
  k.expd = expand_kernel<NV, NU, NF, NS>;
  k.expd_threads = 64;
  k.expd_lds = ExpCfg<NV, NU, NF>::LDS_BYTES;
  k.scan_elt = scan_element_kernel<NV, NU, NS>;
  k.scan_comb = scan_combine_kernel<NV>;
  k.scan_elt_lds = scan::ElementCfg<NV, NU, NS>::LDS_BYTES;
  k.scan_comb_threads = scan_comb_nt(NV);
  k.scan_comb_lds = scan::CombineCfg<NV, scan_comb_nt(NV)>::LDS_BYTES;
  k.scan_elt_stride = scan::EltLayout<NV>::STRIDE;
  k.scan_ps_stride = scan::EltLayout<NV>::PS_STRIDE;
  k.scan_ps_soff = scan::EltLayout<NV>::PS_S;
  k.scan_policy_variant = 1;  // NW1 waves share the tiles of the one stage
  k.fscan_elt = fwd_scan_element_kernel<NV, NU, NS>;
  k.fscan_comb = fwd_scan_combine_kernel<NV, NU, NS>;
  k.fscan_fin = fwd_scan_finish_kernel<NV, NU, NS>;
  k.fscan_lds = scan::FwdCfg<NV, NU>::LDS_BYTES;
  k.sto_prep = scan_sto_prep_kernel<NV, NU, NS>;
  k.sto_vec = scan_sto_vector_kernel<NV, NU, NS>;
  k.sto_prep_lds = scan::StoPrepCfg<NV, NU, NS>::LDS_DOUBLES * (int)sizeof(double);
  k.sto_vec_lds = scan::StoVecCfg<NV, NU, NS>::LDS_BYTES;
  k.sto_vec_threads = scan_sto_vec_nt(NV);
  k.sto_scr_stride = scan::StoScratch<NV, NU, NS>::STRIDE;
  static_assert(scan::StoVecCfg<NV, NU, NS>::LDS_BYTES <= 160 * 1024, "the vector pass keeps one grid point's bundle in LDS");
  static_assert(scan::CombineCfg<NV, scan_comb_nt(NV)>::LDS_BYTES <= 160 * 1024, "combination scratch must fit the LDS of a CU");
  return k;
}


}  // namespace rtoc

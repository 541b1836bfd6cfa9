// condense.hpp -- PLACEHOLDER (replaced by the real condensation / expansion kernels).
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"
namespace rtoc {
struct CondArgs { double* kkt; double* cdd; const rtoc_grid* grid; uint32_t* status; int nstages, batch; rtoc_record_layout kl, cl; };
struct ExpArgs { double* cdd; double* dir; const rtoc_grid* grid; int nstages, batch; rtoc_record_layout cl, dl; double tau; };
template <int NV, int NU, int NF, int NS> struct CondCfg { static constexpr int NT = 64; static constexpr int LDS_BYTES = 1024; };
template <int NV, int NU, int NF, int NS> __global__ void condense_kernel(CondArgs a) {}
template <int NV, int NU, int NF, int NS> __global__ void expand_kernel(ExpArgs a) {}
}
static inline int rtoc_con_stride(const rtoc_dims* d) { return 8 * ((d->nc_max * 7 + 7) / 8); }

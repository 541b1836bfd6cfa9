// shape_inst.hip -- one translation unit per robot shape: compiled once per entry of the Makefile's SHAPES list with
//   -DRTOC_SHAPE_NV=.. -DRTOC_SHAPE_NU=.. -DRTOC_SHAPE_NS=.. -DRTOC_SHAPE_NW0=.. -DRTOC_SHAPE_NW1=..
// (NW0 / NW1: wavefronts per instance of the two tile-split backward variants).  Instantiates every kernel of the
// shape and hands their entry points to the host runtime (rtoc_capi.hip: kernel_table).
#include <hip/hip_runtime.h>

#include "kernel_set.hpp"

#define RTOC_CAT5_(a, b, c, d, e) a##b##c##d##e
#define RTOC_CAT5(a, b, c, d, e) RTOC_CAT5_(a, b, c, d, e)
#define RTOC_SHAPE_FN RTOC_CAT5(rtoc_shape_, RTOC_SHAPE_NV, RTOC_CAT5(_, RTOC_SHAPE_NU, _, RTOC_SHAPE_NS, ), , )

namespace rtoc {
KernelSet RTOC_SHAPE_FN() {
  return make_set<RTOC_SHAPE_NV, RTOC_SHAPE_NU, RTOC_SHAPE_NS, RTOC_SHAPE_NW0, RTOC_SHAPE_NW1>();
}
}  // namespace rtoc

#ifdef RTOC_SHAPE_PLUGIN
// Built on its own (make plugin SHAPE=nv:nu:ns:nw0:nw1 -> ../librtoc_shape_<nv>_<nu>_<ns>.so): the host runtime loads the
// kernel set of a shape that is not in its compiled-in table through this one entry point (rtoc_capi.hip: load_plugin).
extern "C" int rtoc_shape_plugin(rtoc::KernelSet* out, size_t size_of_kernel_set, size_t abi_stamp) {
  // built from another revision of the kernel headers (table or argument blocks differ): refuse
  if (!out || size_of_kernel_set != sizeof(rtoc::KernelSet) || abi_stamp != rtoc::kernel_abi_stamp()) return -1;
  *out = rtoc::RTOC_SHAPE_FN();
  return 0;
}
#endif

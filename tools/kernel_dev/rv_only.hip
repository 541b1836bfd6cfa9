// compile-only harness: resource usage of the register-resident backward kernel    tools/kernel_dev/build.sh rv_only.hip
#include "../../robotoc_amd/csrc/riccati_backward_rv.hpp"
#ifdef RV_DENSE
template __global__ void rtoc::riccati_backward_rv_kernel<18, 12, 12, false>(rtoc::BwdArgs);
#else
template __global__ void rtoc::riccati_backward_rv_kernel<18, 12, 12, true>(rtoc::BwdArgs);
#endif

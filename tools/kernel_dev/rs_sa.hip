#include <hip/hip_runtime.h>
#include "../../robotoc_amd/csrc/riccati_backward_rs.hpp"
template __global__ void rtoc::riccati_backward_rs4_kernel<18, 12, 12, true>(rtoc::BwdArgs);

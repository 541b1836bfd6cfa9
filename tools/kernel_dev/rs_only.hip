// Development aid: instantiates only the ANYmal role-split backward kernel so that a change to
// riccati_backward_rs.hpp compiles in seconds and its register / scratch usage can be read off
// (hipcc ... -Rpass-analysis=kernel-resource-usage).  Not part of the library.
#include <hip/hip_runtime.h>
#include "../../robotoc_amd/csrc/riccati_backward_rs.hpp"
template __global__ void rtoc::riccati_backward_rs4_kernel<18, 12, 12>(rtoc::BwdArgs);

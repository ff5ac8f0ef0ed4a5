// kid_hot_plain.hip -- the plain hot build of the fused RK4 step as a translation unit of its own (see kid_hot_plain.inc),
// compiled with -mllvm -amdgpu-sched-strategy=max-ilp by the Makefile.
#include <hip/hip_runtime.h>
#include "../../include/kid.h"
#include "kid_berg_kernel.hpp"
using namespace kid;
#include "kid_hot_plain.inc"

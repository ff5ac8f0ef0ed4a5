// One instantiation of the per-berg kernel alone (the hot build of config 2 by default), for ISA / register work:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S \
//         -Rpass-analysis=kernel-resource-usage -o /tmp/hot.s tools/profiling/hot_only.hip
// KID_HOT_ARGS picks the template arguments.
#include <hip/hip_runtime.h>
#include "../../include/kid.h"
#include "../../icebergs_amd/csrc/kid_berg_kernel.hpp"
#ifndef KID_HOT_ARGS
#define KID_HOT_ARGS true, true, (PH_EVOLVE | PH_THERMO | PH_SPREAD), true, 1
#endif
void *kid_hot_only_ref() { return (void *)&berg_kernel<KID_HOT_ARGS>; }   // referencing the kernel from the host emits it

// pik_inst.hip -- the kernels and their launches for ONE chain length (-DPIK_INST_D=<dof>); the
// build compiles this file once per supported length, in parallel (pick_ik_amd/build.py).
// -DPIK_INST_STUB: no kernels for this length (experiment builds that only need some lengths).
#ifndef PIK_INST_D
#error "compile with -DPIK_INST_D=<dof>"
#endif
#define PIK_CAT2(a, b) a##b
#define PIK_CAT(a, b) PIK_CAT2(a, b)

#if defined(PIK_INST_STUB)
#include "pik_solver.hpp"
namespace pik {
const LaunchOps* PIK_CAT(launch_ops_d, PIK_INST_D)() { return nullptr; }
} // namespace pik
#else
#include "pik_launch.hpp"
namespace pik {
const LaunchOps* PIK_CAT(launch_ops_d, PIK_INST_D)() { return make_ops<PIK_INST_D>(); }
} // namespace pik
#endif

#if defined(PIK_PHASE_TIMING) && !defined(PIK_INST_STUB)
// experiments only: read and clear the phase counters of this chain length's kernels
extern "C" int pik_debug_phase_cycles(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pik::pik_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(pik::pik_phase_cycles), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

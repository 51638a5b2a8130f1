// lo_solve_fused_r32.hip -- instantiations of the fused end-to-end solve (lo_solve_fused_impl.h) for roots of 32 columns
// (one translation unit per root width: the kernel is large and the three compile in parallel).
#include "lo_solve_fused_impl.h"

namespace lo {
int fused_launch_r32(const FusedArgs& a, int nwg, hipStream_t st) { return fused_launch_rc<32>(a, nwg, st); }
}  // namespace lo

// lo_solve_fused_r8.hip -- instantiations of the fused end-to-end solve (lo_solve_fused_impl.h) for roots of 8 columns
// (one translation unit per root width: the kernel is large and the three compile in parallel).
#include "lo_solve_fused_impl.h"

namespace lo {
int fused_launch_r8(const FusedArgs& a, int nwg, hipStream_t st) { return fused_launch_rc<8>(a, nwg, st); }
}  // namespace lo

// lo_cg_close.h -- closing step of the single-column operator-resident solves (k_cg_onchip5, k_cg_rspace): the first
// workgroup of the group that finishes LAST does what k_cg_ctrl_onchip does (stop rule linear_cg.py:302-308, NaN check
// :199-200, "all converged before the first iteration" :207-208), mirrors the control block to pinned host memory and
// writes the ticket.  Every member left {final residual norm | tag + flags} as one 8-byte granule in a.close_gran.
#pragma once
#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"
#include "lo_group_reduce.h"

namespace lo {

// called by the workgroups with wig == 0 (all 256 threads)
__device__ __forceinline__ void cg_close_solve(const OnchipArgs& a, const int ngroups, const int t) {
  __shared__ int closer_s;
  __shared__ float red_s[R4_TPB];
  if (t == 0) closer_s = (atomicAdd(a.close_count, 1) == ngroups - 1) ? 1 : 0;
  __syncthreads();
  if (closer_s) {
    // (the other groups' granules were stored before their counter increments, but nothing orders the two for us:
    //  every granule is polled until its tag is there -- no fence anywhere)
    float lsum = 0.f, lnan = 0.f, lnotconv = 0.f, lredo = 0.f;
    unsigned spin = 0;
    bool lost = false;
    for (int64_t i = t; i < a.B && !lost; i += R4_TPB) {
      unsigned long long gr;
      for (;;) {
        gr = __hip_atomic_load(a.close_gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(gr >> 32) & 0x80000000u) break;
        if (++spin > R4_MAXSPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          lost = true;  // a group gave up (hand-off timeout): its members never arrive -- the host redoes the solve
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lost) break;
      const float rn = __uint_as_float((unsigned)(gr & 0xffffffffull));
      const unsigned fl = (unsigned)(gr >> 32);
      lsum += rn;
      if (rn != rn || (fl & 2u)) lnan = 1.f;
      if (!(fl & 1u)) lnotconv = 1.f;
      if (fl & 4u) lredo = 1.f;  // (diagonal form: this member wants the dense form)
    }
    if (lost) atomicExch(a.err, 1);
    const float mean = block_sum256(lsum, red_s) / (float)a.B;   // (the summation order of k_cg_ctrl_onchip)
    const float anynan = block_sum256(lnan, red_s);
    const float notconv = block_sum256(lnotconv, red_s);
    const float redo = block_sum256(lredo, red_s);
    if (t == 0) {
      CgCtrl* c = a.close_ctrl;
      c->rs_redo = redo > 0.f ? 1 : 0;
      c->iterations = a.iters;
      c->mean_resid = mean;
      if (anynan > 0.f) {
        c->nan_detected = 1;
        c->stop = 1;
      } else if (notconv == 0.f) {                 // every column converged before the first iteration (:207-208)
        c->skipped = 1;
        c->iterations = 0;
        c->stop = 1;
      } else if (a.close_floor_ok && mean < a.close_tol) {
        c->tol_reached = 1;
        c->stop = 1;
      }
      c->oc_err = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.close_mirror) {
        *a.close_mirror = *c;
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned*>(a.close_mirror) + 63, a.close_ticket, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace lo

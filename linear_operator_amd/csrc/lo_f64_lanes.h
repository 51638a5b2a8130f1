// lo_f64_lanes.h -- fp64 cross-lane helpers of the R-space kernels (lo_rspace.hip, lo_rspace3.hip): two 32-bit moves per
// value on the same hardware paths as lo_device.h (permlane swaps, DPP, ds_swizzle).
#pragma once
#include "lo_device.h"

namespace lo {

// ---------------------------------------------------------------------------------------------------------------------
// fp64 cross-lane helpers (two 32-bit moves per value on the same hardware paths as lo_device.h)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mk_d(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ unsigned lo_w(double x) { return (unsigned)__double2loint(x); }
__device__ __forceinline__ unsigned hi_w(double x) { return (unsigned)__double2hiint(x); }

template <int M>
__device__ __forceinline__ double xor_lane_d(double v) {
  return mk_d((unsigned)xor_lane_i<M>((int)lo_w(v)), (unsigned)xor_lane_i<M>((int)hi_w(v)));
}
template <int M>
__device__ __forceinline__ double bfly_add_d(double x) {
  if constexpr (M == 32) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo_w(x), lo_w(x), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi_w(x), hi_w(x), false, false);
    return mk_d(r0[0], r1[0]) + mk_d(r0[1], r1[1]);
  } else if constexpr (M == 16) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(lo_w(x), lo_w(x), false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(hi_w(x), hi_w(x), false, false);
    return mk_d(r0[0], r1[0]) + mk_d(r0[1], r1[1]);
  } else {
    return x + xor_lane_d<M>(x);
  }
}
__device__ __forceinline__ double lanes32_sum_d(double v) {
  v = bfly_add_d<1>(v); v = bfly_add_d<2>(v); v = bfly_add_d<4>(v); v = bfly_add_d<8>(v); v = bfly_add_d<16>(v);
  return v;
}
__device__ __forceinline__ double wave_sum_fast_d(double v) { return bfly_add_d<32>(lanes32_sum_d(v)); }
// sum over each row of 16 lanes, every lane ends with the total: rotations instead of the xor pattern (a plain sum needs no
// pairing, and row_ror:4 stays on the DPP path where the xor-4 exchange is an LDS-crossbar ds_swizzle with its own wait)
template <int CTRL>
__device__ __forceinline__ double dpp_add_d(double x) {
  return x + mk_d((unsigned)__builtin_amdgcn_update_dpp(0, (int)lo_w(x), CTRL, 0xf, 0xf, false),
                  (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi_w(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double row16_sum_d(double x) {
  x = dpp_add_d<0x128>(x);  // row_ror:8
  x = dpp_add_d<0x124>(x);  // row_ror:4
  x = dpp_add_d<0x4e>(x);   // quad_perm [2,3,0,1]
  x = dpp_add_d<0xb1>(x);   // quad_perm [1,0,3,2]
  return x;
}
// {value of the lower half-wave's lane, value of the upper half-wave's lane} for every lane pair (l, l + 32)
__device__ __forceinline__ void halves_d(double x, double& lower, double& upper) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo_w(x), lo_w(x), false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi_w(x), hi_w(x), false, false);
  lower = mk_d(r0[0], r1[0]);
  upper = mk_d(r0[1], r1[1]);
}
template <int M>
__device__ __forceinline__ double halve_pair_d(double lo, double hi, int lane) {
  if constexpr (M == 32) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo_w(lo), lo_w(hi), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi_w(lo), hi_w(hi), false, false);
    return mk_d(r0[0], r1[0]) + mk_d(r0[1], r1[1]);
  } else if constexpr (M == 16) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(lo_w(lo), lo_w(hi), false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(hi_w(lo), hi_w(hi), false, false);
    return mk_d(r0[0], r1[0]) + mk_d(r0[1], r1[1]);
  } else {
    const bool up = (lane & M) != 0;
    const double keep = up ? hi : lo;
    const double send = up ? lo : hi;
    return keep + xor_lane_d<M>(send);
  }
}
template <int CNT, int M, int NV>
__device__ __forceinline__ void halving_steps_d(double (&v)[NV], int lane) {
  if constexpr (M >= 1) {
    if constexpr (CNT > 1) {
      constexpr int half = CNT / 2;
#pragma unroll
      for (int j = 0; j < half; ++j) v[j] = halve_pair_d<M>(v[j], v[j + half], lane);
      halving_steps_d<half, M / 2, NV>(v, lane);
    } else {
      v[0] = halve_pair_d<M>(v[0], v[0], lane);
      halving_steps_d<1, M / 2, NV>(v, lane);
    }
  }
}
// wave reduce-scatter of n (<= 32, power of two) fp64 register values: lane l ends with the wave sum of component
// l >> (6 - log2 n)
template <int n>
__device__ __forceinline__ double wave_rs_d(double (&v)[n], int lane) {
  constexpr int h0 = n / 2;
  double w[h0];
#pragma unroll
  for (int j = 0; j < h0; ++j) w[j] = halve_pair_d<32>(v[j], v[j + h0], lane);
  halving_steps_d<h0, 16, h0>(w, lane);
  return w[0];
}

}  // namespace lo

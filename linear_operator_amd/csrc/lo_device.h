// lo_device.h -- device-side helpers (wave64 / 256-thread workgroups, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace lo {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- xor-butterfly partner fetch on the gfx950 cross-lane hardware (no LDS crossbar except M == 4) ----
//   M = 32 / 16 : v_permlane32_swap / v_permlane16_swap     M = 8 : DPP row_ror:8
//   M = 4       : ds_swizzle SWAP,4                         M = 2, 1 : DPP quad_perm
template <int M>
__device__ __forceinline__ int xor_lane_i(int v) {
  if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
  else if constexpr (M == 4) return __builtin_amdgcn_ds_swizzle(v, 0x101f);
  else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);
  else return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);
}
// {a, b} = {own value, value of lane ^ M} in an order that depends on the lane: combine them symmetrically.
template <int M>
__device__ __forceinline__ void bfly_i(int x, int& a, int& b) {
  if constexpr (M == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    a = (int)r[0];
    b = (int)r[1];
  } else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    a = (int)r[0];
    b = (int)r[1];
  } else {
    a = x;
    b = xor_lane_i<M>(x);
  }
}
template <int M>
__device__ __forceinline__ float bfly_add(float x) {
  int a, b;
  bfly_i<M>(__float_as_int(x), a, b);
  return __int_as_float(a) + __int_as_float(b);
}
// butterfly sums over groups of 64 / 16 / 8 consecutive lanes: every lane of the group ends with the same bits
__device__ __forceinline__ float wave_sum_fast(float v) {
  v = bfly_add<1>(v); v = bfly_add<2>(v); v = bfly_add<4>(v); v = bfly_add<8>(v);
  v = bfly_add<16>(v); v = bfly_add<32>(v);
  return v;
}
__device__ __forceinline__ float lanes32_sum(float v) {  // sum over each half-wave (lanes 0-31 / 32-63)
  v = bfly_add<1>(v); v = bfly_add<2>(v); v = bfly_add<4>(v); v = bfly_add<8>(v);
  v = bfly_add<16>(v);
  return v;
}
__device__ __forceinline__ float lanes16_sum(float v) {
  v = bfly_add<1>(v); v = bfly_add<2>(v); v = bfly_add<4>(v); v = bfly_add<8>(v);
  return v;
}
__device__ __forceinline__ float lanes8_sum(float v) {
  v = bfly_add<1>(v); v = bfly_add<2>(v); v = bfly_add<4>(v);
  return v;
}

// ---- wave64 reduce-scatter primitives -------------------------------------------------------------
// One halving step over lane bit M: every lane holds `lo` (a component it keeps if its bit M is clear) and `hi`
// (kept if the bit is set); returns own kept value + the partner lane's value of the same component.
// gfx950 cross-lane hardware, no LDS crossbar (ds_bpermute) except for M == 4:
//   M = 32 / 16 : v_permlane32_swap / v_permlane16_swap  (upper half-rows of `lo` swap with lower half-rows of `hi`;
//                 afterwards lo' + hi' is exactly keep + partner's send in every lane)
//   M = 8       : DPP row_ror:8     M = 2, 1 : DPP quad_perm     M = 4 : ds_swizzle SWAP,4
template <int M>
__device__ __forceinline__ float xor_lane(float v) {
  return __int_as_float(xor_lane_i<M>(__float_as_int(v)));
}

template <int M>
__device__ __forceinline__ float halve_pair(float lo, float hi, int lane) {
  if constexpr (M == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const bool up = (lane & M) != 0;
    const float keep = up ? hi : lo;
    const float send = up ? lo : hi;
    return keep + xor_lane<M>(send);
  }
}

// Recursive-halving tail, fully compile-time indexed (runtime-indexed register arrays would go to scratch):
// v holds CNT live values; the step over lane bit M keeps the half selected by that bit and adds the partner's.
template <int CNT, int M, int NV>
__device__ __forceinline__ void halving_steps(float (&v)[NV], int lane) {
  if constexpr (M >= 1) {
    if constexpr (CNT > 1) {
      constexpr int half = CNT / 2;
#pragma unroll
      for (int j = 0; j < half; ++j) v[j] = halve_pair<M>(v[j], v[j + half], lane);
      halving_steps<half, M / 2, NV>(v, lane);
    } else {
      v[0] = halve_pair<M>(v[0], v[0], lane) ;
      halving_steps<1, M / 2, NV>(v, lane);
    }
  }
}

// ---- coalesced row-block staging for the operator-resident kernels ------------------------------------------
// `nv` rows x W floats (W = 4, 8, 16 or 32; row-major, contiguous, 16-byte aligned) -> registers -> LDS rows of
// stride LD floats; a workgroup of NT threads moves NT rows (rows >= nv are zero-filled).  Consecutive lanes read
// consecutive 16-byte pieces (one wave instruction = 1 KiB contiguous) instead of each lane walking its own row.
template <int W, int NT>
__device__ __forceinline__ void rows_issue(const float* __restrict__ src, int nv, float4 (&reg)[W / 4]) {
  constexpr int QW = W / 4;
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int f = i * NT + (int)threadIdx.x;
    reg[i] = (f / QW < nv) ? reinterpret_cast<const float4*>(src)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int W, int LD, int NT>
__device__ __forceinline__ void rows_commit(float* __restrict__ lds, const float4 (&reg)[W / 4]) {
  constexpr int QW = W / 4;
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int f = i * NT + (int)threadIdx.x;
    *reinterpret_cast<float4*>(lds + (f / QW) * LD + 4 * (f % QW)) = reg[i];
  }
}

// Sum over all threads of a 256-thread block; result valid in every thread.  `red` >= 4 floats of LDS.
// Fixed order (wave butterfly, then waves 0..3) -> bitwise reproducible.
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Column-wise block reduction for the "thread t owns column t % c, row slot t / c" map:
// threads t < c*nrs participate; returns the sum over row slots in threads with slot == 0 (t < c).
// `red` = 256 floats of LDS.  Fixed tree order.
__device__ __forceinline__ float block_colsum(float v, int c, int nrs, float* red) {
  const int t = threadIdx.x;
  const int slot = t / c;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  int h = 1;
  while (h < nrs) h <<= 1;
  for (h >>= 1; h >= 1; h >>= 1) {
    if (slot < h && slot + h < nrs) red[t] += red[t + h * c];
    __syncthreads();
  }
  return red[t];
}

}  // namespace lo

// lo_dense_mfma.hip -- y = K v + d o v for dense K [B,N,N] and 2 <= c <= 32 columns (per pass) on the matrix cores
// (reference: AddedDiagLinearOperator._matmul added_diag_linear_operator.py:72-76 over DenseLinearOperator._matmul
// dense_linear_operator.py:60-64; BASELINE cfg5: dense 16384^2 with 16 probes + 1 rhs).
// HBM-bound: K (N^2 floats per member) is streamed exactly ONCE for all columns -- the 4-column VALU kernel in
// lo_dense.hip would stream it ceil(c/4) times.  8.5 flop/B at c = 17, far below the fp32-MFMA ridge.
//
// Workgroup = 64 rows x all N; loop over K in slabs of 128 columns (512 contiguous bytes per row: DRAM locality);
// waves = 2 row tiles x 2 k-halves, the k-halves are summed through LDS at the end:
//   global -> registers (next slab, issued before the MFMAs of the current one) -> LDS:
//     K slab [128 rows][64 k], row stride 68 floats: float4 stores stay aligned and the ds_read_b128 of the A
//     operand (lane = row) is bank-conflict free (4 * row mod 64 distinct within a 16-lane group);
//     v slab [64 k][32 cols] (columns >= c zero).
//   wave w owns rows 32 w .. 32 w + 31: D[row][col] += sum_k K[row][k] v[k][col] with v_mfma_f32_32x32x2_f32, lane
//   half h = lane >> 5 taking k in [32 h, 32 h + 32) of the slab.
// Epilogue fused: + d o v, store, CG inner product partial sum_rows v o y (one per 128-row tile).
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int DM_ROWS = 64, DM_KB = 128, DM_LD = DM_KB + 4;  // 512 contiguous bytes per row and slab (DRAM locality)

__device__ __forceinline__ int dm_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <bool DOT>
__global__ __launch_bounds__(kThreads) void k_dense_mv_mfma(const float* __restrict__ K, const float* __restrict__ dd,
                                                             int dd_mode, const float* __restrict__ v, int ldv, int c,
                                                             float* __restrict__ y, float* __restrict__ dot_part,
                                                             int ldd, int N, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float k_s[DM_ROWS * DM_LD];
  __shared__ float v_s[DM_KB * 32];
  __shared__ float dot_s[4][32];
  const int tile = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int row0 = tile * DM_ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wk = wave >> 1;  // row tile / k half of the slab this wave owns
  const float* Kb = K + (size_t)b * N * N;
  const size_t vbase = (size_t)b * N * ldv;

  // staging map: thread t loads float4 #(t + 256 u), u = 0..7 of the [128][64] slab: row = f / 16, quad = f % 16
  float4 kreg[8];
  float vreg[16];
  const bool full_rows = ((N & 3) == 0) && (row0 + DM_ROWS <= N);  // workgroup-uniform fast path condition
  auto load_slab = [&](int kb) {
    if (full_rows && kb + DM_KB <= N) {  // unguarded 16-byte loads (keeps global_load_dwordx4 in the ISA)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = threadIdx.x + kThreads * u;
        const int r = f >> 5, q = f & 31;
        kreg[u] = *reinterpret_cast<const float4*>(Kb + (size_t)(row0 + r) * N + kb + 4 * q);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = threadIdx.x + kThreads * u;
        const int r = f >> 5, q = f & 31;
        const int grow = row0 + r, gk = kb + 4 * q;
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (grow < N && gk + e < N) ? Kb[(size_t)grow * N + gk + e] : 0.f;
        kreg[u] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = threadIdx.x + kThreads * u;  // [128 k][32 cols]
      const int kk = e >> 5, col = e & 31;
      vreg[u] = (col < c && kb + kk < N) ? v[vbase + (size_t)(kb + kk) * ldv + col] : 0.f;
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = threadIdx.x + kThreads * u;
      const int r = f >> 5, q = f & 31;
      *reinterpret_cast<float4*>(&k_s[r * DM_LD + 4 * q]) = kreg[u];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      v_s[threadIdx.x + kThreads * u] = vreg[u];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  load_slab(0);
  for (int kb = 0; kb < N; kb += DM_KB) {
    __syncthreads();  // previous slab fully consumed
    store_slab();
    __syncthreads();
    if (kb + DM_KB < N) load_slab(kb + DM_KB);  // in flight during the MFMAs below
    const float* arow = &k_s[(32 * wr + li) * DM_LD + 64 * wk + 32 * h];
    const float* bcol = &v_s[(64 * wk + 32 * h) * 32 + li];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bcol[(4 * q + 0) * 32], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bcol[(4 * q + 1) * 32], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bcol[(4 * q + 2) * 32], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bcol[(4 * q + 3) * 32], acc, 0, 0, 0);
    }
  }

  // the two waves that share a row tile sum their k-halves through LDS (fixed order: half 0 + half 1)
  __syncthreads();
  float* red = k_s;  // reuse: [2 row tiles][16 regs][64 lanes]
  if (wk == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wr * 16 + e) * 64 + lane] = acc[e];
  }
  __syncthreads();
  if (wk == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += red[(wr * 16 + e) * 64 + lane];
  }
  const float ddc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  float dacc = 0.f;
  if (wk == 0 && li < c) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + 32 * wr + dm_row(e, lane);
      if (row < N) {
        const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + row] : ddc;
        const size_t o = vbase + (size_t)row * ldv + li;
        const float vin = v[o];
        const float yv = fmaf(dv, vin, acc[e]);
        y[o] = yv;
        if (DOT) dacc = fmaf(vin, yv, dacc);
      }
    }
  }
  if (DOT) {
    dacc += __shfl_xor(dacc, 32, 64);
    if (h == 0) dot_s[wave][li] = dacc;  // waves with wk == 1 contribute 0
    __syncthreads();
    if (threadIdx.x < c)
      dot_part[((size_t)b * S + tile) * ldd + threadIdx.x] =
          (dot_s[0][threadIdx.x] + dot_s[1][threadIdx.x]) + (dot_s[2][threadIdx.x] + dot_s[3][threadIdx.x]);
  }
}

// ---- 5 <= c <= 20: sixteen columns on v_mfma_f32_16x16x4_f32, up to four more on the vector ALU ---------------------
// The 32-column tile above spends half of its matrix-core time on zero columns at c = 17 (16 probes + the right-hand
// side) and sits at 50 % matrix-core duty next to the HBM stream.  Here the first 16 columns use the 16-wide
// instruction (A operand: lane (m = l & 15, kk = l >> 4) reads K[row m][4 kk .. 4 kk + 3] of a 16-k group with one
// 16-byte LDS read and feeds the four values to four MFMAs whose k index is 4 kk + j; B: v[16 g + 4 kk + j][m]) and the
// columns 16 .. c-1 ride on the SAME A registers: one FMA per K element and column (v of those columns contiguous in k
// in LDS, one broadcast 16-byte read per group).  Same workgroup tile (64 rows, waves = 2 row tiles x 2 k-halves),
// same staging, same epilogue semantics and dot-partial layout as the kernel above.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DM_VLD = 20;  // row stride of the v slab [128 k][16 cols]: the four kk groups of a B read hit 2 x 16 banks

// Split-K (kchunk < N, gridDim.z slices): a single operator of N = 4000 has 63 row tiles for 256 CUs -- each slice takes a
// range of K's columns and leaves its raw partial product in ypart [slice][B][N][c]; k_dense_mv_finish adds the slices in
// fixed order, the diagonal term and the dot partials.
// UNAL: rows that are not 16-byte aligned (N % 4 != 0 or a misaligned base pointer) -- its own instantiation so that the
// aligned kernel keeps its 112 - 176 VGPRs (three workgroups per CU): with both paths in one body the compiler took 200 - 256.
template <bool DOT, int NV, bool UNAL>
__global__ __launch_bounds__(kThreads) void k_dense_mv_mfma16(const float* __restrict__ K, const float* __restrict__ dd,
                                                               int dd_mode, const float* __restrict__ v, int ldv, int c,
                                                               float* __restrict__ y, float* __restrict__ dot_part,
                                                               int ldd, int N, int kchunk, float* __restrict__ ypart,
                                                               const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ __attribute__((aligned(16))) float k_s[DM_ROWS * DM_LD];
  __shared__ float v_s[DM_KB * DM_VLD];
  __shared__ __attribute__((aligned(16))) float vx_s[(NV > 0 ? NV : 1) * DM_KB];
  __shared__ float dot_s[4][32];
  const int tile = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int row0 = tile * DM_ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 15, kk = lane >> 4;
  const int wr = wave & 1, wk = wave >> 1;
  const int cm = min(c, 16);  // matrix-core columns
  const float* Kb = K + (size_t)b * N * N;
  const size_t vbase = (size_t)b * N * ldv;

  float4 kreg[8];
  float vreg[8];
  float vxreg[2];
  const bool full_rows = ((N & 3) == 0) && (row0 + DM_ROWS <= N);
  auto load_slab = [&](int kb) {
    if constexpr (!UNAL) {
      if (full_rows && kb + DM_KB <= N) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = threadIdx.x + kThreads * u;
          const int r = f >> 5, q = f & 31;
          kreg[u] = *reinterpret_cast<const float4*>(Kb + (size_t)(row0 + r) * N + kb + 4 * q);
        }
      } else {
        // (last slab of a row whose length is not a multiple of the slab: guarded scalar loads; sizes with a ragged last
        // TILE do not come here -- the launcher gives them the buffer-load instantiation, whose range check makes the
        // missing rows free, because these loads made the last tile of every member a straggler: 7 x 10000^2 4.7 TB/s
        // against 5.0 at N = 10001.  __builtin_amdgcn_raw_buffer_load_b128 would keep the 16-byte pieces, but this
        // compiler lowers it to ONE dword load splat over the four components.)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = threadIdx.x + kThreads * u;
          const int r = f >> 5, q = f & 31;
          const int grow = row0 + r, gk = kb + 4 * q;
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = (grow < N && gk + e < N) ? Kb[(size_t)grow * N + gk + e] : 0.f;
          kreg[u] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    } else {
      // one float per lane, consecutive lanes on consecutive floats of a row: 256 contiguous bytes per wave instruction
      // whatever the alignment of the row (four scalar loads per lane at a 16-byte stride streamed at 2.9 TB/s, this
      // order at 3.7 - 4.2; 16-byte pieces read from the aligned address below each row and stored to LDS shifted by the
      // row's misalignment -- four conflicting 4-byte LDS stores per piece -- were SLOWER: 3.0 - 3.2 TB/s)
      // (buffer loads: the tile's rows as one descriptor, a 32-bit lane offset -- rows beyond N fall outside its range and
      // read 0; with 64-bit addresses the 32 loads in flight took 64 address registers: 179 - 243 VGPRs)
      const int tile_rows = min(DM_ROWS, N - row0);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(Kb) + (size_t)row0 * N, 0, tile_rows * N * 4, 0x00020000);
      const int kq = threadIdx.x & 127;
      const int voff0 = (kb + kq < N) ? (((int)(threadIdx.x >> 7) * N + kb + kq) * 4) : 0x7f000000;
#pragma unroll
      for (int u = 0; u < 32; ++u) {  // float #(t + 256 u) of the [64 rows][128 k] slab: row (t >> 7) + 2 u, column t & 127
        const float val = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff0 + u * 8 * N, 0, 0));
        if ((u & 3) == 0) kreg[u >> 2].x = val;
        else if ((u & 3) == 1) kreg[u >> 2].y = val;
        else if ((u & 3) == 2) kreg[u >> 2].z = val;
        else kreg[u >> 2].w = val;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = threadIdx.x + kThreads * u;  // [128 k][16 cols]
      const int kr = e >> 4, col = e & 15;
      vreg[u] = (col < cm && kb + kr < N) ? v[vbase + (size_t)(kb + kr) * ldv + col] : 0.f;
    }
    if constexpr (NV > 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + kThreads * u;  // [NV][128 k]
        const int x = e >> 7, kr = e & 127;
        vxreg[u] = (x < NV && kb + kr < N) ? v[vbase + (size_t)(kb + kr) * ldv + 16 + x] : 0.f;
      }
    }
  };
  auto store_slab = [&]() {
    if constexpr (!UNAL) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = threadIdx.x + kThreads * u;
        const int r = f >> 5, q = f & 31;
        *reinterpret_cast<float4*>(&k_s[r * DM_LD + 4 * q]) = kreg[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int e = threadIdx.x + kThreads * u;
        const float4 kq4 = kreg[u >> 2];
        k_s[(e >> 7) * DM_LD + (e & 127)] = (u & 3) == 0 ? kq4.x : ((u & 3) == 1 ? kq4.y : ((u & 3) == 2 ? kq4.z : kq4.w));
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = threadIdx.x + kThreads * u;
      v_s[(e >> 4) * DM_VLD + (e & 15)] = vreg[u];
    }
    if constexpr (NV > 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + kThreads * u;
        if (e < NV * DM_KB) vx_s[e] = vxreg[u];
      }
    }
  };

  f32x4 acc[2];
  float yx[2][NV > 0 ? NV : 1];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < (NV > 0 ? NV : 1); ++x) yx[rb][x] = 0.f;
  }

  const int kbeg = blockIdx.z * kchunk, kend = min(N, kbeg + kchunk);  // (kchunk: a multiple of the slab width)
  const bool split = kchunk < N;
  load_slab(kbeg);
  for (int kb = kbeg; kb < kend; kb += DM_KB) {
    __syncthreads();
    store_slab();
    __syncthreads();
    if (kb + DM_KB < kend) load_slab(kb + DM_KB);
    const float* arow = &k_s[(32 * wr + m) * DM_LD + 64 * wk + 4 * kk];
    const float* bcol = &v_s[(64 * wk + 4 * kk) * DM_VLD + m];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 16 * g);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + 16 * DM_LD + 16 * g);
      float bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = bcol[(16 * g + j) * DM_VLD];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bv[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bv[j], acc[1], 0, 0, 0);
      }
      if constexpr (NV > 0) {
#pragma unroll
        for (int x = 0; x < NV; ++x) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(&vx_s[x * DM_KB + 64 * wk + 16 * g + 4 * kk]);
          yx[0][x] = fmaf(a0[3], w4[3], fmaf(a0[2], w4[2], fmaf(a0[1], w4[1], fmaf(a0[0], w4[0], yx[0][x]))));
          yx[1][x] = fmaf(a1[3], w4[3], fmaf(a1[2], w4[2], fmaf(a1[1], w4[1], fmaf(a1[0], w4[0], yx[1][x]))));
        }
      }
    }
  }
  // vector-ALU columns: sum the four k groups of a lane quadruple (every lane ends with the total of row m)
  if constexpr (NV > 0) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int x = 0; x < NV; ++x) yx[rb][x] = bfly_add<32>(bfly_add<16>(yx[rb][x]));
  }
  // the two waves that share a row tile sum their k-halves through LDS (fixed order: half 0 + half 1)
  __syncthreads();
  float* red = k_s;  // reuse: [2 row tiles][8 + 2 NV values][64 lanes]
  constexpr int NRED = 8 + 2 * (NV > 0 ? NV : 0);
  if (wk == 1) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(wr * NRED + 4 * rb + i) * 64 + lane] = acc[rb][i];
      if constexpr (NV > 0) {
#pragma unroll
        for (int x = 0; x < NV; ++x) red[(wr * NRED + 8 + NV * rb + x) * 64 + lane] = yx[rb][x];
      }
    }
  }
  __syncthreads();
  if (wk == 0) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[rb][i] += red[(wr * NRED + 4 * rb + i) * 64 + lane];
      if constexpr (NV > 0) {
#pragma unroll
        for (int x = 0; x < NV; ++x) yx[rb][x] += red[(wr * NRED + 8 + NV * rb + x) * 64 + lane];
      }
    }
  }
  if (split) {  // raw partial of this slice (the epilogue runs in k_dense_mv_finish)
    if (wk == 0) {
      float* yp = ypart + ((size_t)blockIdx.z * gridDim.y + b) * (size_t)N * c;
      if (m < cm) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = row0 + 32 * wr + 16 * rb + 4 * kk + i;
            if (row < N) yp[(size_t)row * c + m] = acc[rb][i];
          }
      }
      if constexpr (NV > 0) {
        if (kk == 0) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const int row = row0 + 32 * wr + 16 * rb + m;
            if (row < N) {
#pragma unroll
              for (int x = 0; x < NV; ++x) yp[(size_t)row * c + 16 + x] = yx[rb][x];
            }
          }
        }
      }
    }
    return;
  }
  const float ddc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  float dacc = 0.f;
  float dax[NV > 0 ? NV : 1];
#pragma unroll
  for (int x = 0; x < (NV > 0 ? NV : 1); ++x) dax[x] = 0.f;
  if (wk == 0) {
    if (m < cm) {  // matrix-core columns: D[row = 4 kk + i][col = m] per 16-row block
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 32 * wr + 16 * rb + 4 * kk + i;
          if (row < N) {
            const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + row] : ddc;
            const size_t o = vbase + (size_t)row * ldv + m;
            const float vin = v[o];
            const float yv = fmaf(dv, vin, acc[rb][i]);
            y[o] = yv;
            if (DOT) dacc = fmaf(vin, yv, dacc);
          }
        }
    }
    if constexpr (NV > 0) {
      if (kk == 0) {  // vector-ALU columns: lane m holds row m of the block
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int row = row0 + 32 * wr + 16 * rb + m;
          if (row < N) {
            const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + row] : ddc;
#pragma unroll
            for (int x = 0; x < NV; ++x) {
              const size_t o = vbase + (size_t)row * ldv + 16 + x;
              const float vin = v[o];
              const float yv = fmaf(dv, vin, yx[rb][x]);
              y[o] = yv;
              if (DOT) dax[x] = fmaf(vin, yv, dax[x]);
            }
          }
        }
      }
    }
  }
  if (DOT) {
    dacc = bfly_add<32>(bfly_add<16>(dacc));  // over the four row groups: every lane has the total of column m
    if (kk == 0) dot_s[wave][m] = dacc;        // (waves with wk == 1 contribute 0)
    if constexpr (NV > 0) {
#pragma unroll
      for (int x = 0; x < NV; ++x) {
        const float tot = wave_sum_fast(dax[x]);  // lanes with kk != 0 hold 0
        if (lane == 0) dot_s[wave][16 + x] = tot;
      }
    }
    __syncthreads();
    if (threadIdx.x < c)
      dot_part[((size_t)b * S + tile) * ldd + threadIdx.x] =
          (dot_s[0][threadIdx.x] + dot_s[1][threadIdx.x]) + (dot_s[2][threadIdx.x] + dot_s[3][threadIdx.x]);
  }
}

// y = sum_slices ypart + dd o v; dot partial per 64-row tile and column (the layout of the fused epilogue).
// grid (tiles, B).  The tile's [64 rows, c] elements are CONTIGUOUS in the [N, c] layout: a flat, coalesced walk (element
// e = t + 256 u), all slices of an element in flight together, the products parked in LDS and summed per column in
// fixed row order.  (A first version walked the columns four at a time with two barriers per pass: 10.8 us per call for
// one operator of 4000 rows and 17 columns -- a fifth of that CG iteration.)
__global__ __launch_bounds__(kThreads) void k_dense_mv_finish(const float* __restrict__ ypart, int nslice,
                                                               const float* __restrict__ dd, int dd_mode,
                                                               const float* __restrict__ v, int c,
                                                               float* __restrict__ y, float* __restrict__ dot_part,
                                                               int N, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float prod_s[DM_ROWS * 33];  // [row][col], stride 33 (c <= 32)
  const int tile = blockIdx.x, b = blockIdx.y, S = gridDim.x, B = gridDim.y;
  const int row0 = tile * DM_ROWS;
  const int nr = min(DM_ROWS, N - row0);
  const int total = nr * c;
  const size_t base = ((size_t)b * N + row0) * c;
  const size_t sstride = (size_t)B * N * c;
  const float dconst = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  int r = (int)threadIdx.x / c, col = (int)threadIdx.x - r * c;
  const int dr = kThreads / c, dc = kThreads - dr * c;
  for (int e = threadIdx.x; e < total; e += kThreads) {
    float acc = 0.f;
    for (int s0 = 0; s0 < nslice; s0 += 8) {
      float part[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] = (s0 + u < nslice) ? ypart[(size_t)(s0 + u) * sstride + base + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < nslice) acc += part[u];  // slices in fixed order
    }
    const float vin = v[base + e];
    const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + row0 + r] : (dd_mode == LO_DIAG_CONST ? dconst : 0.f);
    const float yv = fmaf(dv, vin, acc);
    y[base + e] = yv;
    prod_s[r * 33 + col] = vin * yv;
    r += dr;
    col += dc;
    if (col >= c) {
      col -= c;
      ++r;
    }
  }
  if (dot_part) {
    __syncthreads();
    if ((int)threadIdx.x < c) {
      float t = 0.f;
      for (int rr = 0; rr < nr; ++rr) t += prod_s[rr * 33 + threadIdx.x];
      dot_part[((size_t)b * S + tile) * c + threadIdx.x] = t;
    }
  }
}

// number of K slices: enough workgroups for the chip (3 per CU) when tiles x B alone are too few
int dense_mfma_slices(int64_t B, int64_t N, int64_t c) {
  if (!dense_mfma_ok(N, c) || c > 20) return 1;
  const int64_t wgs = B * dense_mfma_tiles(N);
  int ks = (int)std::min<int64_t>(16, (768 + wgs - 1) / wgs);
  ks = (int)std::min<int64_t>(ks, std::max<int64_t>(1, N / (4 * DM_KB)));  // at least four slabs per slice
  return std::max(1, ks);
}

template <bool DOT>
static void launch_mv16(int nv, dim3 grid, hipStream_t st, const float* K, const float* d, int dd_mode, const float* v,
                        int ldv, int c, float* y, float* dot_part, int N, int kchunk, float* ypart, const int* stop) {
  dim3 block(kThreads);
  // (N % 64 != 0: a ragged last tile -- or rows that are not 16-byte aligned -- take the buffer-load instantiation)
  const bool unal = (N & 63) != 0 || (reinterpret_cast<uintptr_t>(K) & 15) != 0;
  // Three workgroups fit a CU (45 KB of LDS each); every workgroup streams the same number of bytes and the chip is
  // HBM-bound, so a grid runs in rounds.  With between three and four workgroups per CU (768 < W <= 1024 on 256 CUs) the
  // second round of three-per-CU is at most one third full; two per CU finish the same grid in two full rounds.  The
  // launch then asks for 12 KB of dynamic LDS it does not use, which caps the occupancy at two (tools/mb_dense_occ.py,
  // interleaved best of four: 4 x 16384^2 778 -> 720 us at 11 columns, 5 x 12288^2 593 -> 550 us; for larger grids the
  // cap LOSES 2 - 9 % and is not applied).
  size_t pad_lds = 0;
  {
    const int64_t W = (int64_t)grid.x * grid.y * grid.z, cus = std::max(64, onchip_num_workgroups());
    if (W > 3 * cus && W <= 4 * cus && !getenv("LO_DENSE_OCC3")) pad_lds = 12 * 1024;
  }
#define LO_MV16(NV_)                                                                                                   \
  do {                                                                                                                 \
    if (unal)                                                                                                          \
      hipLaunchKernelGGL((k_dense_mv_mfma16<DOT, NV_, true>), grid, block, pad_lds, st, K, d, dd_mode, v, ldv, c, y,   \
                         dot_part, ldv, N, kchunk, ypart, stop);                                                       \
    else                                                                                                               \
      hipLaunchKernelGGL((k_dense_mv_mfma16<DOT, NV_, false>), grid, block, pad_lds, st, K, d, dd_mode, v, ldv, c, y,  \
                         dot_part, ldv, N, kchunk, ypart, stop);                                                       \
  } while (0)
  switch (nv) {
    case 0: LO_MV16(0); break;
    case 1: LO_MV16(1); break;
    case 2: LO_MV16(2); break;
    case 3: LO_MV16(3); break;
    default: LO_MV16(4); break;
  }
#undef LO_MV16
}

// c = 1 stays on the vector-ALU kernel (6.2 TB/s at 16384^2); from two columns on the 16-wide matrix-core tile streams K
// faster (5.8 - 6.0 TB/s against 4.2 / 4.2 / 2.2 TB/s of the VALU kernel at c = 2 / 3 / 4: tools/mb_dense_cols.py)
bool dense_mfma_ok(int64_t N, int64_t c) { return c >= 2 && N >= 256; }
int dense_mfma_tiles(int64_t N) { return (int)((N + DM_ROWS - 1) / DM_ROWS); }
int dense_matvec_mfma(const float* K, const float* d, int dd_mode, const float* v, float* y, float* dot_part, int64_t B,
                      int64_t N, int64_t c, float* ypart, const int* stop, hipStream_t st) {
  dim3 grid(dense_mfma_tiles(N), (unsigned)B), block(kThreads);
  if (c <= 20 && !getenv("LO_DENSE_MFMA32")) {  // 16 columns on the 16-wide instruction + up to 4 on the vector ALU
    const int nv = (int)std::max<int64_t>(0, c - 16);
    const int ks = ypart ? dense_mfma_slices(B, N, c) : 1;
    int kchunk = (int)N;
    if (ks > 1) {
      kchunk = (int)(((N + ks - 1) / ks + DM_KB - 1) / DM_KB * DM_KB);
      grid.z = (unsigned)((N + kchunk - 1) / kchunk);
    }
    LO_PROF_BEGIN("dense_mv_mfma", st);
    if (dot_part) launch_mv16<true>(nv, grid, st, K, d, dd_mode, v, (int)c, (int)c, y, dot_part, (int)N, kchunk, ypart, stop);
    else launch_mv16<false>(nv, grid, st, K, d, dd_mode, v, (int)c, (int)c, y, dot_part, (int)N, kchunk, ypart, stop);
    LO_PROF_END(st);
    if (grid.z > 1) {
      LO_PROF_BEGIN("dense_mv_finish", st);
      hipLaunchKernelGGL(k_dense_mv_finish, dim3(grid.x, (unsigned)B), block, 0, st, ypart, (int)grid.z, d,
                         d ? dd_mode : LO_DIAG_NONE, v, (int)c, y, dot_part, (int)N, stop);
      LO_PROF_END(st);
    }
    LO_LAUNCH_CHECK();
    return LO_OK;
  }
  for (int64_t c0 = 0; c0 < c; c0 += 32) {  // column tiles of 32 (K is re-streamed per tile: only for c > 32)
    const int cn = (int)std::min<int64_t>(32, c - c0);
    LO_PROF_BEGIN("dense_mv_mfma", st);
    if (dot_part)
      hipLaunchKernelGGL((k_dense_mv_mfma<true>), grid, block, 0, st, K, d, dd_mode, v + c0, (int)c, cn, y + c0,
                         dot_part + c0, (int)c, (int)N, stop);
    else
      hipLaunchKernelGGL((k_dense_mv_mfma<false>), grid, block, 0, st, K, d, dd_mode, v + c0, (int)c, cn, y + c0,
                         dot_part, (int)c, (int)N, stop);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

}  // namespace lo

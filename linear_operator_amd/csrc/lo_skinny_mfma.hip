// lo_skinny_mfma.hip -- matrix-core versions of the two skinny-operand kernels for MANY right-hand-side columns
// (8 < c <= 32 per launch; cfg3 / cfg5 of BASELINE.json carry 16 probe columns + 1 rhs).  Same contracts as the
// VALU kernels in lo_skinny.hip:
//   tn : tpart[b,s] = A[rows_s]^T v[rows_s]                (optionally with the fused CG vector updates)
//   nn : y = sgn * A (sum_s tpart) + dd o v ; dot partials sum_rows v o y
// The dense contractions run on v_mfma_f32_32x32x2_f32 (exact fp32, an fmaf chain per output -- the one place
// north_star asks for MFMA).  With 17 columns the VALU path has to re-stream A three times (8-column register
// tiles); here every column rides in the 32-wide N dimension of the MFMA, so A is streamed ONCE.
//
// tn:  D[i = r][j = col] += sum_k A^T[r][k] v[k][col], k = 2 rows per instruction.  Lane l supplies
//      A-operand A[row + (l>>5)][r0 + (l&31)] -- one wave load covers two consecutive rows = 256 contiguous bytes --
//      and B-operand v[row + (l>>5)][l&31]; each (row, col) element of v is owned by exactly ONE lane, which is
//      where the fused updates (p = z + beta p / r -= alpha Ap, x += alpha p, ||r||^2) happen.
// nn:  D[i = row][j = col] += sum_k A[row][k] t[k][col]; lane half h = l>>5 takes the contiguous k range
//      [h*R4/2, (h+1)*R4/2) so that a lane reads R4/2 consecutive floats (float4 loads) of its row.
// C/D layout of 32x32 MFMA (cdna_hip_programming.md section 3): lane l holds col = l & 31 and rows
// (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), reg = 0..15.
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

using f32x16 = __attribute__((ext_vector_type(16))) float;


using TnFuseM = TnFuse;

__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int NT, int VMODE>
__global__ __launch_bounds__(kThreads) void k_skinny_tn_mfma(const float* __restrict__ A, int R4,
                                                              float* __restrict__ v, int ldv, int c,
                                                              float* __restrict__ tpart, int N, int rows_per,
                                                              TnFuseM f, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float red[4][32][33];
  __shared__ float coef_s[32];
  __shared__ float rr_s[4][32];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, kk = lane >> 5;
  const int lda = R4;
  const float* Ab = A + (size_t)b * N * lda;
  const size_t vbase = (size_t)b * N * ldv;

  float coef = 0.f;
  if (VMODE == 1) coef = (li < c && !f.first) ? f.beta[(size_t)b * ldv + li] : 0.f;
  if (VMODE == 2) {
    if (threadIdx.x < c) {
      const int k = threadIdx.x;
      float pAp = 0.f;
      for (int ss = 0; ss < f.S_dot; ++ss) pAp += f.pAp_part[((size_t)b * f.S_dot + ss) * ldv + k];
      const float rz = f.rz[(size_t)b * ldv + k];
      float a = (pAp < f.eps) ? 0.f : rz / pAp;        // linear_cg.py:254-257
      if (f.has_conv[(size_t)b * ldv + k]) a = 0.f;     // :260
      coef_s[k] = a;
      if (s == 0) f.alpha_out[(size_t)b * ldv + k] = a;
    }
    __syncthreads();
    coef = (li < c) ? coef_s[li] : 0.f;
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float rr = 0.f;

  // waves interleave row pairs: wave w takes pairs w, w+4, ... of the slice; UNR pairs in flight per wave so
  // that the 256-byte loads of several steps overlap (memory-level parallelism), MFMAs follow the loads
  constexpr int UNR = 4;
  const int npairs = (r1 - r0 + 1) / 2;
  for (int pr0 = wave; pr0 < npairs; pr0 += 4 * UNR) {
    float bval[UNR];
    float aval[UNR][NT];
    float t_old[UNR], t_z[UNR], t_ap[UNR], t_p[UNR], t_x[UNR];
    int rows[UNR];
    bool rvs[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pr = pr0 + 4 * u;
      rows[u] = r0 + 2 * pr + kk;
      rvs[u] = (pr < npairs) && (rows[u] < r1);
      const bool act = rvs[u] && li < c;
      const size_t o = vbase + (size_t)rows[u] * ldv + li;
      t_old[u] = act ? v[o] : 0.f;
      if (VMODE == 1) t_z[u] = act ? f.z[o] : 0.f;
      if (VMODE == 2) {
        t_ap[u] = act ? f.Ap[o] : 0.f;
        t_p[u] = act ? f.p[o] : 0.f;
        t_x[u] = act ? f.x[o] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int r = 32 * t + li;
        aval[u][t] = (rvs[u] && r < R4) ? Ab[(size_t)rows[u] * lda + r] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool act = rvs[u] && li < c;
      const size_t o = vbase + (size_t)rows[u] * ldv + li;
      if (VMODE == 0) {
        bval[u] = t_old[u];
      } else if (VMODE == 1) {
        bval[u] = f.first ? t_z[u] : fmaf(t_old[u], coef, t_z[u]);  // p.mul_(beta).add_(z)   :46
        if (act) v[o] = bval[u];
      } else {
        bval[u] = fmaf(-coef, t_ap[u], t_old[u]);                   // r - alpha * Ap         :264
        if (act) {
          v[o] = bval[u];
          f.x[o] = fmaf(coef, t_p[u], t_x[u]);                      // x + alpha * p          :31
        }
        rr = fmaf(bval[u], bval[u], rr);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval[u][t], bval[u], acc[t], 0, 0, 0);
  }

  // cross-wave reduction per output tile, fixed order
  float* out = tpart + ((size_t)b * S + s) * (size_t)R4 * c;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][d_row(e, lane)][li] = acc[t][e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32; idx += kThreads) {
      const int i = idx >> 5, j = idx & 31;
      const int r = 32 * t + i;
      if (r < R4 && j < c) out[(size_t)r * c + j] = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
    }
  }
  if (VMODE == 2) {
    rr += __shfl_xor(rr, 32, 64);
    if (kk == 0) rr_s[wave][li] = rr;
    __syncthreads();
    if (threadIdx.x < c)
      f.rr_part[((size_t)b * S + s) * ldv + threadIdx.x] =
          (rr_s[0][threadIdx.x] + rr_s[1][threadIdx.x]) + (rr_s[2][threadIdx.x] + rr_s[3][threadIdx.x]);
  }
}

template <bool DOT, int KQ>  // KQ = float4 loads per lane and tile = R4 / 8
__global__ __launch_bounds__(kThreads) void k_skinny_nn_mfma(const float* __restrict__ A, int R4,
                                                              const float* __restrict__ tpart,
                                                              const float* __restrict__ dd, int dd_mode, float sgn,
                                                              const float* __restrict__ v, int ldv, int c,
                                                              float* __restrict__ y, float* __restrict__ dot_part,
                                                              int ldd, int N, int rows_per,
                                                              const int* __restrict__ stop) {
  if (stop && *stop) return;
  extern __shared__ float t_s[];  // [R4][32]  (columns >= c zero)
  __shared__ float dot_s[4][32];
  __shared__ __attribute__((aligned(16))) float io_s[4][32 * 32 + 32];  // per wave: the v / y tile and its 32 diagonal values
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  for (int idx = threadIdx.x; idx < R4 * 32; idx += kThreads) {
    const int k = idx >> 5, j = idx & 31;
    float acc = 0.f;
    if (j < c) {
      const float* tp = tpart + (size_t)b * S * R4 * c + (size_t)k * c + j;
      for (int ss = 0; ss < S; ++ss) acc += tp[(size_t)ss * R4 * c];
    }
    t_s[idx] = acc;
  }
  __syncthreads();

  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, h = lane >> 5;
  const int lda = R4;
  const int KH = R4 / 2;  // k range per lane half
  const float* Ab = A + (size_t)b * N * lda;
  const size_t vbase = (size_t)b * N * ldv;
  const float* ddb = (dd_mode == LO_DIAG_FULL) ? dd + (size_t)b * N : dd;
  const float ddc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  float dacc = 0.f;

  for (int base = r0 + 32 * wave; base < r1; base += 128) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int arow = base + li;
    const bool av = arow < r1;
    const float* ap = Ab + (size_t)(av ? arow : r0) * lda + h * KH;
    // all loads of the tile first (A row half, and the epilogue's v / diagonal values), then the MFMA chain
    float4 a4[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) a4[q] = av ? *reinterpret_cast<const float4*>(ap + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float vin[16], dvv[16];
    // a full tile of a packed block (ldv == c, c a multiple of 4): its 32 rows x c columns are 128 c contiguous bytes --
    // in (and, below, out) through the wave's LDS tile with 16-byte accesses; the direct path touches two 4 c-byte pieces
    // per instruction with the lanes li >= c idle (16 such loads + 16 stores per tile)
    const bool staged = ((c & 3) == 0) && ldv == c && base + 32 <= r1;
    float* tile = io_s[wave];
    if (staged) {
      const float4* src = reinterpret_cast<const float4*>(v + vbase + (size_t)base * c);
      const int n4 = 8 * c;  // float4 of the tile
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = lane + 64 * q;
        if (f < n4) *reinterpret_cast<float4*>(tile + 4 * f) = src[f];
      }
      if (dd_mode == LO_DIAG_FULL && lane < 32) tile[32 * 32 + lane] = ddb[base + lane];
      __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order; this pins the compiler's order)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int lr = d_row(e, lane);
        vin[e] = (li < c) ? tile[lr * c + li] : 0.f;
        dvv[e] = (dd_mode == LO_DIAG_FULL) ? ((li < c) ? tile[32 * 32 + lr] : 0.f) : ddc;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = base + d_row(e, lane);
        const bool ok = (li < c) && (row < r1);
        vin[e] = ok ? v[vbase + (size_t)row * ldv + li] : 0.f;
        dvv[e] = (dd_mode == LO_DIAG_FULL) ? (ok ? ddb[row] : 0.f) : ddc;
      }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float* tb = t_s + (size_t)(h * KH + 4 * q) * 32 + li;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].x, tb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].y, tb[32], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].z, tb[64], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].w, tb[96], acc, 0, 0, 0);
    }
    if (staged) {
      __builtin_amdgcn_wave_barrier();  // every lane holds its vin / dvv: the tile now takes the output
      if (li < c) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float yv = fmaf(dvv[e], vin[e], sgn * acc[e]);
          tile[d_row(e, lane) * c + li] = yv;
          if (DOT) dacc = fmaf(vin[e], yv, dacc);
        }
      }
      __builtin_amdgcn_wave_barrier();
      float4* dst = reinterpret_cast<float4*>(y + vbase + (size_t)base * c);
      const int n4 = 8 * c;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = lane + 64 * q;
        if (f < n4) dst[f] = *reinterpret_cast<const float4*>(tile + 4 * f);
      }
      __builtin_amdgcn_wave_barrier();  // the next tile reuses it
    } else if (li < c) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = base + d_row(e, lane);
        if (row < r1) {
          const float yv = fmaf(dvv[e], vin[e], sgn * acc[e]);
          y[vbase + (size_t)row * ldv + li] = yv;
          if (DOT) dacc = fmaf(vin[e], yv, dacc);
        }
      }
    }
  }
  if (DOT) {
    dacc += __shfl_xor(dacc, 32, 64);
    if (h == 0) dot_s[wave][li] = dacc;
    __syncthreads();
    if (threadIdx.x < c)
      dot_part[((size_t)b * S + s) * ldd + threadIdx.x] =
          (dot_s[0][threadIdx.x] + dot_s[1][threadIdx.x]) + (dot_s[2][threadIdx.x] + dot_s[3][threadIdx.x]);
  }
}

bool skinny_mfma_ok(int R4, int64_t c) {
  return c > 8 && c <= 32 && (R4 == 8 || R4 == 16 || R4 == 32 || R4 == 64 || R4 == 128);
}

template <int VMODE>
static int tn_mfma_launch(const float* A, int R4, float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
                          const TnFuseM& f, const int* stop, hipStream_t st) {
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  const int NT = (R4 + 31) / 32;
  LO_PROF_BEGIN(R4 == 32 ? "skinny_tn_mfma_R32" : (R4 == 16 ? "skinny_tn_mfma_R16" : "skinny_tn_mfma"), st);
#define LO_T(NTV)                                                                                              \
  hipLaunchKernelGGL((k_skinny_tn_mfma<NTV, VMODE>), grid, block, 0, st, A, R4, v, (int)c, (int)c, tpart, (int)N, \
                     sp.rows, f, stop)
  if (NT == 1) LO_T(1);
  else if (NT == 2) LO_T(2);
  else if (NT == 3) LO_T(3);
  else LO_T(4);
#undef LO_T
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int skinny_tn_mfma(int vmode, const float* A, int R4, float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
                   const TnFuseM& f, const int* stop, hipStream_t st) {
  if (vmode == 0) return tn_mfma_launch<0>(A, R4, v, c, tpart, B, N, sp, f, stop, st);
  if (vmode == 1) return tn_mfma_launch<1>(A, R4, v, c, tpart, B, N, sp, f, stop, st);
  return tn_mfma_launch<2>(A, R4, v, c, tpart, B, N, sp, f, stop, st);
}

int skinny_nn_mfma(const float* A, int R4, const float* tpart, const float* dd, int dd_mode, float sgn, const float* v,
                   int64_t c, float* y, float* dot_part, int64_t B, int64_t N, Split sp, const int* stop,
                   hipStream_t st) {
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  const size_t shm = (size_t)R4 * 32 * sizeof(float);
  LO_PROF_BEGIN(R4 == 32 ? "skinny_nn_mfma_R32" : (R4 == 16 ? "skinny_nn_mfma_R16" : "skinny_nn_mfma"), st);
#define LO_NNM(KQV)                                                                                                   \
  do {                                                                                                                \
    if (dot_part)                                                                                                     \
      hipLaunchKernelGGL((k_skinny_nn_mfma<true, KQV>), grid, block, shm, st, A, R4, tpart, dd, dd_mode, sgn, v, (int)c, \
                         (int)c, y, dot_part, (int)c, (int)N, sp.rows, stop);                                         \
    else                                                                                                              \
      hipLaunchKernelGGL((k_skinny_nn_mfma<false, KQV>), grid, block, shm, st, A, R4, tpart, dd, dd_mode, sgn, v,     \
                         (int)c, (int)c, y, dot_part, (int)c, (int)N, sp.rows, stop);                                 \
  } while (0)
  switch (R4 / 8) {
    case 1: LO_NNM(1); break;
    case 2: LO_NNM(2); break;
    case 4: LO_NNM(4); break;
    case 8: LO_NNM(8); break;
    case 16: LO_NNM(16); break;
    default: return LO_ERR_UNSUPPORTED;
  }
#undef LO_NNM
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // namespace lo

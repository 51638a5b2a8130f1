// lo_kron.hip -- y = (K1 (x) K2) v  ==  vec(K1 V K2^T), V = v viewed as [n1, n2, c]
// (reference: module-level _matmul of operators/kronecker_product_linear_operator.py:34-45: per factor a
//  view [n_i, -1], a batched matmul, and a transposing reshape).  Two batched strided GEMMs, no
//  transposing copies:
//    W[i1,(j2,col)]  = sum_j1 K1[i1,j1] V[j1,(j2,col)]                (M=n1, K=n1, N=n2*c)
//    Y[i1,i2,col]    = sum_j2 W[i1,j2,col] K2[i2,j2]   per (b,col)    (M=n1, K=n2, N=n2)
// Two engines: a generic strided VALU GEMM (any n1, n2, c: 64x64 tile, BK=16, 4x4 register micro-tile) and, for one
// right-hand-side column and factors that are multiples of 128, an MFMA engine (v_mfma_f32_32x32x2_f32): both
// products in the "NT" form D = A B^T with k-contiguous operands,
//    Tt[j2, i1] = sum_i2 K2[j2, i2] V[i1, i2]          (the intermediate is produced transposed, coalesced)
//    Y[j1, j2]  = sum_i1 K1[j1, i1] Tt[j2, i1]  + d o v, with the CG inner product sum v o y fused in the epilogue
// (~64 flop/B, the one matrix-core-bound operator of the path).
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct GemmArgs {
  const float* A;
  const float* Bm;
  float* C;
  int M, N, K;
  int64_t sa_r, sa_c, sb_r, sb_c, sc_r, sc_c;
  // batch z = zo * inner + zi
  int inner;
  int64_t a_zo, a_zi, b_zo, b_zi, c_zo, c_zi;
  int accumulate;  // C += A B instead of C = A B
};

constexpr int BM = 64, BN = 64, BK = 16;

__global__ __launch_bounds__(kThreads) void k_gemm(GemmArgs g, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  const int z = blockIdx.z;
  const int zo = z / g.inner, zi = z % g.inner;
  const float* A = g.A + zo * g.a_zo + zi * g.a_zi;
  const float* Bm = g.Bm + zo * g.b_zo + zi * g.b_zi;
  float* C = g.C + zo * g.c_zo + zi * g.c_zi;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;  // 16 x 16 threads, 4x4 each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < g.K; k0 += BK) {
    // stage A tile [BM x BK] and B tile [BK x BN]
    for (int e = threadIdx.x; e < BM * BK; e += kThreads) {
      int mm, kk;
      if (g.sa_c == 1) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < g.M && gk < g.K) ? A[gm * g.sa_r + gk * g.sa_c] : 0.f;
    }
    for (int e = threadIdx.x; e < BK * BN; e += kThreads) {
      int nn, kk;
      if (g.sb_c == 1) { nn = e % BN; kk = e / BN; } else { kk = e % BK; nn = e / BK; }
      const int gn = n0 + nn, gk = k0 + kk;
      Bs[kk][nn] = (gn < g.N && gk < g.K) ? Bm[gk * g.sb_r + gn * g.sb_c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&As[kk][4 * ty]);
      const float4 bq = *reinterpret_cast<const float4*>(&Bs[kk][4 * tx]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + 4 * ty + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + 4 * tx + j;
      if (gn < g.N) {
        float* o = C + gm * g.sc_r + gn * g.sc_c;
        *o = g.accumulate ? *o + acc[i][j] : acc[i][j];
      }
    }
  }
}

static int launch_gemm(const GemmArgs& g, int nbatch, const int* stop, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nbatch);
  LO_PROF_BEGIN("kron_gemm", st);
  hipLaunchKernelGGL(k_gemm, grid, dim3(kThreads), 0, st, g, stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// ---- MFMA engine --------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int KM_BM = 128, KM_BN = 128, KM_BK = 64, KM_LD = KM_BK + 4;

struct KmArgs {
  const float* A;   // [z][M][K]
  const float* Bm;  // [z][N][K]
  float* D;         // [z][M][N]
  int M, N, K, B;
  // epilogue (EPI): y = D + diag o v, dot partial sum v o y per workgroup tile
  const float* diag;
  int diag_mode;
  const float* v;   // [z][M][N]
  float* dot_part;  // [z][tiles] or nullptr
  int a_div;        // operand A of batch item z is matrix z / a_div (several right-hand-side columns share a factor)
};

__device__ __forceinline__ int km_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// D = A B^T for one 128 x 128 tile, K in slabs of 64 through LDS (row stride 68 floats: conflict-free ds_read_b128
// of the MFMA operands), next slab fetched into registers while the current one is multiplied.  4 waves, each a
// 64 x 64 quadrant = 2 x 2 MFMA tiles of 32 x 32.
// GUARD: M, N, K are multiples of 4 but not of the tile (factor sizes of real Kronecker kernels are arbitrary): rows /
// columns beyond the matrices are loaded as zeros and not stored.
template <bool EPI, bool GUARD = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_kron_nt_mfma(
    KmArgs g, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float a_s[KM_BM * KM_LD];
  __shared__ float b_s[KM_BN * KM_LD];
  __shared__ float dot_s[4];
  // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs, so all tiles of a member get ids
  // of the same residue mod 8 -> one XCD, one L2: the member's operands are fetched from HBM once, not once per XCD
  const int tiles_n = (g.N + KM_BN - 1) / KM_BN, tiles = tiles_n * ((g.M + KM_BM - 1) / KM_BM);
  const int id = blockIdx.x, xcd = id & 7, rest = id >> 3;
  const int tile = rest % tiles, z = (rest / tiles) * 8 + xcd;
  if (z >= g.B) return;
  const int m0 = (tile / tiles_n) * KM_BM, n0 = (tile % tiles_n) * KM_BN;
  const float* A = g.A + (size_t)(z / g.a_div) * g.M * g.K + (size_t)m0 * g.K;
  const float* Bm = g.Bm + (size_t)z * g.N * g.K + (size_t)n0 * g.K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;  // wave quadrant: rows 64 wr .. +63, columns 64 wc .. +63
  // staging map: thread t moves float4 #(t + 256 u): row = f / 16, quad = f % 16 of the [128][64] slabs.  Sixteen
  // NAMED registers, not arrays: the compiler demoted loop-carried float4 arrays to LDS (promote-alloca), which put a
  // wait right behind the loads and serialised fetch and MFMA; the loads are UNCONDITIONAL (the last iteration
  // re-fetches its own slab) for the same reason.
  const int sr = threadIdx.x >> 4, sq = threadIdx.x & 15;  // + 16 rows per u
  const float* ag = A + (size_t)sr * g.K + 4 * sq;
  const float* bg = Bm + (size_t)sr * g.K + 4 * sq;
  const size_t rs = (size_t)16 * g.K;
  float* al = &a_s[sr * KM_LD + 4 * sq];
  float* bl = &b_s[sr * KM_LD + 4 * sq];
  float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
#define KM_LD4(p_) (*reinterpret_cast<const float4*>(p_))
  // (guarded variant: float4 of row sr + 16 u at k if inside the matrix, else zeros)
  auto ldg = [&](const float* base, int row_lim, int u, int kq) -> float4 {
    return (sr + 16 * u < row_lim && kq + 4 * sq < g.K) ? KM_LD4(base + (size_t)u * rs + kq)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const int a_lim = g.M - m0, b_lim = g.N - n0;  // valid rows of this tile's operands
#define KM_LOADG(k0_)                                                                                        \
  ra0 = ldg(ag, a_lim, 0, k0_); ra1 = ldg(ag, a_lim, 1, k0_); ra2 = ldg(ag, a_lim, 2, k0_);                    \
  ra3 = ldg(ag, a_lim, 3, k0_); ra4 = ldg(ag, a_lim, 4, k0_); ra5 = ldg(ag, a_lim, 5, k0_);                    \
  ra6 = ldg(ag, a_lim, 6, k0_); ra7 = ldg(ag, a_lim, 7, k0_);                                                  \
  rb0 = ldg(bg, b_lim, 0, k0_); rb1 = ldg(bg, b_lim, 1, k0_); rb2 = ldg(bg, b_lim, 2, k0_);                    \
  rb3 = ldg(bg, b_lim, 3, k0_); rb4 = ldg(bg, b_lim, 4, k0_); rb5 = ldg(bg, b_lim, 5, k0_);                    \
  rb6 = ldg(bg, b_lim, 6, k0_); rb7 = ldg(bg, b_lim, 7, k0_);
#define KM_LOAD(k0_)                                                                                         \
  ra0 = KM_LD4(ag + (k0_)); ra1 = KM_LD4(ag + rs + (k0_)); ra2 = KM_LD4(ag + 2 * rs + (k0_));               \
  ra3 = KM_LD4(ag + 3 * rs + (k0_)); ra4 = KM_LD4(ag + 4 * rs + (k0_)); ra5 = KM_LD4(ag + 5 * rs + (k0_));   \
  ra6 = KM_LD4(ag + 6 * rs + (k0_)); ra7 = KM_LD4(ag + 7 * rs + (k0_));                                      \
  rb0 = KM_LD4(bg + (k0_)); rb1 = KM_LD4(bg + rs + (k0_)); rb2 = KM_LD4(bg + 2 * rs + (k0_));               \
  rb3 = KM_LD4(bg + 3 * rs + (k0_)); rb4 = KM_LD4(bg + 4 * rs + (k0_)); rb5 = KM_LD4(bg + 5 * rs + (k0_));   \
  rb6 = KM_LD4(bg + 6 * rs + (k0_)); rb7 = KM_LD4(bg + 7 * rs + (k0_));
#define KM_ST4(p_, v_) (*reinterpret_cast<float4*>(p_) = (v_))
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc00[e] = 0.f;
    acc01[e] = 0.f;
    acc10[e] = 0.f;
    acc11[e] = 0.f;
  }
  if constexpr (GUARD) {
    KM_LOADG(0)
  } else {
    KM_LOAD(0)
  }
  for (int k0 = 0; k0 < g.K; k0 += KM_BK) {
    __syncthreads();
    KM_ST4(al, ra0); KM_ST4(al + 16 * KM_LD, ra1); KM_ST4(al + 32 * KM_LD, ra2); KM_ST4(al + 48 * KM_LD, ra3);
    KM_ST4(al + 64 * KM_LD, ra4); KM_ST4(al + 80 * KM_LD, ra5); KM_ST4(al + 96 * KM_LD, ra6);
    KM_ST4(al + 112 * KM_LD, ra7);
    KM_ST4(bl, rb0); KM_ST4(bl + 16 * KM_LD, rb1); KM_ST4(bl + 32 * KM_LD, rb2); KM_ST4(bl + 48 * KM_LD, rb3);
    KM_ST4(bl + 64 * KM_LD, rb4); KM_ST4(bl + 80 * KM_LD, rb5); KM_ST4(bl + 96 * KM_LD, rb6);
    KM_ST4(bl + 112 * KM_LD, rb7);
    __syncthreads();
    if constexpr (GUARD) {
      const int kn = k0 + KM_BK;  // (beyond K: the guard returns zeros, the loop ends before they are used)
      KM_LOADG(kn)
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const int kn = min(k0 + KM_BK, g.K - KM_BK);
      KM_LOAD(kn)
      __builtin_amdgcn_sched_barrier(0);  // the scheduler would otherwise sink the loads below the MFMAs
    }
    // lane half h takes k in [32 h, 32 h + 32) of the slab: a float4 feeds 4 consecutive MFMA steps
    const float* a0 = &a_s[(64 * wr + li) * KM_LD + 32 * h];
    const float* a1 = a0 + 32 * KM_LD;
    const float* b0 = &b_s[(64 * wc + li) * KM_LD + 32 * h];
    const float* b1 = b0 + 32 * KM_LD;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 p4 = *reinterpret_cast<const float4*>(a0 + 4 * q);
      const float4 r4 = *reinterpret_cast<const float4*>(a1 + 4 * q);
      const float4 x4 = *reinterpret_cast<const float4*>(b0 + 4 * q);
      const float4 y4 = *reinterpret_cast<const float4*>(b1 + 4 * q);
#define KM_STEP(c_)                                                                  \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.c_, x4.c_, acc00, 0, 0, 0);        \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(p4.c_, y4.c_, acc01, 0, 0, 0);        \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(r4.c_, x4.c_, acc10, 0, 0, 0);        \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(r4.c_, y4.c_, acc11, 0, 0, 0);
      KM_STEP(x) KM_STEP(y) KM_STEP(z) KM_STEP(w)
#undef KM_STEP
    }
  }
#undef KM_LOAD
#undef KM_LOADG
#undef KM_LD4
#undef KM_ST4
  float* D = g.D + (size_t)z * g.M * g.N;
  const float dconst = (EPI && g.diag_mode == LO_DIAG_CONST) ? g.diag[z] : 0.f;
  float dacc = 0.f;
  auto epilogue = [&](const f32x16& acc, int rt, int ct) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + 64 * wr + 32 * rt + km_row(e, lane);
      const int colg = n0 + 64 * wc + 32 * ct + li;
      if (GUARD && (row >= g.M || colg >= g.N)) continue;
      const size_t o = (size_t)row * g.N + colg;
      float yv = acc[e];
      if (EPI) {
        const float vin = g.v[(size_t)z * g.M * g.N + o];
        const float dv = (g.diag_mode == LO_DIAG_FULL) ? g.diag[(size_t)z * g.M * g.N + o] : dconst;
        yv = fmaf(dv, vin, yv);
        dacc = fmaf(vin, yv, dacc);
      }
      D[o] = yv;
    }
  };
  epilogue(acc00, 0, 0);
  epilogue(acc01, 0, 1);
  epilogue(acc10, 1, 0);
  epilogue(acc11, 1, 1);
  if (EPI && g.dot_part) {
    dacc = wave_sum(dacc);
    if (lane == 0) dot_s[wave] = dacc;
    __syncthreads();
    if (threadIdx.x == 0) g.dot_part[(size_t)z * tiles + tile] = (dot_s[0] + dot_s[1]) + (dot_s[2] + dot_s[3]);
  }
}

// ---- both GEMMs of the Kronecker matvec in ONE kernel (round 3) -----------------------------------------------------
// Y = K1 (V K2^T) + d o V for V [n1, n2], n1 = 128 or 256.  A workgroup owns 128 columns j of Y (wave w: 32 of them) and
// ALL n1 rows, so the intermediate T = V K2^T never leaves the register file:
//   stage 1   T[i, j] = sum_l V[i, l] K2[j, l]     wave w: n1 x 32 block = n1 / 32 accumulator tiles of 32 x 32
//   stage 2   Y[m, j] = sum_i K1[m, i] T[i, j]     T is the B operand STRAIGHT FROM THE ACCUMULATORS:
// register r of an accumulator tile holds T[32 ib + 8 (r / 4) + 4 h + r % 4][j] in lane (j, h) -- exactly the
// (k = h, n = j) layout of a 32x32x2 B operand whose two contraction indices are i and i + 4; the A operand supplies
// K1[m][i + 4 h] to match (four consecutive registers = a float4 of K1 at column 32 ib + 8 (r / 4) + 4 h).  The order of
// the contraction is free, the values are the same products.  V / K1 (n1 x 32 slabs) and K2 (128 x 32) pass through
// double-buffered LDS; the slab after next is fetched into registers behind the matrix-core stream.
// Per member: V, K1 read by n2 / 128 workgroups of one XCD (L2), K2 and Y once -- the 2 x n1 n2 floats of the
// intermediate (67 MB per launch at cfg4, a third of the two-launch traffic) and one launch are gone.
struct KfArgs {
  const float* K1;  // [B][n1][n1]
  const float* K2;  // [B][n2][n2]
  const float* v;   // [B][n1][n2]
  float* y;
  const float* diag;
  int diag_mode;
  float* dot_part;  // [B][tiles of 128 x 128] or nullptr (layout of kron_S_dot)
  int n2, B;
};

constexpr int KF_LDE = KM_BN + 4;  // row stride of the epilogue's staging tile (128 rows x 128 columns)
// Slabs of 32 contraction indices in TWO LDS buffers: the slab after next is on its way from HBM into registers and the
// next slab moves from registers into the idle buffer while the matrix cores work on the current one -- one barrier per
// slab (a first version with 64-wide slabs in one buffer needed barrier - store - barrier: 85.8 vs 82.3 us).
template <int NI, int NS>  // n1 = 32 NI; n2 = 64 NS (NS = 0: any multiple of 128)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_kron_fused(
    KfArgs g, const int* __restrict__ stop) {
  if (stop && *stop) return;
  constexpr int N1 = 32 * NI;
  constexpr int LDD = 36;  // row stride of a 32-wide slab (floats): conflict-free ds_read_b128 / ds_write_b128
  constexpr int SA = N1 * LDD, SB = KM_BN * LDD;
  constexpr int SMEM = (2 * SA + 2 * SB > 128 * KF_LDE) ? 2 * SA + 2 * SB : 128 * KF_LDE;
  __shared__ __attribute__((aligned(16))) float sm[SMEM];  // A0 | A1 | B0 | B1; the epilogue's staging tile afterwards
  __shared__ float dot_s[4][2];
  const int n2 = NS ? 64 * NS : g.n2, nblk = n2 / KM_BN;
  const int id = blockIdx.x, xcd = id & 7, rest = id >> 3;
  const int blk = rest % nblk, z = (rest / nblk) * 8 + xcd;
  if (z >= g.B) return;
  const float* V = g.v + (size_t)z * N1 * n2;
  const float* K2 = g.K2 + (size_t)z * n2 * n2 + (size_t)blk * KM_BN * n2;
  const float* K1 = g.K1 + (size_t)z * N1 * N1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int sr = threadIdx.x >> 3, sq = threadIdx.x & 7;  // staging: float4 (row sr + 32 u, columns 4 sq ..) of a slab
  const int S1 = n2 / 32;  // slabs of stage 1; stage 2 has NI more (one accumulator tile of T each)
  float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3;  // (named: see k_kron_nt_mfma)
  ra4 = ra5 = ra6 = ra7 = make_float4(0.f, 0.f, 0.f, 0.f);
  rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define KF_LD4(p_) (*reinterpret_cast<const float4*>(p_))
#define KF_ST4(p_, v_) (*reinterpret_cast<float4*>(p_) = (v_))
#define KF2_LOAD_A(p_, ld_)                                                                              \
  ra0 = KF_LD4(p_); ra1 = KF_LD4((p_) + (size_t)32 * (ld_)); ra2 = KF_LD4((p_) + (size_t)64 * (ld_));    \
  ra3 = KF_LD4((p_) + (size_t)96 * (ld_));                                                               \
  if constexpr (NI == 8) {                                                                               \
    ra4 = KF_LD4((p_) + (size_t)128 * (ld_)); ra5 = KF_LD4((p_) + (size_t)160 * (ld_));                  \
    ra6 = KF_LD4((p_) + (size_t)192 * (ld_)); ra7 = KF_LD4((p_) + (size_t)224 * (ld_));                  \
  }
#define KF2_LOAD_B(p_, ld_)                                                                              \
  rb0 = KF_LD4(p_); rb1 = KF_LD4((p_) + (size_t)32 * (ld_)); rb2 = KF_LD4((p_) + (size_t)64 * (ld_));    \
  rb3 = KF_LD4((p_) + (size_t)96 * (ld_));
  // slab s_ of the whole sequence -> registers (s_ < S1: V and K2 columns 32 s_ ..; else K1 columns 32 (s_ - S1) ..)
#define KF2_LOAD(s_)                                                                                     \
  if ((s_) < S1) {                                                                                       \
    const float* vg_ = V + (size_t)sr * n2 + 32 * (s_) + 4 * sq;                                         \
    const float* kg_ = K2 + (size_t)sr * n2 + 32 * (s_) + 4 * sq;                                        \
    KF2_LOAD_A(vg_, n2)                                                                                  \
    KF2_LOAD_B(kg_, n2)                                                                                  \
  } else {                                                                                               \
    const float* kg_ = K1 + (size_t)sr * N1 + 32 * ((s_) - S1) + 4 * sq;                                 \
    KF2_LOAD_A(kg_, N1)                                                                                  \
  }
  // registers -> buffer (s_ & 1); stage-2 slabs have no B part
#define KF2_STORE(s_)                                                                                    \
  {                                                                                                      \
    float* al_ = sm + ((s_) & 1) * SA + sr * LDD + 4 * sq;                                               \
    KF_ST4(al_, ra0); KF_ST4(al_ + 32 * LDD, ra1); KF_ST4(al_ + 64 * LDD, ra2); KF_ST4(al_ + 96 * LDD, ra3); \
    if constexpr (NI == 8) {                                                                             \
      KF_ST4(al_ + 128 * LDD, ra4); KF_ST4(al_ + 160 * LDD, ra5); KF_ST4(al_ + 192 * LDD, ra6);          \
      KF_ST4(al_ + 224 * LDD, ra7);                                                                      \
    }                                                                                                    \
    if ((s_) < S1) {                                                                                     \
      float* bl_ = sm + 2 * SA + ((s_) & 1) * SB + sr * LDD + 4 * sq;                                    \
      KF_ST4(bl_, rb0); KF_ST4(bl_ + 32 * LDD, rb1); KF_ST4(bl_ + 64 * LDD, rb2); KF_ST4(bl_ + 96 * LDD, rb3); \
    }                                                                                                    \
  }
  f32x16 T[NI], Y[NI];
#pragma unroll
  for (int ib = 0; ib < NI; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      T[ib][e] = 0.f;
      Y[ib][e] = 0.f;
    }
  KF2_LOAD(0)
  KF2_STORE(0)
  KF2_LOAD(1)
  // ---- stage 1: T = V K2[j block]^T ----
#pragma unroll NS ? 2 * NS : 1  // (fully unrolled when the factor size is a template argument)
  for (int s = 0; s < (NS ? 2 * NS : S1); ++s) {
    __syncthreads();  // buffer s & 1 complete; everybody is done with the other one
    KF2_STORE(s + 1)  // (always exists: the first stage-2 slab follows the last stage-1 slab)
    KF2_LOAD(s + 2)
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would otherwise sink the loads below the matrix-core stream)
    const float* a0 = sm + (s & 1) * SA + li * LDD + 16 * h;
    const float* b0 = sm + 2 * SA + (s & 1) * SB + (32 * wave + li) * LDD + 16 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 av[NI];
#pragma unroll
      for (int ib = 0; ib < NI; ++ib) av[ib] = KF_LD4(a0 + ib * 32 * LDD + 4 * q);
      const float4 bv = KF_LD4(b0 + 4 * q);
#define KF_STEP(c_)                                                                     \
  _Pragma("unroll") for (int ib = 0; ib < NI; ++ib)                                     \
      T[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ib].c_, bv.c_, T[ib], 0, 0, 0);
      KF_STEP(x) KF_STEP(y) KF_STEP(z) KF_STEP(w)
#undef KF_STEP
    }
  }
  // ---- stage 2: Y = K1 T, slab s2 = accumulator tile s2 of T (B operand as it lies, see above) ----
#pragma unroll
  for (int s2 = 0; s2 < NI; ++s2) {
    const int s = S1 + s2;
    __syncthreads();
    if (s2 + 1 < NI) KF2_STORE(s + 1)
    if (s2 + 2 < NI) { KF2_LOAD(s + 2) }
    __builtin_amdgcn_sched_barrier(0);
    const f32x16& Tt = T[s2];
    const float* abase = sm + (s & 1) * SA + li * LDD + 4 * h;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float4 av[NI];
#pragma unroll
      for (int mb = 0; mb < NI; ++mb) av[mb] = KF_LD4(abase + mb * 32 * LDD + 8 * g4);
#define KF_STEP(c_, e_)                                                                       \
  _Pragma("unroll") for (int mb = 0; mb < NI; ++mb)                                           \
      Y[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb].c_, Tt[4 * g4 + e_], Y[mb], 0, 0, 0);
      KF_STEP(x, 0) KF_STEP(y, 1) KF_STEP(z, 2) KF_STEP(w, 3)
#undef KF_STEP
    }
  }
#undef KF2_LOAD
#undef KF2_STORE
#undef KF2_LOAD_A
#undef KF2_LOAD_B
#undef KF_LD4
#undef KF_ST4
  // ---- epilogue: 128 rows of Y at a time through LDS (the slab buffers are free), so that v is read and y written as
  //      whole 512-byte row segments; + d o v, dot partials per 128 x 128 tile (the layout the two-launch path writes) ----
  float* Yg = g.y + (size_t)z * N1 * n2;
  const float dconst = (g.diag_mode == LO_DIAG_CONST) ? g.diag[z] : 0.f;
  const int col0 = blk * KM_BN;
  const int er = threadIdx.x >> 5, ec = threadIdx.x & 31;
#pragma unroll
  for (int half = 0; half < NI / 4; ++half) {
    __syncthreads();
#pragma unroll
    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        sm[(32 * mq + km_row(e, lane)) * KF_LDE + 32 * wave + li] = Y[4 * half + mq][e];
    __syncthreads();
    float dacc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int rl = er + 8 * u;
      float4 yv = *reinterpret_cast<const float4*>(&sm[rl * KF_LDE + 4 * ec]);
      const size_t o = (size_t)(128 * half + rl) * n2 + col0 + 4 * ec;
      if (g.diag_mode != LO_DIAG_NONE) {
        const float4 vin = *reinterpret_cast<const float4*>(V + o);
        float4 dv = make_float4(dconst, dconst, dconst, dconst);
        if (g.diag_mode == LO_DIAG_FULL) dv = *reinterpret_cast<const float4*>(g.diag + (size_t)z * N1 * n2 + o);
        yv.x = fmaf(dv.x, vin.x, yv.x); yv.y = fmaf(dv.y, vin.y, yv.y);
        yv.z = fmaf(dv.z, vin.z, yv.z); yv.w = fmaf(dv.w, vin.w, yv.w);
        dacc = fmaf(vin.x, yv.x, dacc); dacc = fmaf(vin.y, yv.y, dacc);
        dacc = fmaf(vin.z, yv.z, dacc); dacc = fmaf(vin.w, yv.w, dacc);
      }
      *reinterpret_cast<float4*>(Yg + o) = yv;
    }
    if (g.dot_part) {
      dacc = wave_sum(dacc);
      if (lane == 0) dot_s[wave][half] = dacc;
    }
  }
  if (g.dot_part) {
    __syncthreads();
    if (threadIdx.x < NI / 4) {
      const int rt = threadIdx.x;
      g.dot_part[(size_t)z * (NI / 4) * nblk + rt * nblk + blk] =
          (dot_s[0][rt] + dot_s[1][rt]) + (dot_s[2][rt] + dot_s[3][rt]);
    }
  }
}

bool kron_fused_ok(int n1, int n2) {
  return (n1 == 128 || n1 == 256) && n2 % 128 == 0 && n2 >= 128 && !getenv("LO_NO_KRON_FUSED");
}

static bool km_aligned(int n1, int n2) { return n1 % 128 == 0 && n2 % 128 == 0; }
// (factor sizes that are multiples of 4 and at least 64: tiles beyond the matrices are guarded; smaller or odd sizes
//  stay on the strided VALU GEMM)
bool kron_mfma_ok(int n1, int n2, int64_t c) { return c == 1 && n1 % 4 == 0 && n2 % 4 == 0 && n1 >= 64 && n2 >= 64; }
int kron_S_dot(int n1, int n2, int64_t c, int S_default) {
  return kron_mfma_ok(n1, n2, c) ? ((n1 + KM_BM - 1) / KM_BM) * ((n2 + KM_BN - 1) / KM_BN) : S_default;
}

template <bool EPI>
static void km_launch(const KmArgs& g, int64_t Z, bool guard, const int* stop, hipStream_t st) {
  const unsigned tiles = (unsigned)(((g.M + KM_BM - 1) / KM_BM) * ((g.N + KM_BN - 1) / KM_BN));
  const dim3 grid((unsigned)(((Z + 7) / 8) * 8) * tiles);
  if (guard) hipLaunchKernelGGL((k_kron_nt_mfma<EPI, true>), grid, dim3(kThreads), 0, st, g, stop);
  else hipLaunchKernelGGL((k_kron_nt_mfma<EPI, false>), grid, dim3(kThreads), 0, st, g, stop);
}

// y = (K1 (x) K2) v + diag o v and (optionally) the dot partials sum v o y, c == 1, matrix cores
int kron_matvec_mfma(const float* K1, const float* K2, const float* diag, int diag_mode, const float* v, float* tmp,
                     float* y, float* dot_part, int64_t B, int n1, int n2, const int* stop, hipStream_t st) {
  // both GEMMs in one launch, the intermediate stays in the accumulators -- once the batch offers a workgroup to at least
  // a third of the CUs (a small batch is spread wider by the 128 x 128 tiles of the two-launch path: 5 members of
  // 256 (x) 128: 40 us against 60)
  if (kron_fused_ok(n1, n2) && B * (n2 / KM_BN) >= 96) {
    KfArgs f;
    f.K1 = K1; f.K2 = K2; f.v = v; f.y = y; f.diag = diag; f.diag_mode = diag ? diag_mode : LO_DIAG_NONE;
    f.dot_part = dot_part; f.n2 = n2; f.B = (int)B;
    const dim3 grid((unsigned)(((B + 7) / 8) * 8) * (unsigned)(n2 / KM_BN));
    LO_PROF_BEGIN("kron_fused", st);
    if (n1 == 256 && n2 == 256) hipLaunchKernelGGL((k_kron_fused<8, 4>), grid, dim3(kThreads), 0, st, f, stop);
    else if (n1 == 256) hipLaunchKernelGGL((k_kron_fused<8, 0>), grid, dim3(kThreads), 0, st, f, stop);
    else if (n2 == 128) hipLaunchKernelGGL((k_kron_fused<4, 2>), grid, dim3(kThreads), 0, st, f, stop);
    else hipLaunchKernelGGL((k_kron_fused<4, 0>), grid, dim3(kThreads), 0, st, f, stop);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
    return LO_OK;
  }
  KmArgs g;
  g.B = (int)B;
  g.a_div = 1;
  g.A = K2; g.Bm = v; g.D = tmp;  // Tt [n2, n1]
  g.M = n2; g.N = n1; g.K = n2;
  g.diag = nullptr; g.diag_mode = LO_DIAG_NONE; g.v = nullptr; g.dot_part = nullptr;
  LO_PROF_BEGIN("kron_gemm_mfma", st);
  km_launch<false>(g, B, !km_aligned(n1, n2), stop, st);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  g.A = K1; g.Bm = tmp; g.D = y;  // Y [n1, n2]
  g.M = n1; g.N = n2; g.K = n1;
  g.diag = diag; g.diag_mode = diag ? diag_mode : LO_DIAG_NONE; g.v = v; g.dot_part = dot_part;
  LO_PROF_BEGIN("kron_gemm_mfma", st);
  km_launch<true>(g, B, !km_aligned(n1, n2), stop, st);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// ---- several columns on the matrix-core engine ---------------------------------------------------------------------
// The matrix-core GEMMs want k-contiguous operands, the vectors arrive as [B, N, c] (columns innermost).  For c > 1 the
// columns are moved to the front once ([B, c, N]: a batch of B c single-column problems that share the factors),
// both GEMMs run on the single-column engine with batch B c, and the result is moved back: two extra passes over
// 2 N c floats per member against 4.5 ms -> 2.x ms for the products at cfg4's shape with 17 columns (the strided VALU
// GEMM reaches 33 TFLOP/s, the matrix-core engine 72).
//   in [Z][n][c] -> out [Z][c][n]   (to_front)   or   in [Z][c][n] -> out [Z][n][c] (+ dd o v: the operator's diagonal
//   rides on the way back)
__global__ __launch_bounds__(kThreads) void k_kron_cols(const float* __restrict__ in, float* __restrict__ out, int n,
                                                         int c, int to_front, const float* __restrict__ dd, int dd_mode,
                                                         const float* __restrict__ vin,
                                                         const int* __restrict__ stop) {
  if (stop && *stop) return;
  extern __shared__ float tl[];  // [256][c + 1]
  const int64_t z = blockIdx.y;
  const int r0 = blockIdx.x * kThreads;
  const int nr = min(kThreads, n - r0);
  const int ld = c | 1;
  if (to_front) {
    const float* src = in + ((size_t)z * n + r0) * c;  // nr * c contiguous floats
    for (int e = threadIdx.x; e < nr * c; e += kThreads) tl[(e / c) * ld + e % c] = src[e];
    __syncthreads();
    for (int col = 0; col < c; ++col)
      if (threadIdx.x < nr) out[((size_t)z * c + col) * n + r0 + threadIdx.x] = tl[threadIdx.x * ld + col];
  } else {
    for (int col = 0; col < c; ++col)
      if (threadIdx.x < nr) tl[threadIdx.x * ld + col] = in[((size_t)z * c + col) * n + r0 + threadIdx.x];
    __syncthreads();
    float* dst = out + ((size_t)z * n + r0) * c;
    const float* vs = vin + ((size_t)z * n + r0) * c;
    const float dconst = (dd_mode == LO_DIAG_CONST) ? dd[z] : 0.f;
    for (int e = threadIdx.x; e < nr * c; e += kThreads) {
      float val = tl[(e / c) * ld + e % c];
      if (dd_mode == LO_DIAG_FULL) val = fmaf(dd[(size_t)z * n + r0 + e / c], vs[e], val);
      else if (dd_mode == LO_DIAG_CONST) val = fmaf(dconst, vs[e], val);
      dst[e] = val;
    }
  }
}

bool kron_mfma_cols_ok(int n1, int n2, int64_t c) { return c > 1 && c <= 64 && kron_mfma_ok(n1, n2, 1); }

// y = (K1 (x) K2) v + dd o v for v, y [B, N, c]; buf_a, buf_b: B N c floats each
int kron_matvec_mfma_cols(const float* K1, const float* K2, const float* diag, int diag_mode, const float* v,
                          float* buf_a, float* buf_b, float* y, int64_t B, int n1, int n2, int64_t c, const int* stop,
                          hipStream_t st) {
  const int N = n1 * n2;
  const size_t lds = sizeof(float) * (size_t)kThreads * ((size_t)c | 1);
  dim3 tgrid((unsigned)((N + kThreads - 1) / kThreads), (unsigned)B);
  LO_PROF_BEGIN("kron_cols", st);
  hipLaunchKernelGGL(k_kron_cols, tgrid, dim3(kThreads), lds, st, v, buf_a, N, (int)c, 1, nullptr, LO_DIAG_NONE, nullptr,
                     stop);
  LO_PROF_END(st);
  KmArgs g;
  const int64_t Z = B * c;
  g.B = (int)Z;
  g.a_div = (int)c;
  g.diag = nullptr; g.diag_mode = LO_DIAG_NONE; g.v = nullptr; g.dot_part = nullptr;
  g.A = K2; g.Bm = buf_a; g.D = buf_b;  // Tt [n2, n1] per (member, column)
  g.M = n2; g.N = n1; g.K = n2;
  LO_PROF_BEGIN("kron_gemm_mfma", st);
  km_launch<false>(g, Z, !km_aligned(n1, n2), stop, st);
  LO_PROF_END(st);
  g.A = K1; g.Bm = buf_b; g.D = buf_a;  // Y [n1, n2]
  g.M = n1; g.N = n2; g.K = n1;
  LO_PROF_BEGIN("kron_gemm_mfma", st);
  km_launch<false>(g, Z, !km_aligned(n1, n2), stop, st);
  LO_PROF_END(st);
  LO_PROF_BEGIN("kron_cols", st);
  hipLaunchKernelGGL(k_kron_cols, tgrid, dim3(kThreads), lds, st, buf_a, y, N, (int)c, 0, diag,
                     diag ? diag_mode : LO_DIAG_NONE, v, stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int kron_matvec(const float* K1, const float* K2, const float* v, float* tmp, float* y, int64_t B, int n1, int n2,
                int64_t c, const int* stop, hipStream_t st) {
  const int64_t N = (int64_t)n1 * n2;
  GemmArgs g1;
  g1.A = K1; g1.Bm = v; g1.C = tmp;
  g1.M = n1; g1.K = n1; g1.N = (int)(n2 * c);
  g1.sa_r = n1; g1.sa_c = 1; g1.sb_r = n2 * c; g1.sb_c = 1; g1.sc_r = n2 * c; g1.sc_c = 1;
  g1.inner = 1;
  g1.accumulate = 0;
  g1.a_zo = (int64_t)n1 * n1; g1.a_zi = 0; g1.b_zo = N * c; g1.b_zi = 0; g1.c_zo = N * c; g1.c_zi = 0;
  int rc = launch_gemm(g1, (int)B, stop, st);
  if (rc) return rc;
  GemmArgs g2;
  g2.A = tmp; g2.Bm = K2; g2.C = y;
  g2.M = n1; g2.K = n2; g2.N = n2;
  g2.sa_r = n2 * c; g2.sa_c = c; g2.sb_r = 1; g2.sb_c = n2; g2.sc_r = n2 * c; g2.sc_c = c;
  g2.inner = (int)c;
  g2.accumulate = 0;
  g2.a_zo = N * c; g2.a_zi = 1; g2.b_zo = (int64_t)n2 * n2; g2.b_zi = 0; g2.c_zo = N * c; g2.c_zi = 1;
  if (B * c > 65535) return LO_ERR_UNSUPPORTED;
  return launch_gemm(g2, (int)(B * c), stop, st);
}

// d/dK1, d/dK2 of sum_d u_d^T (K1 (x) K2) v_d (generic autograd of the Kronecker matvec, reference
// operators/_linear_operator.py:336-393 over kronecker_product_linear_operator.py:34-45): with U_d, V_d the [n1, n2]
// views of the vectors,  dK1 = sum_d U_d K2 V_d^T,  dK2 = sum_d U_d^T K1 V_d.  Four strided GEMM stages:
//   T[i1,j2,d] = sum_i2 V[i1,i2,d] K2[j2,i2]            S[i1,(j2,d)] = sum_j1 K1[i1,j1] V[j1,(j2,d)]
//   dK1[j1,i1] = sum_(j2,d) U[j1,(j2,d)] T[i1,(j2,d)]   dK2[a,b] += sum_j1 U[j1,a,d] S[j1,b,d]   for every d
int kron_bilinear(const float* K1, const float* K2, const float* U, const float* V, float* tmp, float* dK1, float* dK2,
                  int64_t B, int n1, int n2, int64_t D, hipStream_t st) {
  const int64_t N = (int64_t)n1 * n2;
  if (B * D > 65535) return LO_ERR_UNSUPPORTED;
  GemmArgs g;
  g.accumulate = 0;
  // T = (I (x) K2) V  -> tmp, batches (b, d)
  g.A = V; g.Bm = K2; g.C = tmp;
  g.M = n1; g.K = n2; g.N = n2;
  g.sa_r = n2 * D; g.sa_c = D; g.sb_r = 1; g.sb_c = n2; g.sc_r = n2 * D; g.sc_c = D;
  g.inner = (int)D;
  g.a_zo = N * D; g.a_zi = 1; g.b_zo = (int64_t)n2 * n2; g.b_zi = 0; g.c_zo = N * D; g.c_zi = 1;
  int rc = launch_gemm(g, (int)(B * D), nullptr, st);
  if (rc) return rc;
  // dK1 = U T^T over k = (j2, d)
  g.A = U; g.Bm = tmp; g.C = dK1;
  g.M = n1; g.K = (int)(n2 * D); g.N = n1;
  g.sa_r = n2 * D; g.sa_c = 1; g.sb_r = 1; g.sb_c = n2 * D; g.sc_r = n1; g.sc_c = 1;
  g.inner = 1;
  g.a_zo = N * D; g.a_zi = 0; g.b_zo = N * D; g.b_zi = 0; g.c_zo = (int64_t)n1 * n1; g.c_zi = 0;
  rc = launch_gemm(g, (int)B, nullptr, st);
  if (rc) return rc;
  // S = (K1 (x) I) V -> tmp
  g.A = K1; g.Bm = V; g.C = tmp;
  g.M = n1; g.K = n1; g.N = (int)(n2 * D);
  g.sa_r = n1; g.sa_c = 1; g.sb_r = n2 * D; g.sb_c = 1; g.sc_r = n2 * D; g.sc_c = 1;
  g.inner = 1;
  g.a_zo = (int64_t)n1 * n1; g.a_zi = 0; g.b_zo = N * D; g.b_zi = 0; g.c_zo = N * D; g.c_zi = 0;
  rc = launch_gemm(g, (int)B, nullptr, st);
  if (rc) return rc;
  // dK2[a, b] (+)= sum_j1 U[j1, a, d] S[j1, b, d], one accumulating GEMM per d
  for (int64_t d = 0; d < D; ++d) {
    g.A = U + d; g.Bm = tmp + d; g.C = dK2;
    g.M = n2; g.K = n1; g.N = n2;
    g.sa_r = D; g.sa_c = n2 * D; g.sb_r = n2 * D; g.sb_c = D; g.sc_r = n2; g.sc_c = 1;
    g.inner = 1;
    g.a_zo = N * D; g.a_zi = 0; g.b_zo = N * D; g.b_zi = 0; g.c_zo = (int64_t)n2 * n2; g.c_zi = 0;
    g.accumulate = d > 0;
    rc = launch_gemm(g, (int)B, nullptr, st);
    if (rc) return rc;
  }
  return LO_OK;
}

}  // namespace lo

// lo_solve_fused_impl.h -- ONE resident launch for the whole end-to-end solve of a low-rank-root + diagonal operator
//   A = C C^T + D:   pivoted Cholesky of C C^T  ->  root form of the Woodbury preconditioner  ->  preconditioned CG.
//
// Replaces the three-launch sequence PivotedCholesky.forward (linear_operator/functions/_pivoted_cholesky.py:14-105) ->
// AddedDiagLinearOperator._init_cache (operators/added_diag_linear_operator.py:144-184) -> linear_cg
// (utils/linear_cg.py:98-359) for the case in which the reference runs exactly its guaranteed iterations.  The
// operator is read from HBM ONCE per solve: the three-launch path reads C three times (factorisation, Gram matrix of
// the root form, CG), writes the 15 rows of L and reads them back although the root-form CG never needs L, and
// synchronises with the host twice in between.
//
// Per member (a group of GW workgroups x 256 threads x 4 rows, two workgroups of different members per CU; same
// layout as k_pc_onchip4 / k_cg_onchip5, whose phases these are):
//   1. load: the 4 C rows of a thread -> VGPRs (whole cache lines per wave instruction through a per-wave LDS window),
//      d -> 4 VGPRs;
//   2. `rank` pivots of the pivoted Cholesky, L rows in LDS, one tagged-granule exchange per pivot -- the arithmetic of
//      lo_pivchol_onchip.hip operation for operation (bit-identical pivots and L); every workgroup ALSO advances the
//      R x m recurrence  M[:, m] = (C[pi_m, :]^T - sum_{i<m} M[:, i] L[pi_m, i]) / L[pi_m, m]  (L = C M; fp64) from the
//      exchanged pivot row -- the data k_pb_rootform fetched from HBM afterwards;
//   3. E = C^T D^-1 C: rows scaled by 1/sqrt(d) staged through LDS (the wave's own rows: no workgroup barrier), fp32
//      matrix cores inside a 32-row tile, fp64 across tiles (as k_pb_gram_root), ONE group exchange of the R x R partials
//      (+ sum log d);
//   4. the small algebra of k_pb_rootform in every workgroup (fp64, LDS): G = I + M^T E M = Lg Lg^T,
//      F = (M Lg^-T)(M Lg^-T)^T, EF, logdet P -> F / EF stay in LDS;
//   5. the CG iterations of k_cg_onchip5 (root form, ONE all-reduce per iteration) on the resident rows; only the
//      solution (x * ||rhs||) is written -- no continuation state.
// What cannot be decided inside (groups work on different members at different times) is recorded and checked by
// k_fused_ctrl: the batch-global pivot rule (_pivoted_cholesky.py:57) -- exact whenever every member takes all `rank`
// pivots on its own error, otherwise the caller redoes the solve with the three-launch path -- and the batch-global
// CG stop rule at the floor (linear_cg.py:302-308).
#pragma once
#include <math.h>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_group_reduce.h"

namespace lo {

constexpr int FU_SLOT = 72;        // granules per workgroup and parity of the pivot exchange (header + C row + L entries)
constexpr int FU_HDR = 4;          // value, position, (unused), error partial
constexpr int FU_MAXRANK = 16;     // pivots (L rows of 16 floats in LDS)
constexpr int FU_INVALID = 0x7fffffff;
constexpr int FU_ESLOT = 784;      // granules per workgroup of the E exchange: 768 partials + 2 (sum log d) + padding
constexpr int FU_LD = 33;          // row stride (doubles) of the small fp64 matrices

struct FusedArgs {
  const float* C;      // [B, N, RC]
  const float* d;      // [B, N] (FULL) or [B] (CONST)
  int d_mode;
  const float* rhs;    // [B, N, c]
  float* xout;         // [B, N, c]  result * rhs_norm (linear_cg.py:335)
  int c;
  int64_t B;
  int N, RW, GW;
  int rank;            // pivots to take (<= FU_MAXRANK, <= N)
  float pc_tol;        // settings.preconditioner_tolerance
  int iters;           // CG iterations (the reference's floor)
  float eps, stop_after;
  // preconditioner out (root form; what WoodburyPreconditioner carries), any of them may be nullptr
  float* F;            // [B, RC, RC]
  float* EF;           // [B, RC, RC]
  float* E;            // [B, RC, RC]
  float* dinv;         // [B, N] or [B]
  float* logdet_p;     // [B]
  int* swaps;          // [B, rank] position exchanged with position m at pivot m (permutation = the recorded swaps)
  float* err_rec;      // [rank, B] sum |diag| over the positions >= m before pivot m
  float* orig;         // [B]
  // CG bookkeeping for k_fused_ctrl
  float* resid_rec;    // [iters, B, c]
  int* init_conv;      // [B, c]
  int* flags;          // [0] a member would stop / hit NaN before `rank` pivots (batch-global rule needed)
  // exchange buffers (zeroed by the host)
  unsigned long long* pgbuf;  // [ngroups][2][GW][FU_SLOT]
  unsigned long long* egbuf;  // [ngroups][GW][FU_ESLOT]
  unsigned long long* cgbuf;  // [ngroups][2][GW][R4_SLOT]
  int* err;
  int* next_member;
  int allow_l2_handoff;
  long long* dbg;
  int dbg_member;
};

__device__ __forceinline__ bool fu_better(float ov, int oj, float mv, int mj) {
  // FIRST maximal position wins (torch.max on CPU, _pivoted_cholesky.py:61-63)
  return oj != FU_INVALID && (mj == FU_INVALID || ov > mv || (ov == mv && oj < mj));
}
template <int M>
__device__ __forceinline__ void fu_amax_step(float& v, int& j) {
  int va, vb, ja, jb;
  bfly_i<M>(__float_as_int(v), va, vb);
  bfly_i<M>(j, ja, jb);
  const float fa = __int_as_float(va), fb = __int_as_float(vb);
  const bool tb = fu_better(fb, jb, fa, ja);
  v = tb ? fb : fa;
  j = tb ? jb : ja;
}

template <int GW>
struct alignas(16) FuPShared {
  float wv[R4_WAVES];
  int wj[R4_WAVES];
  float we[R4_WAVES];
  int pad[4];
  unsigned part[FU_SLOT];
  unsigned gath[GW][FU_SLOT];
};

__device__ __forceinline__ int fu_l_slot(int r, int q) { return r * 4 + (q ^ ((r >> 2) & 3)); }

struct FuWait {
  int* err;
  unsigned spin;
  __device__ __forceinline__ bool give_up() {
    if (++spin > R4_MAXSPIN ||
        ((spin & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 1);  // timed out, or another workgroup already did: give up at once
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
    return false;
  }
};

// thread t < cnt publishes sh.part[t] and fetches component t of every workgroup of the group into sh.gath
template <int GW>
__device__ __forceinline__ void fu_gather(FuPShared<GW>& sh, int cnt, unsigned long long* gslot_base, int wig,
                                          unsigned tag, int* err, bool same_xcd, const int t) {
  __syncthreads();  // sh.part complete
  if (t < cnt) {
    unsigned long long* slot = gslot_base + (size_t)(tag & 1u) * GW * FU_SLOT;
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)sh.part[t];
    if (same_xcd)
      __hip_atomic_store(slot + (size_t)wig * FU_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)wig * FU_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FuWait wt{err, 0};
    if constexpr (GW <= 16) {
      unsigned vals[GW];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
          const unsigned long long x =
              __hip_atomic_load(slot + (size_t)w * FU_SLOT + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(x >> 32) == tag);
          vals[w] = (unsigned)(x & 0xffffffffull);
        }
        if (ok || wt.give_up()) break;
      }
#pragma unroll
      for (int w = 0; w < GW; ++w) sh.gath[w][t] = vals[w];
    } else {
      // large groups: poll the tag words (all loads in flight together), then fetch the value words -- final once the
      // tags match: a granule of this parity is not rewritten before every workgroup has finished this exchange
      const unsigned* words = reinterpret_cast<const unsigned*>(slot);
      for (;;) {
        unsigned bad = 0;
#pragma unroll
        for (int w = 0; w < GW; ++w)
          bad |= __hip_atomic_load(words + 2 * ((size_t)w * FU_SLOT + t) + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) ^ tag;
        if (bad == 0 || wt.give_up()) break;
      }
#pragma unroll
      for (int h = 0; h < GW; h += 16) {
        unsigned v[16];
#pragma unroll
        for (int w = 0; w < 16; ++w)
          v[w] = __hip_atomic_load(words + 2 * ((size_t)(h + w) * FU_SLOT + t), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh.gath[h + w][t] = v[w];
      }
    }
  }
  __syncthreads();
}

struct alignas(16) FuPost {
  float v[32];
  float t[32];
  float w[32];  // (single column: this wave's copy of w = C^T (r / d), carried by recurrence -- k_cg_onchip5 MODE 2)
};

typedef float fu_f32x2 __attribute__((ext_vector_type(2)));
typedef float fu_f32x4 __attribute__((ext_vector_type(4)));
typedef double fu_f64x4 __attribute__((ext_vector_type(4)));

// LDS map (bytes); region A is a union over the phases
//   A  [0, 65536):       pivots: L rows (1024 x 16 floats, swizzled 16-byte slots; first the per-wave load windows)
//                        E:      row tile [256][33] floats | cross-wave partials double [4][64][12]
//                        algebra: E, T, G, Fm double [32][33]
//                        CG:     x_s, d_s, dinv_s [1024] floats | f_s, ef_s, e_s [RC][RC + 4] floats
//   B  union { FuPShared<GW> (pivots) ; R4Shared + FuPost[4] (CG) }
//   M  double [32][17]  (E phase -> algebra)
//   H  float [16][48]   pivot history: row m = the winner's C row (32) | its L entries 0..m-1, piv at m (pivots -> M)
template <int GW>
struct FuLds {
  static constexpr int kA = 65536;
  static constexpr int kBp = (int)sizeof(FuPShared<GW>);
  static constexpr int kBc = (int)(sizeof(R4Shared) + R4_WAVES * sizeof(FuPost));
  static constexpr int kB = ((kBp > kBc ? kBp : kBc) + 15) / 16 * 16;
  static constexpr int kM = 32 * 17 * 8;
  static constexpr int kH = FU_MAXRANK * 48 * 4;
  static constexpr int kTotal = kA + kB + kM + kH;
};

// 1 / sqrt(x) for x >= 1 (the pivots of I + M^T E M): v_rsq_f64 seed, two Newton steps -- ~10 dependent instructions
// where sqrt followed by a division is ~35 (these sit on the serial path of the 16-step factorisation)
__device__ __forceinline__ double fu_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

__device__ __forceinline__ double fu_readlane_d(double v, int srclane) {  // (srclane: compile-time constant at every use)
  const long long x = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)(x & 0xffffffffll), srclane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)x >> 32), srclane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int RC, int GW, bool MC>
__global__ __launch_bounds__(R4_TPB, 2) void k_solve_fused(FusedArgs a) {
  constexpr int FLD = RC + 4;
  constexpr int CH = RC / 4;
  constexpr int NB = (RC == 32) ? 3 : 1;  // 16 x 16 blocks of E computed: 00 | 00, 01, 11
  __shared__ __attribute__((aligned(16))) unsigned char smem[FuLds<GW>::kTotal];
  unsigned char* const regA = smem;
  FuPShared<GW>& shp = *reinterpret_cast<FuPShared<GW>*>(smem + FuLds<GW>::kA);
  R4Shared& sh = *reinterpret_cast<R4Shared*>(smem + FuLds<GW>::kA);
  FuPost* const post = reinterpret_cast<FuPost*>(smem + FuLds<GW>::kA + sizeof(R4Shared));
  double(*const Mm)[17] = reinterpret_cast<double(*)[17]>(smem + FuLds<GW>::kA + FuLds<GW>::kB);
  float(*const hist)[48] = reinterpret_cast<float(*)[48]>(smem + FuLds<GW>::kA + FuLds<GW>::kB + FuLds<GW>::kM);
  float4* const l_s = reinterpret_cast<float4*>(regA);
  // CG view of region A
  float* const x_s = reinterpret_cast<float*>(regA);
  float* const d_s = x_s + R4_ROWS;
  float* const dinv_s = d_s + R4_ROWS;
  float* const f_s = dinv_s + R4_ROWS;
  float* const ef_s = f_s + RC * FLD;
  float* const e_s = ef_s + RC * FLD;
  constexpr bool WR = !MC;  // one column: w by recurrence, exactly as k_cg_onchip5 MODE 2 (same bits)

  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;  // block b runs on XCD b % 8: keep a group behind one L2 (speed only)
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t0 = threadIdx.x;  // (phase code takes its own opaque copy: see the pivot phase)
  unsigned long long* const pslot = a.pgbuf + (size_t)grp * 2 * GW * FU_SLOT;
  unsigned long long* const eslot = a.egbuf + (size_t)grp * GW * FU_ESLOT;
  unsigned ptag = 0, etag = 0;
  R4Group g;
  g.gslot = a.cgbuf + (size_t)grp * 2 * GW * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  g.tag = 0;
  g.err = a.err;
  g.same_xcd = false;
  {  // placement check through the agent-scope path: plain-store hand-off only when the whole group shares an XCD
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t0 == 0) shp.part[0] = xcc;
    fu_gather<GW>(shp, 1, pslot, wig, ++ptag, a.err, false, t0);
    bool same = true;
#pragma unroll
    for (int w = 1; w < GW; ++w) same = same && (shp.gath[w][0] == shp.gath[0][0]);
    g.same_xcd = same && (a.allow_l2_handoff != 0);
    __syncthreads();
  }
  const bool same_xcd = g.same_xcd;
  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));
  const bool d_full = a.d_mode == LO_DIAG_FULL;
  const int rank = a.rank;

  int64_t b = grp;
  while (b < a.B) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t0 == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    // ================================ 1. load ================================
    fu_f32x2 Cr[R4_NR][RC / 2];
    float dq[R4_NR];
    float dg[R4_NR];
    int pos[R4_NR];
    {
      // every product and sum that feeds a pivot decision is individually rounded, in the reference's order (no FMA
      // contraction in this block and in the pivot loop: bit-identical pivots and L, see lo_pivchol_onchip.hip)
#pragma clang fp contract(off)
      int tl = t0;
      asm volatile("" : "+v"(tl));  // keeps the load-phase address arithmetic inside the member loop (VGPR budget)
      const int wv = tl >> 6, ln = tl & 63;
      // every load of the member is issued before anything waits (vmcnt counts in order); a wave fetches its 64
      // consecutive rows as consecutive 16-byte chunks, parks them in its own window of the (not yet used) L rows and
      // reads its row back; padding rows read a clamped valid row and are zeroed
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int gi = 64 * i + ln;
          const int rw = gi / CH, ck = gi % CH;
          const size_t grow = (size_t)b * a.N + min(row0 + R4_TPB * q + 64 * wv + rw, a.N - 1);
          const float4 c4 = *reinterpret_cast<const float4*>(a.C + grow * RC + 4 * ck);
          Cr[q][2 * i] = fu_f32x2{c4.x, c4.y};
          Cr[q][2 * i + 1] = fu_f32x2{c4.z, c4.w};
        }
      }
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const size_t grow = (size_t)b * a.N + min(row0 + tl + R4_TPB * q, a.N - 1);
        dq[q] = d_full ? a.d[grow] : a.d[b];
      }
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = tl + R4_TPB * q;
        const bool valid = lr < nv;
        float4* win = l_s + wv * (64 * CH);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int gi = 64 * i + ln;
          const int rw = gi / CH, ck = gi % CH;
          win[rw * CH + (ck ^ ((rw ^ (rw >> 3)) & (CH - 1)))] =
              make_float4(Cr[q][2 * i].x, Cr[q][2 * i].y, Cr[q][2 * i + 1].x, Cr[q][2 * i + 1].y);
        }
        __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order; this pins the compiler's order)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const float4 c4 = win[ln * CH + (i ^ ((ln ^ (ln >> 3)) & (CH - 1)))];
          Cr[q][2 * i] = fu_f32x2{c4.x, c4.y};
          Cr[q][2 * i + 1] = fu_f32x2{c4.z, c4.w};
        }
#pragma unroll
        for (int i = 0; i < RC / 2; ++i) Cr[q][i] = valid ? Cr[q][i] : fu_f32x2{0.f, 0.f};
        __builtin_amdgcn_wave_barrier();  // the next row set reuses the window
        // (root ** 2).sum(-1), sequential in r, products rounded before the sum (root_linear_operator.py:22-28)
        float acc = Cr[q][0].x * Cr[q][0].x;
        acc = acc + Cr[q][0].y * Cr[q][0].y;
#pragma unroll
        for (int r = 1; r < RC / 2; ++r) {
          acc = acc + Cr[q][r].x * Cr[q][r].x;
          acc = acc + Cr[q][r].y * Cr[q][r].y;
        }
        dg[q] = valid ? acc : 0.f;
        pos[q] = valid ? row0 + lr : FU_INVALID;
      }
      __syncthreads();  // the windows are done: region A is free for the row tile of the E phase
    }
    if (stamp) a.dbg[1] = wall_clock64();
    // ================================ 2. E = C^T D^-1 C: partials ================================
    // (before the pivots: region A is still free for the row tile, and the partials have the whole pivot phase to
    // travel -- they are fetched afterwards without waiting)
    // fp32 arithmetic for the row scales (IEEE division and square root, logf): the fp64 versions of the three-launch
    // path cost ~1.5 us per member here; the partial sums of log d are still accumulated in fp64
    float sq[R4_NR];  // 1 / sqrt(d) (CONST: 1, E is divided by sigma afterwards)
    float dinvq[R4_NR];
    double lsum = 0.0;
    int te = t0;
    asm volatile("" : "+v"(te));
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const int lr = te + R4_TPB * q;
      const bool valid = lr < nv;
      const float di = 1.0f / dq[q];
      sq[q] = !valid ? 0.f : (d_full ? sqrtf(di) : 1.0f);
      if (valid && d_full) lsum += (double)logf(dq[q]);
      dq[q] = valid ? dq[q] : 0.f;
      dinvq[q] = valid ? di : 0.f;
      // (the diagonal and its inverse move to their CG slots only after the E phase: region A holds the row tile now)
      if (a.dinv && valid && d_full) a.dinv[(size_t)b * a.N + row0 + lr] = di;
    }
    unsigned etg;
    {
      float* const tile = reinterpret_cast<float*>(regA);                       // [256][33]
      double* const red = reinterpret_cast<double*>(regA + 256 * 33 * 4 + 64);  // [4][64][12]  (8-byte aligned)
      const int wave_e = te >> 6, lane_e = te & 63;
      const int aa = lane_e & 15, kk = lane_e >> 4;
      double acc[NB][4];
#pragma unroll
      for (int bk = 0; bk < NB; ++bk)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[bk][r] = 0.0;
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        // every wave stages ITS OWN 64 rows (row set q) and multiplies them: no workgroup barrier
        float* my = tile + te * 33;
#pragma unroll
        for (int i = 0; i < RC / 2; ++i) {
          my[2 * i] = Cr[q][i].x * sq[q];
          my[2 * i + 1] = Cr[q][i].y * sq[q];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two 32-row tiles per row set: fp32 inside a tile, fp64 across tiles
          fu_f32x4 t00 = {0.f, 0.f, 0.f, 0.f}, t01 = t00, t11 = t00;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int row = 64 * wave_e + 32 * h + 4 * e + kk;
            const float w0 = (aa < RC) ? tile[row * 33 + aa] : 0.f;
            t00 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, w0, t00, 0, 0, 0);
            if constexpr (NB == 3) {
              const float w1 = tile[row * 33 + aa + 16];
              t01 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, w1, t01, 0, 0, 0);
              t11 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, w1, t11, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0][r] += (double)t00[r];
            if constexpr (NB == 3) {
              acc[1][r] += (double)t01[r];
              acc[2][r] += (double)t11[r];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();  // the next row set reuses the wave's tile rows
      }
#pragma unroll
      for (int bk = 0; bk < NB; ++bk)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[te * 12 + 4 * bk + r] = acc[bk][r];
      const double lw = wave_sum_d(lsum);
      if (lane_e == 0) red[4 * 64 * 12 + wave_e] = lw;
      __syncthreads();
      // entry e = 12 lane + 4 block + r of the workgroup partial: summed over the waves, published, fetched from every
      // workgroup of the group and summed in fixed order (fp64) -> identical bits in all workgroups
      const unsigned tg = ++etag;
      etg = tg;
      constexpr int NE = 64 * 12;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int e = te + R4_TPB * s;
        const double v = (red[e] + red[NE + e]) + (red[2 * NE + e] + red[3 * NE + e]);
        const bool used = (NB == 3) || ((e % 12) < 4);
        if (used) {
          const unsigned long long mine = ((unsigned long long)tg << 32) | (unsigned long long)__float_as_uint((float)v);
          __hip_atomic_store(eslot + (size_t)wig * FU_ESLOT + e, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (te < 2) {  // sum log d of the workgroup as a (hi, lo) pair of floats
        const double ls = (red[4 * NE] + red[4 * NE + 1]) + (red[4 * NE + 2] + red[4 * NE + 3]);
        const float hi = (float)ls;
        const float lo = (float)(ls - (double)hi);
        const unsigned long long mine =
            ((unsigned long long)tg << 32) | (unsigned long long)__float_as_uint(te == 0 ? hi : lo);
        __hip_atomic_store(eslot + (size_t)wig * FU_ESLOT + NE + te, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // tile / red are dead: the L rows can be cleared
    {
      int tz = t0;
      asm volatile("" : "+v"(tz));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) l_s[fu_l_slot(tz + R4_TPB * q, i)] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (stamp) a.dbg[8] = wall_clock64();


    // ================================ 3. pivots ================================
    bool early = false;  // this member's own error fell to the tolerance (or went NaN) before `rank` pivots
    float orig = 0.f;
    int tp = t0;  // (opaque per phase: LDS addresses derived from it are not hoisted out of the MEMBER loop, where they
                 // would stay live through every other phase and spill)
    asm volatile("" : "+v"(tp));
    const int lane = tp & 63, wave = tp >> 6;
    for (int m = 0; m < rank; ++m) {
#pragma clang fp contract(off)
      // ---- workgroup candidate: argmax of the running diagonal over the positions >= m, error 1-norm partial ----
      float bv = -INFINITY, es = 0.f;
      int bj = FU_INVALID;
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const bool cand = pos[q] != FU_INVALID && pos[q] >= m;
        if (cand) {
          es += fabsf(dg[q]);
          if (fu_better(dg[q], pos[q], bv, bj)) {
            bv = dg[q];
            bj = pos[q];
          }
        }
      }
      fu_amax_step<1>(bv, bj); fu_amax_step<2>(bv, bj); fu_amax_step<4>(bv, bj);
      fu_amax_step<8>(bv, bj); fu_amax_step<16>(bv, bj); fu_amax_step<32>(bv, bj);
      es = wave_sum_fast(es);
      if (lane == 0) {
        shp.wv[wave] = bv;
        shp.wj[wave] = bj;
        shp.we[wave] = es;
      }
      __syncthreads();
      float gv = shp.wv[lane & 3];
      int gj = shp.wj[lane & 3];
      float ge = shp.we[lane & 3];
      ge = bfly_add<1>(ge); ge = bfly_add<2>(ge);
      fu_amax_step<1>(gv, gj); fu_amax_step<2>(gv, gj);
      gv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gv)));
      gj = __builtin_amdgcn_readfirstlane(gj);
      ge = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ge)));
      // the owner of the candidate publishes it: header, its C row, its L entries 0..m-1
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        if (pos[q] != FU_INVALID && pos[q] >= m && pos[q] == gj) {
          shp.part[0] = __float_as_uint(gv);
          shp.part[1] = (unsigned)gj;
          float4* dst = reinterpret_cast<float4*>(&shp.part[FU_HDR]);
#pragma unroll
          for (int r = 0; r < RC / 4; ++r)
            dst[r] = make_float4(Cr[q][2 * r].x, Cr[q][2 * r].y, Cr[q][2 * r + 1].x, Cr[q][2 * r + 1].y);
          const int lr = tp + R4_TPB * q;
          for (int j4 = 0; 4 * j4 < m; ++j4) dst[RC / 4 + j4] = l_s[fu_l_slot(lr, j4)];
        }
      }
      if (tp == 0) {
        if (gj == FU_INVALID) {
          shp.part[0] = __float_as_uint(-INFINITY);
          shp.part[1] = (unsigned)FU_INVALID;
        }
        shp.part[3] = __float_as_uint(ge);
      }
      fu_gather<GW>(shp, FU_HDR + RC + m, pslot, wig, ++ptag, a.err, same_xcd, tp);

      // ---- group winner (identical in all workgroups): lane l holds candidate l % GW ----
      float vb = __uint_as_float(shp.gath[lane % GW][0]);
      const int myj = (int)shp.gath[lane % GW][1];
      int jb = myj;
      float etot = __uint_as_float(shp.gath[lane % GW][3]);
      if constexpr (GW >= 2) { etot = bfly_add<1>(etot); fu_amax_step<1>(vb, jb); }
      if constexpr (GW >= 4) { etot = bfly_add<2>(etot); fu_amax_step<2>(vb, jb); }
      if constexpr (GW >= 8) { etot = bfly_add<4>(etot); fu_amax_step<4>(vb, jb); }
      if constexpr (GW >= 16) { etot = bfly_add<8>(etot); fu_amax_step<8>(vb, jb); }
      if constexpr (GW == 32) { etot = bfly_add<16>(etot); fu_amax_step<16>(vb, jb); }
      vb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vb)));
      jb = __builtin_amdgcn_readfirstlane(jb);
      etot = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(etot)));
      const unsigned long long wbal = __ballot(lane < GW && myj == jb);
      const int wb = wbal ? __ffsll((long long)wbal) - 1 : 0;
      if (m == 0) orig = vb;  // max of the initial diagonal (:43)
      // the reference takes pivot m >= 1 only while max_b error > tol (:57), error = ||diag[pi_m:]||_1 / orig (:99);
      // torch.max propagates NaN and (NaN > tol) is False
      if (m > 0 && !(etot / orig > a.pc_tol)) early = true;
      if (wig == 0 && tp == 0) {
        a.err_rec[(size_t)m * a.B + b] = etot;
        if (m == 0) a.orig[b] = vb;
        a.swaps[(size_t)b * rank + m] = jb;
      }
      const float piv = sqrtf(vb);  // :73-74
      const float* gp = reinterpret_cast<const float*>(shp.gath[wb]) + FU_HDR;
      // pivot history for the root-form recurrence (M, after the loop): the pivot row of C and the pivot row's L entries
      // just arrived with the exchange (the data k_pb_rootform fetched from HBM afterwards)
      if (tp < RC + m) hist[m][tp < RC ? tp : tp - RC + 32] = gp[tp];
      if (tp == 63) hist[m][32 + m] = piv;
      float gc[RC];
#pragma unroll
      for (int i = 0; i < RC / 4; ++i) {
        const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * i);
        gc[4 * i] = g4.x; gc[4 * i + 1] = g4.y; gc[4 * i + 2] = g4.z; gc[4 * i + 3] = g4.w;
      }
      // Schur update of row m for the 4 rows of this thread IN LOCKSTEP: the products and sums of a row are a
      // dependent chain in the mandated order, the 4 rows give the instruction-level parallelism
      float rowv[R4_NR], accs[R4_NR];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) rowv[q] = gc[0] * Cr[q][0].x;
#pragma unroll
      for (int r = 1; r < RC; ++r)
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) rowv[q] = rowv[q] + gc[r] * ((r & 1) ? Cr[q][r >> 1].y : Cr[q][r >> 1].x);
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) accs[q] = 0.f;
      for (int j4 = 0; 4 * j4 < m; ++j4) {  // :83-89, sequential in j
        const float4 u4 = *reinterpret_cast<const float4*>(gp + RC + 4 * j4);
        const int j = 4 * j4;
        float4 l4[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) l4[q] = l_s[fu_l_slot(tp + R4_TPB * q, j4)];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) accs[q] = (j == 0) ? u4.x * l4[q].x : accs[q] + u4.x * l4[q].x;
        if (j + 1 < m) {
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) accs[q] = accs[q] + u4.y * l4[q].y;
        }
        if (j + 2 < m) {
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) accs[q] = accs[q] + u4.z * l4[q].z;
        }
        if (j + 3 < m) {
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) accs[q] = accs[q] + u4.w * l4[q].w;
        }
      }
      const int ms = m >> 2, me = m & 3;
      float vq[R4_NR];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) vq[q] = ((m > 0) ? rowv[q] - accs[q] : rowv[q]) / piv;  // :91
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int pq = pos[q];
        const bool live = pq != FU_INVALID;
        const int np = (pq == jb) ? m : ((pq == m) ? jb : pq);  // permutation swap of positions m and jb (:67-70)
        if (live) pos[q] = np;
        const float val = (np == m) ? piv : vq[q];  // the pivot row gets sqrt(max) (:73-74)
        if (live && np >= m) reinterpret_cast<float*>(&l_s[fu_l_slot(tp + R4_TPB * q, ms)])[me] = val;
        if (live && np > m) dg[q] = dg[q] - vq[q] * vq[q];  // :94-95
      }
    }
    if (early || !(orig == orig)) {
      if (wig == 0 && tp == 0) atomicExch(a.flags, 1);
    }
    __syncthreads();  // L rows and the exchange block are dead from here on (region A / B change hands)
    if (stamp) a.dbg[2] = wall_clock64();

    // ================================ 4. E: totals, and the root-form recurrence ================================
    {
      int te = t0;
      asm volatile("" : "+v"(te));
      const int wave_e = te >> 6, lane_e = te & 63;
      const unsigned tg = etg;
      constexpr int NE = 64 * 12;
      double tot[4];
      const int idx = te - 64;  // waves 1-3 fetch (192 threads x 4 entries); the first wave computes M meanwhile
      // the root-form recurrence (first wave, lane r < RC holds row r of M in registers; compile-time loops, the pivot
      // rows' L entries are wave-uniform LDS broadcasts) while the other waves start fetching
      //   M[r][j] = (C[pi_j][r] - sum_{i<j} M[r][i] L[pi_j][i]) / L[pi_j][j],   L = C M
      if (wave_e == 0) {
        // 1 / L[pi_j][j] for all j at once (lane j), then broadcast; the substitution is column-oriented: finishing
        // column j updates the 15 - j later columns with INDEPENDENT multiply-adds (no chain through one accumulator)
        const double rp = 1.0 / (double)hist[lane_e & 15][32 + (lane_e & 15)];
        double mr[FU_MAXRANK];
        const int r = lane_e < RC ? lane_e : 0;
#pragma unroll
        for (int j = 0; j < FU_MAXRANK; ++j) mr[j] = (j < rank) ? (double)hist[j][r] : 0.0;
#pragma unroll
        for (int j = 0; j < FU_MAXRANK; ++j) {
          if (j < rank) {
            mr[j] = mr[j] * fu_readlane_d(rp, j);
#pragma unroll
            for (int k2 = j + 1; k2 < FU_MAXRANK; ++k2)
              if (k2 < rank) mr[k2] = fma(-mr[j], (double)hist[k2][32 + j], mr[k2]);
          }
        }
        if (lane_e < RC) {
#pragma unroll
          for (int j = 0; j < FU_MAXRANK; ++j) Mm[lane_e][j] = mr[j];  // (columns >= rank are zero: G is padded with I)
        }
      }
      auto fetch_sum = [&](int e) -> double {
        double s = 0.0;
        FuWait wt{a.err, 0};
        for (int w0 = 0; w0 < GW; w0 += 8) {
          constexpr int W = GW < 8 ? GW : 8;
          unsigned long long x[W];
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              x[w] = __hip_atomic_load(eslot + (size_t)(w0 + w) * FU_ESLOT + e, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && ((unsigned)(x[w] >> 32) == tg);
            }
            if (ok || wt.give_up()) break;
          }
#pragma unroll
          for (int w = 0; w < W; ++w) s += (double)__uint_as_float((unsigned)(x[w] & 0xffffffffull));
        }
        return s;
      };
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int e = idx + 192 * s;
        const bool used = idx >= 0 && ((NB == 3) || ((e % 12) < 4));
        tot[s] = used ? fetch_sum(e) : 0.0;
      }
      double logd = 0.0;
      if (idx == 0) logd = fetch_sum(NE) + fetch_sum(NE + 1);
      if (stamp) a.dbg[10] = wall_clock64();
      double(*const Em)[FU_LD] = reinterpret_cast<double(*)[FU_LD]>(regA);
      const double sigma = d_full ? 1.0 : (double)a.d[b];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int e = (idx < 0 ? 0 : idx) + 192 * s;
        const int l2 = e / 12, bk = (e % 12) / 4, r = e % 4;
        const int gi = 4 * (l2 >> 4) + r + (bk == 2 ? 16 : 0), gj = (l2 & 15) + (bk >= 1 ? 16 : 0);
        const bool used = idx >= 0 && ((NB == 3) || (bk == 0));
        if (used && gi < RC && gj < RC) {
          Em[gi][gj] = tot[s] / sigma;
          if (bk == 1) Em[gj][gi] = tot[s] / sigma;  // E10 = E01^T
        }
      }
      if (idx == 0) {
        // logdet P = logdet(I + M^T E M) + sum log d  (added_diag_linear_operator.py:168-184); the first term is added
        // after the Cholesky factorisation below
        Em[RC][0] = d_full ? logd : (double)a.N * log(sigma);
      }
      __syncthreads();
    }
    if (stamp) a.dbg[3] = wall_clock64();


    // ================================ 5. the small algebra (k_pb_rootform) ================================
    // fp64; every inner loop has a compile-time trip count (M is padded to 16 columns with zeros, G with the identity),
    // the 16 x 16 Cholesky factorisation and the forward substitution run in registers of the first wave with
    // v_readlane broadcasts: the chains of dependent LDS round trips of a runtime-bounded version cost 27 us per member
    {
      double(*const Em)[FU_LD] = reinterpret_cast<double(*)[FU_LD]>(regA);
      double(*const Tm)[FU_LD] = reinterpret_cast<double(*)[FU_LD]>(regA + 1 * 33 * FU_LD * 8);
      double(*const Gm)[FU_LD] = reinterpret_cast<double(*)[FU_LD]>(regA + 2 * 33 * FU_LD * 8);
      double(*const Fm)[FU_LD] = reinterpret_cast<double(*)[FU_LD]>(regA + 3 * 33 * FU_LD * 8);
      constexpr int KP = FU_MAXRANK;
      constexpr int NT = R4_TPB;
      int ta = t0;
      asm volatile("" : "+v"(ta));
      const int wave_a = ta >> 6, lane_a = ta & 63;
      // The four matrix products of this phase run on v_mfma_f64_16x16x4_f64 (lane l supplies A[l % 16][l / 16] and
      // B[l / 16][l % 16] of a step, register r of the result is D[4 r + l / 16][l % 16]): one 16 x 16 tile per wave, the
      // operands one double per lane and step from LDS -- the vector-ALU version spent two 8-byte LDS reads on every
      // multiply-add (256 per thread for E F alone) and was bound by LDS bandwidth with two workgroups per CU.
      constexpr int NTL = (RC + 15) / 16;  // 16-row tiles of an RC x RC matrix
      const int mn = lane_a & 15, kq = lane_a >> 4;
      if (wave_a < NTL) {  // T = E M  [RC][16]
        const int ri = 16 * wave_a + mn;
        fu_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16 * NTL; k0 += 4) {
          const int kk = k0 + kq;
          const double av = (ri < RC && kk < RC) ? Em[ri][kk] : 0.0;
          const double bv = (kk < RC) ? Mm[kk][mn] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ro = 16 * wave_a + 4 * r + kq;
          if (ro < RC) Tm[ro][mn] = acc[r];
        }
      }
      __syncthreads();
      if (wave_a == 0) {  // G = I + M^T T  [16][16]
        fu_f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (4 * r + kq == mn) ? 1.0 : 0.0;
#pragma unroll
        for (int k0 = 0; k0 < 16 * NTL; k0 += 4) {
          const int kk = k0 + kq;
          const double av = (kk < RC) ? Mm[kk][mn] : 0.0;  // A[i = mn][k] = M[k][i]
          const double bv = (kk < RC) ? Tm[kk][mn] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Gm[4 * r + kq][mn] = acc[r];
      }
      __syncthreads();
      double ldt = 0.0;
      if (stamp) a.dbg[11] = wall_clock64();
      if (wave_a == 0) {
        // Lanes 0-15 hold the rows of G, lanes 16 .. 16 + RC - 1 the rows of M: the right-looking Cholesky
        // factorisation G = Lg Lg^T and the forward substitution Y = M Lg^-T are the SAME column operations (column j is
        // scaled by 1 / Lg[j][j], every later column c loses column j times Lg[c][j] -- lane c's entry, one broadcast
        // for both), so one loop does both.
        const int li = lane_a & 15;
        const int mr_ = lane_a - 16;
        const bool is_g = lane_a < 16, is_m = mr_ >= 0 && mr_ < RC;
        double gr[KP];
#pragma unroll
        for (int c2 = 0; c2 < KP; ++c2) gr[c2] = is_g ? Gm[li][c2] : (is_m ? Mm[mr_][c2] : 0.0);
        double myinv = 1.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          const double inv = fu_rsqrt(fu_readlane_d(gr[j], j));  // 1 / Lg[j][j]
          const double lj = gr[j] * inv;                           // Lg[i][j] (G rows) / Y[r][j] (M rows)
          gr[j] = lj;
          myinv = (lane_a == j) ? inv : myinv;
#pragma unroll
          for (int c2 = j + 1; c2 < KP; ++c2) gr[c2] = fma(-lj, fu_readlane_d(lj, c2), gr[c2]);
        }
        const double mylog = -log(myinv);  // log Lg[j][j] in lane j < 16
#pragma unroll
        for (int j = 0; j < KP; ++j) ldt += fu_readlane_d(mylog, j);
        if (is_m) {
#pragma unroll
          for (int j = 0; j < KP; ++j) Tm[mr_][j] = gr[j];
        }
      }
      __syncthreads();
      if (stamp) a.dbg[12] = wall_clock64();
      double(*const EFm)[FU_LD] = Gm;  // (G is dead after the factorisation: its block takes E F)
      if (wave_a < NTL * NTL) {  // F = Y Y^T  [RC][RC], tile (ti, tj) on wave ti NTL + tj
        const int ti = wave_a / NTL, tj = wave_a % NTL;
        const int ri = 16 * ti + mn, cj = 16 * tj + mn;
        fu_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < KP; k0 += 4) {
          const double av = (ri < RC) ? Tm[ri][k0 + kq] : 0.0;
          const double bv = (cj < RC) ? Tm[cj][k0 + kq] : 0.0;  // B[k][j] = Y[j][k]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ro = 16 * ti + 4 * r + kq;
          if (ro < RC && cj < RC) Fm[ro][cj] = acc[r];
        }
      }
      __syncthreads();
      if (wave_a < NTL * NTL) {  // E F  [RC][RC]
        const int ti = wave_a / NTL, tj = wave_a % NTL;
        const int ri = 16 * ti + mn, cj = 16 * tj + mn;
        fu_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16 * NTL; k0 += 4) {
          const int kk = k0 + kq;
          const double av = (ri < RC && kk < RC) ? Em[ri][kk] : 0.0;
          const double bv = (kk < RC && cj < RC) ? Fm[kk][cj] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ro = 16 * ti + 4 * r + kq;
          if (ro < RC && cj < RC) EFm[ro][cj] = acc[r];
        }
      }
      __syncthreads();
      float fv[(RC * RC + NT - 1) / NT], ev[(RC * RC + NT - 1) / NT], e0v[(RC * RC + NT - 1) / NT];
#pragma unroll
      for (int u = 0; u < (RC * RC + NT - 1) / NT; ++u) {
        const int pr = ta + NT * u;
        fv[u] = 0.f;
        ev[u] = 0.f;
        e0v[u] = 0.f;
        if (pr < RC * RC) {
          const int r = pr / RC, c2 = pr % RC;
          fv[u] = (float)Fm[r][c2];
          ev[u] = (float)EFm[r][c2];
          e0v[u] = (float)Em[r][c2];
          if (pr % GW == wig) {  // (every workgroup holds the same bits: each stores its share)
            if (a.F) a.F[(size_t)b * RC * RC + pr] = fv[u];
            if (a.EF) a.EF[(size_t)b * RC * RC + pr] = ev[u];
            if (a.E) a.E[(size_t)b * RC * RC + pr] = e0v[u];
          }
        }
      }
      if (wig == 0 && ta == 0) {
        // logdet P = 2 sum log diag(Lg) + sum log d  (added_diag_linear_operator.py:168-184)
        if (a.logdet_p) a.logdet_p[b] = (float)(2.0 * ldt + Em[RC][0]);
        if (a.dinv && !d_full) a.dinv[b] = 1.0f / a.d[b];
      }
      __syncthreads();  // the fp64 matrices are dead: region A becomes the CG vectors and F / EF
#pragma unroll
      for (int u = 0; u < (RC * RC + NT - 1) / NT; ++u) {
        const int pr = ta + NT * u;
        if (pr < RC * RC) {
          f_s[(pr / RC) * FLD + pr % RC] = fv[u];
          ef_s[(pr / RC) * FLD + pr % RC] = ev[u];
          if (WR) e_s[(pr / RC) * FLD + pr % RC] = e0v[u];
        }
      }
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = ta + R4_TPB * q;
        d_s[lr] = dq[q];
        dinv_s[lr] = dinvq[q];
      }
      __syncthreads();
    }
    if (stamp) a.dbg[4] = wall_clock64();

    // ================================ 6. CG (k_cg_onchip5) ================================
    int64_t b_next = a.B;
    {
    // (compiled with the contraction default of lo_cg_onchip4.hip -- only the load and pivot blocks switch it off --
    // so that this block produces the bits of k_cg_onchip5 given the same F / EF / 1/d: the first solve with an
    // operator -- this kernel -- and later ones -- the memoised root form -- agree bit for bit)
    int tg = t0;  // (opaque, as in the other phases)
    asm volatile("" : "+v"(tg));
    const int lane_c = tg & 63;
    const int nc = MC ? a.c : 1;
    const int clast = nc;
    int drawn = 0;
    auto draw = [&](int k, int col) {
      if (k == a.iters - 1 && col == clast - 1 && wig == 0 && tg == 0) drawn = atomicAdd(a.next_member, 1);
    };
    for (int col = 0; col < clast; ++col) {
      const size_t bc = (size_t)b * nc + col;
      float r[R4_NR], p[R4_NR];
      float sc[4];
      int tc = tg;
      asm volatile("" : "+v"(tc));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q)
        r[q] = a.rhs[((size_t)b * a.N + min(row0 + tc + R4_TPB * q, a.N - 1)) * nc + col];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = tc + R4_TPB * q;
        r[q] = (lr < nv) ? r[q] : 0.f;
        p[q] = 0.f;
        x_s[lr] = 0.f;
      }
      float nrm = 1.0f, inv0 = 1.0f;
      bool rhs_zero = false;
      float t_old = 0.f, tt_old = 0.f, dpp = 0.f, rz = 0.f, alpha = 0.f, beta = 0.f, rn = 0.f;
      float w_reg = 0.f;  // (WR: w_j in lane j of both halves, carried by the recurrence)
      bool conv = false;
      // one reduction: w = C^T (r / d), s1, s2, rp; then (every wave) the small algebra; k = -1 marks the initial one
      auto reduce_and_post = [&](int k) {
        sc[0] = 0.f; sc[1] = 0.f; sc[2] = 0.f;
        float rd[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) {
          const int lr = tg + R4_TPB * q;
          rd[q] = r[q] * dinv_s[lr];
          sc[0] = fmaf(r[q], r[q], sc[0]);
          sc[1] = fmaf(rd[q], r[q], sc[1]);
          sc[2] = fmaf(r[q], p[q], sc[2]);
        }
        const bool draws = (k == a.iters - 1) && (col == clast - 1);
        sc[3] = (draws && wig == 0 && tg == 0) ? (float)(ngroups + drawn) : 0.f;
        const bool wrec = WR && k >= 0;  // w comes from the recurrence: only the scalars are reduced
        float tot[4] = {0.f, 0.f, 0.f, 0.f};
        if (wrec) {
          if (draws) {
            r4_allreduce_small_t<GW, 4>(sh, sc, tot, g, tg);
            b_next = (int64_t)tot[3];
          } else {
            r4_allreduce_small_t<GW, 3>(sh, sc, tot, g, tg);
          }
        } else {
          fu_f32x2 wp[RC / 2];
#pragma unroll
          for (int j = 0; j < RC / 2; ++j) {
            fu_f32x2 v = Cr[0][j] * fu_f32x2{rd[0], rd[0]};
#pragma unroll
            for (int q = 1; q < R4_NR; ++q) v = __builtin_elementwise_fma(Cr[q][j], fu_f32x2{rd[q], rd[q]}, v);
            wp[j] = v;
          }
          const auto gen = [&](int c) { return (c & 1) ? wp[c >> 1].y : wp[c >> 1].x; };
          if (draws) {
            r4_allreduce_t<GW, RC>(sh, gen, sc, 4, g, tg);
            b_next = (int64_t)sh.res[RC + 3];
          } else {
            r4_allreduce_t<GW, RC>(sh, gen, sc, 3, g, tg);
          }
          tot[0] = sh.res[RC]; tot[1] = sh.res[RC + 1]; tot[2] = sh.res[RC + 2];
        }
        {
          const int j = lane_c & 31;
          float mv = 0.f;  // lanes 0-31: (F w)_j, lanes 32-63: (E F w)_j
          if (j < RC) {
            const float* row = (lane_c < 32 ? f_s : ef_s) + j * FLD;
            const float* wsrc = wrec ? post[tg >> 6].w : sh.res;
            fu_f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RC; q += 4) {
              const float4 m4 = *reinterpret_cast<const float4*>(row + q);
              const float4 w4 = *reinterpret_cast<const float4*>(&wsrc[q]);
              a01 = __builtin_elementwise_fma(fu_f32x2{m4.x, m4.y}, fu_f32x2{w4.x, w4.y}, a01);
              a23 = __builtin_elementwise_fma(fu_f32x2{m4.z, m4.w}, fu_f32x2{w4.z, w4.w}, a23);
            }
            mv = (a01.x + a01.y) + (a23.x + a23.y);
          }
          float sc1 = 1.0f, sc2 = 1.0f;
          if (k < 0) {
            nrm = sqrtf(tot[0]);                             // rhs.norm(2, dim=-2)          :177
            rhs_zero = nrm < a.eps;                          // :178
            if (rhs_zero) nrm = 1.0f;                        // :179
            inv0 = 1.0f / nrm;
            sc1 = inv0;
            sc2 = inv0 * inv0;
          }
          mv *= sc1;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mv), __float_as_uint(mv), false, false);
          const float vj = __uint_as_float(sw[0]), evj = __uint_as_float(sw[1]);
          const float wj = wrec ? w_reg : ((j < RC) ? sh.res[j] * sc1 : 0.f);
          const bool own = lane_c < 32 && j < RC;
          const float s1 = tot[0] * sc2, s2 = tot[1] * sc2, rp = tot[2];
          const float zj = wj - evj;                          // (C^T z)_j
          const bool live = j < RC;
          const float wv = lanes32_sum(live ? wj * vj : 0.f);
          const float vev = lanes32_sum(live ? vj * evj : 0.f);
          const float vt = lanes32_sum(live ? vj * t_old : 0.f);
          const float zz = lanes32_sum(live ? zj * zj : 0.f);
          const float zt = lanes32_sum(live ? zj * t_old : 0.f);
          const float rzn = s2 - wv;                         // residual_inner_prod :215 / :35-36
          float rnn = __builtin_amdgcn_sqrtf(s1);            // :298 / :204
          if (k >= 0) {                                      // closes iteration k: beta, residual norm, records
            beta = (rz < a.eps) ? 0.f : rzn * __builtin_amdgcn_rcpf(rz);  // :39-42
            if (rhs_zero) rnn = 0.f;                         // :299
            rn = rnn;
            if (wig == 0 && tg == 0) a.resid_rec[(size_t)k * a.B * nc + bc] = rn;
          } else {
            beta = 0.f;
            rn = rnn;
            if (wig == 0 && tg == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
          }
          conv = rn < a.stop_after;                          // :300
          rz = rzn;
          const float dzz = fmaf(-2.f, wv, s2) + vev;
          const float dzp = rp - vt;
          dpp = fmaf(beta, fmaf(beta, dpp, 2.f * dzp), dzz);
          tt_old = fmaf(beta, fmaf(beta, tt_old, 2.f * zt), zz);  // |C^T p_new|^2
          t_old = fmaf(beta, t_old, zj);                     // C^T p_new (lanes j < RC of both halves)
          const float pAp = tt_old + dpp;
          alpha = (pAp < a.eps) ? 0.f : rz * __builtin_amdgcn_rcpf(pAp);  // :254-257
          if (conv) alpha = 0.f;                             // :260
          FuPost& mine = post[tg >> 6];
          if (own) {
            mine.v[j] = vj;
            mine.t[j] = t_old;
          }
          if (WR) {  // w' = w - alpha (E t + t): see k_cg_onchip5 (lo_cg_onchip4.hip), operation for operation
            __builtin_amdgcn_wave_barrier();
            float et = 0.f;
            if (j < RC) {
              const int q0 = (lane_c < 32) ? 0 : RC / 2;
              const float* row = e_s + j * FLD + q0;
              fu_f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
              for (int q = 0; q < RC / 2; q += 4) {
                const float4 m4 = *reinterpret_cast<const float4*>(row + q);
                const float4 t4 = *reinterpret_cast<const float4*>(&mine.t[q0 + q]);
                a01 = __builtin_elementwise_fma(fu_f32x2{m4.x, m4.y}, fu_f32x2{t4.x, t4.y}, a01);
                a23 = __builtin_elementwise_fma(fu_f32x2{m4.z, m4.w}, fu_f32x2{t4.z, t4.w}, a23);
              }
              et = (a01.x + a01.y) + (a23.x + a23.y);
            }
            const auto se = __builtin_amdgcn_permlane32_swap(__float_as_uint(et), __float_as_uint(et), false, false);
            const float etj = __uint_as_float(se[0]) + __uint_as_float(se[1]);
            w_reg = fmaf(-alpha, etj + t_old, wj);
            if (own) mine.w[j] = w_reg;
          }
        }
      };
      draw(-1, col);
      reduce_and_post(-1);
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) r[q] = r[q] / nrm;     // :182 (the reduction above saw the raw column)
      for (int k = 0; k < a.iters; ++k) {
        draw(k, col);
        const float al = alpha, be = beta;
        const FuPost& mine = post[tg >> 6];
        fu_f32x2 cv2[R4_NR], y2[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) { cv2[q] = fu_f32x2{0.f, 0.f}; y2[q] = fu_f32x2{0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < RC; i += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(&mine.v[i]);
          const float4 t4 = *reinterpret_cast<const float4*>(&mine.t[i]);
          const fu_f32x2 va{v4.x, v4.y}, vb{v4.z, v4.w}, ta{t4.x, t4.y}, tb{t4.z, t4.w};
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) {
            cv2[q] = __builtin_elementwise_fma(Cr[q][i / 2], va, cv2[q]);
            cv2[q] = __builtin_elementwise_fma(Cr[q][i / 2 + 1], vb, cv2[q]);
            y2[q] = __builtin_elementwise_fma(Cr[q][i / 2], ta, y2[q]);
            y2[q] = __builtin_elementwise_fma(Cr[q][i / 2 + 1], tb, y2[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) {
          const int lr = tg + R4_TPB * q;
          const float cv = cv2[q].x + cv2[q].y, y = y2[q].x + y2[q].y;
          p[q] = fmaf(be, p[q], (r[q] - cv) * dinv_s[lr]);   // p = beta p + (r - C v) / d  (:268, :46)
          x_s[lr] = fmaf(al, p[q], x_s[lr]);                 // x += alpha p (:31)
          r[q] = fmaf(-al, fmaf(d_s[lr], p[q], y), r[q]);    // r -= alpha (C t + d p) (:264)
        }
        reduce_and_post(k);
      }
      // ---- only the solution leaves the chip: result * rhs_norm (:335) ----
      int tw = tg;
      asm volatile("" : "+v"(tw));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        if (tw + R4_TPB * q < nv)
          a.xout[((size_t)b * a.N + row0 + tw + R4_TPB * q) * nc + col] = x_s[tg + R4_TPB * q] * nrm;
      }
      __syncthreads();
    }  // columns
    }
    if (stamp) a.dbg[5] = wall_clock64();
    b = b_next;
  }
}

template <int RC, int GW, bool MC>
static int fused_go(const FusedArgs& a, int nwg, hipStream_t st) {
  int per_cu = 0;
  if (LO_OCCUPANCY_CACHED(per_cu, (k_solve_fused<RC, GW, MC>), R4_TPB, 0) != hipSuccess ||
      per_cu < 2 || (2 * nwg / 8) / GW < 1)
    return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("solve_fused", st);
  {
    ResidentLaunch guard(st);
    hipLaunchKernelGGL((k_solve_fused<RC, GW, MC>), dim3(2 * nwg), dim3(R4_TPB), 0, st, a);
  }
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int RC>
static int fused_launch_rc(const FusedArgs& a, int nwg, hipStream_t st) {
  const bool mc = a.c > 1;
#define LO_FU_G(G_) (mc ? fused_go<RC, G_, true>(a, nwg, st) : fused_go<RC, G_, false>(a, nwg, st))
  switch (a.GW) {
    case 1: return LO_FU_G(1);
    case 2: return LO_FU_G(2);
    case 4: return LO_FU_G(4);
    case 8: return LO_FU_G(8);
    case 16: return LO_FU_G(16);
    // (round 4: groups of 16 -- N <= 16384 -- compile with 4 cold spilled registers since the single-column CG carries w
    //  by recurrence; groups of 32 still spill in the pivot loop: those members take the three-launch path)
    default: return LO_ERR_UNSUPPORTED;
  }
#undef LO_FU_G
}

int fused_launch_r8(const FusedArgs& a, int nwg, hipStream_t st);
int fused_launch_r16(const FusedArgs& a, int nwg, hipStream_t st);
int fused_launch_r32(const FusedArgs& a, int nwg, hipStream_t st);

}  // namespace lo

// lo_rspace.hip -- the preconditioned CG of a low-rank + diagonal member in the coordinates of its own Krylov space.
//
// For A = C C^T + D (C [N, R], R <= 32) and the pivoted-Cholesky preconditioner in root form (lo_amd.h: lo_precond_desc.F),
//     P^-1 = D^-1 - D^-1 C F C^T D^-1,        A P^-1 = I + C G C^T D^-1,   G = I - F - E F,   E = C^T D^-1 C,
// every vector linear_cg (linear_operator/utils/linear_cg.py:245-332) ever forms is a combination of the right-hand side
// r0 and the R columns of C:
//     r_k = rho r0 + C g,        p_k = D^-1 (pi r0 + C h),        x_k = D^-1 (xi r0 + C y)          (g, h, y in R^R)
// and every inner product of the iteration follows from ONE reduction over the rows per solve
//     w0 = C^T D^-1 r0,   u0 = C^T r0,   s = r0^T D^-1 r0,   a0 = r0^T r0
// and two Gram matrices of the OPERATOR, E and G2 = C^T C (built once with the preconditioner: lo_precond_desc.RS):
//     w  = C^T D^-1 r = rho w0 + E g          r^T D^-1 r = rho (rho s + g.w0) + g.w        r^T r = rho (rho a0 + 2 g.u0) + g.G2 g
//     z  = P^-1 r = D^-1 (rho r0 + C (g - F w))                     r.z = r^T D^-1 r - w.F w        (:215, :35-36)
//     t  = C^T p = pi w0 + E h,   A p = pi r0 + C (h + t)            p.A p = |t|^2 + (pi^2 s + 2 pi h.w0 + h.E h)   (:247-257)
// The iterations of the reference's floor run on R + 1 coordinates in fp64 inside ONE wave per workgroup (the same
// recurrences as k_cg_onchip5, lo_cg_onchip4.hip: beta, the p-recurrences, alpha -- identical in exact arithmetic);
// the rows of C are touched twice per solve: for the reduction above and for x = D^-1 (xi r0 + C y).  A member costs one
// group all-reduce instead of twelve.  Numerics first: tests/proto/proto_rspace.py (solutions 1e-7 from the fp64 iteration
// where the fp32 iteration of the reference is at 3e-6; needs the Gram matrices in fp64 -- with fp32-level noise in E the
// iteration stalls when r0 lies almost inside span(C)).  The mode serves the result-only first pass of a single-column
// solve (as the w-recurrence mode did): a solve whose residual misses the stop rule at the floor is repeated by the
// three-pass kernel with the continuation state.
//
// Second half of round 5 (template parameter DG of k_cg_rspace, lo_precond_desc.RSD, lo_eigform.hip): the same iteration in
// the basis that diagonalises the preconditioned member on span(C) -- the CG of a diagonal matrix, one reduction of three
// values per iteration, no R x R product and no barrier on the chain; residual norms (in the coordinates of C) off the
// chain; right-hand sides (almost) inside span(C) ask for the dense form above (CgCtrl::rs_redo).  DESIGN.md 4.14.
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"
#include "lo_group_reduce.h"
#include "lo_cg_close.h"
#include "lo_f64_lanes.h"

namespace lo {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// fp64 Gram matrices of a root on the fp64 matrix cores: E = C^T diag(dinv) C and G2 = C^T C, partials per row slice.
// Products of two fp32 values are exact in fp64, so the only rounding is the fp64 accumulation (and c * dinv, 2^-53).
// Layout of v_mfma_f64_16x16x4_f64 (tools/probe/mfma_f64_layout.hip): lane (a = l % 16, kk = l / 16) supplies
// A[a][kk] and B[kk][a], acc[r] = D[4 r + kk][a].
// ---------------------------------------------------------------------------------------------------------------------
template <int NA>
__global__ __launch_bounds__(kThreads) void k_rs_gram64(const float* __restrict__ C, const float* __restrict__ dinv,
                                                         int N, int R, int rows_per, double* __restrict__ gpartE,
                                                         double* __restrict__ gpart2) {
  constexpr int NB = (NA == 2) ? 3 : 1;
  constexpr int TR = 128;  // rows per tile: 32 per wave
  constexpr int LD = 33;   // tile row stride (conflict-free column reads)
  __shared__ float tile[2][TR * LD];
  __shared__ float dtile[2][TR];
  __shared__ double red[4][64][4 * NB];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int a = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float* Cb = C + (size_t)b * N * R;
  const float* db = dinv ? dinv + (size_t)b * N : nullptr;
  const int RQ = R >> 2;
  constexpr int PPT = TR * 8 / kThreads;
  float4 pv[PPT];
  float pd = 0.f;
  auto issue = [&](int base) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = i * kThreads + threadIdx.x;
      const int row = e / RQ, q = e % RQ;
      pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < TR * RQ && base + row < r1) pv[i] = *reinterpret_cast<const float4*>(Cb + (size_t)(base + row) * R + 4 * q);
    }
    pd = 0.f;
    if (db && threadIdx.x < TR && base + (int)threadIdx.x < r1) pd = db[base + threadIdx.x];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = i * kThreads + threadIdx.x;
      if (e < TR * RQ) {
        const int row = e / RQ, q = e % RQ;
        float* t = &tile[buf][row * LD + 4 * q];
        t[0] = pv[i].x; t[1] = pv[i].y; t[2] = pv[i].z; t[3] = pv[i].w;
      }
    }
    if (threadIdx.x < TR) dtile[buf][threadIdx.x] = pd;
  };
  f64x4 e00 = {0.0, 0.0, 0.0, 0.0}, e01 = e00, e11 = e00, g00 = e00, g01 = e00, g11 = e00;
  int buf = 0;
  issue(r0);
  commit(0);
  __syncthreads();
  for (int base = r0; base < r1; base += TR) {
    const bool more = base + TR < r1;
    if (more) issue(base + TR);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 32 * wave + 4 * e + kk;
      const double w0 = (a < R) ? (double)tile[buf][row * LD + a] : 0.0;
      g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, w0, g00, 0, 0, 0);
      double w1 = 0.0;
      if (NA == 2) {
        w1 = (a + 16 < R) ? (double)tile[buf][row * LD + a + 16] : 0.0;
        g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, w1, g01, 0, 0, 0);
        g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, w1, g11, 0, 0, 0);
      }
      if (db) {
        const double dv = (double)dtile[buf][row];
        const double v0 = w0 * dv;
        e00 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, w0, e00, 0, 0, 0);
        if (NA == 2) {
          const double v1 = w1 * dv;
          e01 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, w1, e01, 0, 0, 0);
          e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, w1, e11, 0, 0, 0);
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  for (int which = 0; which < (db ? 2 : 1); ++which) {
    const f64x4& q00 = which ? e00 : g00;
    const f64x4& q01 = which ? e01 : g01;
    const f64x4& q11 = which ? e11 : g11;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave][l][r] = q00[r];
      if (NA == 2) {
        red[wave][l][4 + r] = q01[r];
        red[wave][l][8 + r] = q11[r];
      }
    }
    __syncthreads();
    if (wave == 0) {
      double* gp = (which ? gpartE : gpart2) + ((size_t)b * S + s) * R * R;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = 4 * r + kk, gj = a;  // D[4 r + l / 16][l % 16]
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          const double v = (red[0][l][4 * blk + r] + red[1][l][4 * blk + r]) + (red[2][l][4 * blk + r] + red[3][l][4 * blk + r]);
          const int i = gi + (blk == 2 ? 16 : 0), j = gj + (blk >= 1 ? 16 : 0);
          if (i < R && j < R) {
            gp[i * R + j] = v;
            if (blk == 1) gp[j * R + i] = v;  // the (1, 0) block is the transpose of (0, 1)
          }
        }
      }
    }
  }
}

void rs_gram64_launch(const float* C, const float* dinv_full, int64_t B, int64_t N, int R, Split sp, double* gpartE,
                      double* gpart2, hipStream_t st) {
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  if (R <= 16) hipLaunchKernelGGL((k_rs_gram64<1>), grid, block, 0, st, C, dinv_full, (int)N, R, sp.rows, gpartE, gpart2);
  else hipLaunchKernelGGL((k_rs_gram64<2>), grid, block, 0, st, C, dinv_full, (int)N, R, sp.rows, gpartE, gpart2);
}

// ---------------------------------------------------------------------------------------------------------------------
// The kernel.  Same launch geometry, member hand-out and granule hand-off as k_cg_onchip5 (lo_cg_onchip4.hip): 256
// threads x 4 rows, the rows of C in VGPRs, groups of GW workgroups per member, two workgroups per CU.
//   payload of the member's ONE all-reduce (fp64, two tagged 8-byte granules per value):
//     [per 16-column block: w0 | u0]  s  a0  next-member
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RS_SLOT = 136;  // granules per workgroup and parity: 2 x (2 * 32 + 3), padded
__host__ __device__ constexpr int rs_np(int RC) { return 2 * RC + 3; }
__host__ __device__ constexpr size_t rs_group_granules(int GW) { return (size_t)2 * (GW + 1) * R4_SLOT + (size_t)2 * GW * RS_SLOT; }

template <int RC>
struct alignas(16) RsShared {
  double red[R4_WAVES][2 * RC + 4];
  double res[2 * RC + 4];
  double gv[R4_WAVES][32];      // every wave's own broadcast copy of the vector it multiplies (g, w0, at the end nrm * y)
  double out[2][R4_WAVES][32];  // the four products of an iteration, double-buffered by the iteration's parity
};

template <int RC, int GW, bool DG>
__global__ __launch_bounds__(R4_TPB, 2 * R4_TPB / 256) void k_cg_rspace(OnchipArgs a) {
  constexpr int NP = rs_np(RC) + (DG ? 1 : 0);  // (diagonal form: + sum dinv^2, the lower bound of the residual norm)
  constexpr int MLD = RC + 2;            // row stride of the fp64 matrices in LDS (16-byte aligned, conflict-free b128 reads)
  constexpr int H = 8;                   // columns per accumulation round (2 H fp64 accumulators per thread)
  constexpr int NH = RC / 2;             // columns per half-wave in the R x R products
  __shared__ R4Shared sh;
  __shared__ RsShared<RC> rs;
  __shared__ __attribute__((aligned(16))) double mat_s[4 * RC * MLD];  // E | F E | E F E | G2
  __shared__ float4 stage_s[R4_WAVES * 64 * (RC / 4)];                 // per-wave transposition window of the member load
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  R4Group g;
  // granules of the group: [placement check: 2 x (GW + 1) x R4_SLOT] [the members' all-reduces: 2 x GW x RS_SLOT]
  g.gslot = a.gbuf + (size_t)grp * rs_group_granules(GW);
  unsigned long long* const rs_slot = g.gslot + 2 * (GW + 1) * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  g.tag = 0;
  g.err = a.err;
  g.same_xcd = false;
  {  // placement check (see k_cg_onchip4): plain-store hand-off only when the whole group shares an XCD
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t < 64) {
      sh.red[0][0] = (float)xcc;
      sh.red[0][1] = (float)(xcc * xcc);
    }
    if (t < 2 * (R4_WAVES - 1)) sh.red[1 + t / 2][t % 2] = 0.f;
    r4_group_sum<GW>(sh, 2, g);
    const float fx = (float)xcc;
    g.same_xcd = (sh.res[0] == GW * fx) && (sh.res[1] == GW * fx * fx) && (a.allow_l2_handoff != 0);
    __syncthreads();
  }
  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));
  const bool di_full = a.dinv_mode == LO_DIAG_FULL;
  int64_t b = grp;
  while (b < a.B) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    int drawn = 0;
    if (wig == 0 && t == 0) drawn = atomicAdd(a.next_member, 1);  // (its round trip hides behind the member load)
    f32x2 Cr[R4_NR][RC / 2];
    int tl = t;
    asm volatile("" : "+v"(tl));
    // ---- every load of the member is issued before anything waits (vmcnt counts in order) ----
    constexpr int CH = RC / 4;
    const int wv_ = tl >> 6, ln = tl & 63;
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int gch = 64 * i + ln;  // chunk of the wave's block
        const int rw = gch / CH, ck = gch % CH;
        const size_t grow_c = (size_t)b * a.N + min(row0 + R4_TPB * q + 64 * wv_ + rw, a.N - 1);
        const float4 c4 = *reinterpret_cast<const float4*>(a.C + grow_c * RC + 4 * ck);
        Cr[q][2 * i] = f32x2{c4.x, c4.y}; Cr[q][2 * i + 1] = f32x2{c4.z, c4.w};
      }
    }
    float bq[R4_NR], diq[R4_NR];
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const size_t grow = (size_t)b * a.N + min(row0 + tl + R4_TPB * q, a.N - 1);
      bq[q] = a.rhs[grow];
      diq[q] = di_full ? a.dinv[grow] : a.dinv[b];
    }
    constexpr int NM = (4 * RC * RC / 2 + R4_TPB - 1) / R4_TPB;  // 16-byte pieces of the four matrices per thread
    f32x4 mv[NM];
    {
      // E | F E | E F E | G2 (| F | E F), or the diagonal form TinT | E^+ | TuT | G2 (| Tin | lam)
      const f32x4* src = reinterpret_cast<const f32x4*>((DG ? a.RSD : a.RS) + (size_t)b * 6 * RC * RC);
#pragma unroll
      for (int u = 0; u < NM; ++u) mv[u] = src[min(tl + R4_TPB * u, 4 * RC * RC / 2 - 1)];
    }
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const bool valid = tl + R4_TPB * q < nv;
      float4* win = stage_s + wv_ * (64 * CH);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int gch = 64 * i + ln;
        const int rw = gch / CH, ck = gch % CH;
        win[rw * CH + (ck ^ ((rw ^ (rw >> 3)) & (CH - 1)))] =
            make_float4(Cr[q][2 * i].x, Cr[q][2 * i].y, Cr[q][2 * i + 1].x, Cr[q][2 * i + 1].y);
      }
      __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order; this pins the compiler's order)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float4 c4 = win[ln * CH + (i ^ ((ln ^ (ln >> 3)) & (CH - 1)))];
        Cr[q][2 * i] = f32x2{c4.x, c4.y}; Cr[q][2 * i + 1] = f32x2{c4.z, c4.w};
      }
#pragma unroll
      for (int i = 0; i < RC / 2; ++i) Cr[q][i] = valid ? Cr[q][i] : f32x2{0.f, 0.f};
      __builtin_amdgcn_wave_barrier();  // the next row set reuses the window
      bq[q] = valid ? bq[q] : 0.f;
      diq[q] = valid ? diq[q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      const int e = tl + R4_TPB * u;  // piece e = (matrix m, row i, column pair jj)
      if (e < 4 * RC * RC / 2) {
        const int mi = e / (RC / 2), jj = e % (RC / 2);
        *reinterpret_cast<f32x4*>(&mat_s[mi * MLD + 2 * jj]) = mv[u];
      }
    }
    if (stamp) a.dbg[1] = wall_clock64();

    // ---- the member's one reduction over the rows, fp64: w0 = C^T (dinv o b), u0 = C^T b per 16-column block, s, a0 ----
    {
      double bd[R4_NR], bb[R4_NR];
      double s_acc = 0.0, a_acc = 0.0, d2_acc = 0.0;
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        bb[q] = (double)bq[q];
        bd[q] = bb[q] * (double)diq[q];
        s_acc = fma(bb[q], bd[q], s_acc);
        a_acc = fma(bb[q], bb[q], a_acc);
        if constexpr (DG) d2_acc = fma((double)diq[q], (double)diq[q], d2_acc);
      }
#pragma unroll
      for (int blk = 0; blk < RC / H; ++blk) {
        double acc[2 * H];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          double aw = 0.0, au = 0.0;
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) {
            const int col = H * blk + j;
            const double cj = (double)((col & 1) ? Cr[q][col >> 1].y : Cr[q][col >> 1].x);
            aw = fma(cj, bd[q], aw);
            au = fma(cj, bb[q], au);
          }
          acc[j] = aw;
          acc[H + j] = au;
        }
        const double mine = wave_rs_d<2 * H>(acc, lane);
        constexpr int per = 64 / (2 * H);  // lanes per component
        if ((lane & (per - 1)) == 0) rs.red[wave][2 * H * blk + lane / per] = mine;
        __builtin_amdgcn_sched_barrier(0);  // (one block of columns at a time: 2 H fp64 accumulators live)
      }
      const double ssum = wave_sum_fast_d(s_acc), asum = wave_sum_fast_d(a_acc);
      double d2sum = 0.0;
      if constexpr (DG) d2sum = wave_sum_fast_d(d2_acc);
      if (lane == 0) {
        rs.red[wave][2 * RC] = ssum;
        rs.red[wave][2 * RC + 1] = asum;
        rs.red[wave][2 * RC + 2] = (wig == 0 && wave == 0) ? (double)(ngroups + drawn) : 0.0;
        if constexpr (DG) rs.red[wave][2 * RC + 3] = d2sum;
      }
    }
    // (opaque to value numbering: otherwise the fp64 conversions of the rows are kept -- and spilled -- for the last pass)
#pragma unroll
    for (int q = 0; q < R4_NR; ++q)
#pragma unroll
      for (int i = 0; i < RC / 2; ++i) asm volatile("" : "+v"(Cr[q][i]));
    // rows of F (wave 1) and E F (wave 2) for the two products with w0 behind the all-reduce: in flight during the exchange
    constexpr int NFR = DG ? 1 : NH / 2;
    f32x4 frow[NFR];
    double lamj = 1.0;
    __builtin_amdgcn_sched_barrier(0);  // (not earlier: the fp64 accumulators of the reduction need the registers)
    if constexpr (!DG) {
      const int jj = min(lane & 31, RC - 1), hh = lane >> 5;
      const int wsel = (wave == 2) ? 5 : 4;
      const f32x4* fsrc = reinterpret_cast<const f32x4*>(a.RS + ((size_t)b * 6 + wsel) * RC * RC + (size_t)jj * RC + hh * NH);
#pragma unroll
      for (int q = 0; q < NH / 2; ++q) frow[q] = (wave == 1 || wave == 2) ? fsrc[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      // this lane's eigenvalue: in flight during the exchange
      lamj = a.RSD[((size_t)b * 6 + 5) * RC * RC + min(lane & 31, RC - 1)];
      frow[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- group all-reduce of NP doubles: thread tt < NP owns component tt (two granules: low | high word) ----
    if (a.prefetch & 4) __builtin_amdgcn_s_setprio(3);
    {
      const unsigned tag = ++g.tag;
      __syncthreads();
      if (t < NP) {
        const double v = (rs.red[0][t] + rs.red[1][t]) + (rs.red[2][t] + rs.red[3][t]);
        double tot = v;
        if constexpr (GW > 1) {
          unsigned long long* slot = rs_slot + (size_t)(tag & 1u) * GW * RS_SLOT;
          unsigned long long* mine = slot + (size_t)wig * RS_SLOT + 2 * t;
          const unsigned long long g0 = ((unsigned long long)tag << 32) | (unsigned long long)lo_w(v);
          const unsigned long long g1 = ((unsigned long long)tag << 32) | (unsigned long long)hi_w(v);
          if (g.same_xcd) {
            __hip_atomic_store(mine, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(mine + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            __hip_atomic_store(mine, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          tot = 0.0;
          unsigned spin = 0;
          constexpr int CK = (GW < 8) ? GW : 8;  // workgroups polled together (2 CK loads in flight)
          bool lost = false;
          for (int w0_ = 0; w0_ < GW && !lost; w0_ += CK) {
            unsigned long long x0[CK], x1[CK];
            for (;;) {
              bool ok = true;
#pragma unroll
              for (int w = 0; w < CK; ++w) {
                const unsigned long long* src = slot + (size_t)(w0_ + w) * RS_SLOT + 2 * t;
                x0[w] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x1[w] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && ((unsigned)(x0[w] >> 32) == tag) && ((unsigned)(x1[w] >> 32) == tag);
              }
              if (ok) break;
              if (++spin > R4_MAXSPIN ||
                  ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                atomicExch(g.err, 1);  // timed out, or another workgroup already did: give up at once
                lost = true;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int w = 0; w < CK; ++w) tot += mk_d((unsigned)(x0[w] & 0xffffffffull), (unsigned)(x1[w] & 0xffffffffull));
          }
        }
        rs.res[t] = tot;
      }
      __syncthreads();
    }
    if (stamp) a.dbg[2] = wall_clock64();
    const int64_t b_next = (int64_t)rs.res[2 * RC + 2];
    // The iterations are ONE dependent chain per wave (~200 instructions each, 0.6 us when nothing competes) that shares
    // its SIMD with a wave of the CU's other workgroup; when that one is in its instruction-dense reduction the chain ran
    // up to twice as long (8.3 .. 15.8 us per member).  Raised priority lets the chain issue whenever it is ready, the
    // dense phase of the other workgroup fills the gaps.
    if (a.prefetch & 1) __builtin_amdgcn_s_setprio(3);

    // ---- the iterations, on R + 1 coordinates.  All four waves carry the (replicated) state; wave m owns matrix m of
    // E | F E | E F E | G2 and contributes its product with g, one barrier per iteration joins them:
    //     w = rho w0 + E g,    v = F w = rho (F w0) + (F E) g,    E v = rho (E F w0) + (E F E) g
    // (F w0 and E F w0 once per member, by the waves 1 and 2 from rows of F / E F they fetched before the all-reduce) ----
    const int j = lane & 31, hf = lane >> 5;
    const bool live = j < RC;
    const int jr = live ? j : RC - 1;
    double yj = 0.0, xi = 0.0;
    float nrm;
    // (register budget: every wave parks its last row set of C in its own transposition window while it iterates)
    {
      float4* win = stage_s + wave * (64 * (RC / 4));
#pragma unroll
      for (int i = 0; i < RC / 4; ++i)
        win[i * 64 + lane] = make_float4(Cr[R4_NR - 1][2 * i].x, Cr[R4_NR - 1][2 * i].y, Cr[R4_NR - 1][2 * i + 1].x,
                                         Cr[R4_NR - 1][2 * i + 1].y);
    }
    if constexpr (DG) {
      // ---- diagonal form (lo_eigform.hip): every wave runs the whole chain on its own copy, no barrier inside ----
      const int iw = (jr / H) * 2 * H + (jr % H);
      double a0 = rs.res[2 * RC + 1], s = rs.res[2 * RC];
      const double d2 = rs.res[2 * RC + 3];
      nrm = sqrtf((float)a0);                             // rhs.norm(2, dim=-2)          :177
      const bool rhs_zero = nrm < a.eps;                  // :178
      if (rhs_zero) nrm = 1.0f;                           // :179
      const double inv = 1.0 / (double)nrm;
      const double w0 = live ? rs.res[iw] * inv : 0.0, u0 = live ? rs.res[iw + H] * inv : 0.0;
      s *= inv * inv;
      a0 *= inv * inv;
      // r^T r >= r.z / max(dinv) >= r.z / sqrt(sum dinv^2): above this r.z the has_converged mask (:300, 1e-10) cannot hold
      const double sure_rz = 2.0 * (double)a.stop_after * (double)a.stop_after * sqrt(d2);
      auto own_row_dot = [&](const double* mrow, const double* vec) {
        double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
        for (int q = 0; q < NH; q += 2) {
          const double2 xa = *reinterpret_cast<const double2*>(mrow + q);
          const double2 xx = *reinterpret_cast<const double2*>(vec + q);
          a0_ = fma(xa.x, xx.x, a0_);
          a1_ = fma(xa.y, xx.y, a1_);
        }
        double lo_, up_;
        halves_d(a0_ + a1_, lo_, up_);
        return lo_ + up_;
      };
      double* myv = rs.gv[wave];
      if (hf == 0) myv[j] = w0;
      __builtin_amdgcn_wave_barrier();
      const double* row0 = mat_s + (size_t)jr * MLD + hf * NH;
      // c0 = TinT w0 = W^T beta0, g0 = TuT w0 = W^-1 beta0 (beta0 = U^T b^): |beta0|^2 = c0 . g0 and V S^-1 beta0 = Tin g0 --
      // the projection on span(C) through the two well-conditioned transforms (E^+ = V S^-2 V^T would square cond(S))
      double c0 = own_row_dot(row0, myv + hf * NH);                              // TinT w0
      double e0 = own_row_dot(row0 + (size_t)2 * RC * MLD, myv + hf * NH);       // g0 = TuT w0
      const double* nrow = row0 + (size_t)3 * RC * MLD;                          // row j of G2 = C^T C
      const double* tucol = mat_s + (size_t)(2 * RC + hf * NH) * MLD + jr;       // column j of TuT (this half of its rows)
      c0 = live ? c0 : 0.0;
      e0 = live ? e0 : 0.0;
      double tau2 = s - lanes32_sum_d(c0 * e0);            // |b_perp|^2 = s - |beta0|^2
      tau2 = tau2 > 0.0 ? tau2 : 0.0;
      // A right-hand side (almost) inside span(C): the complement's coordinate c' then carries a weight of tau2 / s in every
      // inner product the iteration sees, yet stands for c' r0 in the solution -- the floor's 11 iterations can leave it
      // unresolved without the recurrences noticing (tools/fuzz_eigform.py: one member in 10^4 with tau2 / s = 7.5e-9
      // returned a solution whose true residual was 0.1 at a reported 1.2e-4, depending on the last bit of E).  The dense
      // form keeps r0's coefficient as a coordinate of its own; such members ask for it (CgCtrl::rs_redo).
      const bool delicate = !rhs_zero && tau2 < 1e-2 * s;
      const double lam = live ? lamj : 1.0;
      long long ts_a = 0, ts_b = 0;
      if (stamp) ts_a = wall_clock64();
      double cj = c0, qj = 0.0, etaj = 0.0;
      double cp = 1.0, qp = 0.0, etap = 0.0;
      double rz = 0.0, pApE = 0.0, alpha = 0.0, beta = 0.0;
      float rn = 0.f, last_alpha = 0.f;
      bool conv = false;
      unsigned close_flags = 0u;
      const size_t bc = (size_t)b;
      const bool rec = wig == 0 && lane == 0;
      const int last_owner = (a.iters - 1) & 3;  // the wave that forms the last residual norm writes the member's state
      for (int k = -1; k < a.iters; ++k) {
        if (k >= 0) {  // x += alpha p (:31);  r -= alpha A p (:264)
          last_alpha = (float)alpha;
          etaj = fma(alpha, qj, etaj);
          etap = fma(alpha, qp, etap);
          cj = fma(-alpha * lam, qj, cj);
          cp = fma(-alpha, qp, cp);
        }
        const double lc = lam * cj;
        // three sums over the components in one pass: the lower half-wave carries c.c and c.lam c, the upper c.lam q
        const bool lo_h = hf == 0;
        const float inv_rz = __builtin_amdgcn_rcpf((float)rz);       // (off the chain: rz is the previous iteration's)
        const double red2 = row16_sum_d(halve_pair_d<16>(lo_h ? cj * cj : lc * qj, lo_h ? lc * cj : 0.0, lane));
        const auto pick = [&](double v, int ln) {
          return mk_d((unsigned)__builtin_amdgcn_readlane((int)lo_w(v), ln),
                      (unsigned)__builtin_amdgcn_readlane((int)hi_w(v), ln));
        };
        const double cc = pick(red2, 0), clc = pick(red2, 16), clq = pick(red2, 32);
        const double rzn = fma(cp * cp, tau2, cc);                     // residual_inner_prod :215 / :35-36
        // residual norm (:298 / :204): needed by the stop rule's records and by the has_converged mask
        // (neither closing step reads the records of the iterations in between -- cg_close_solve takes the last norm from the
        //  member's granule, k_cg_ctrl_onchip reads the records of k = 0 and k = iters - 1 -- so the norm is formed for the
        //  last iteration by one wave, for the first one when the host closes, and whenever the mask could hold)
        const bool sure = rzn > sure_rz;
        const bool rec_first = k == 0 && a.close_gran == nullptr && wave == 0;
        const bool rec_last = k == a.iters - 1 && wave == last_owner;
        const bool own = rec_first || rec_last || !sure;
        float rnn = 0.f;
        if (k < 0) {
          const float s1f = (float)a0;
          rnn = __builtin_amdgcn_sqrtf(s1f < 0.f ? 0.f : s1f);
        } else if (own) {
          // in the coordinates of C, as the dense form: r = c' r0 + C g with g = Tu (c - c' c0) (column walk of TuT), then
          // r^T r = c'^2 a0 + 2 c' g.u0 + g.G2 g -- through the orthonormal basis the rounding of G2 would be amplified by cond(E)
          const double dl = fma(-cp, c0, cj);                          // del = c - c' c0
          __builtin_amdgcn_wave_barrier();
          if (hf == 0) myv[j] = dl;
          __builtin_amdgcn_wave_barrier();
          double gj_;
          {
            double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
            for (int q = 0; q < NH; q += 2) {
              const double2 dd = *reinterpret_cast<const double2*>(myv + hf * NH + q);
              a0_ = fma(tucol[(size_t)q * MLD], dd.x, a0_);
              a1_ = fma(tucol[(size_t)(q + 1) * MLD], dd.y, a1_);
            }
            double lo_, up_;
            halves_d(a0_ + a1_, lo_, up_);
            gj_ = live ? lo_ + up_ : 0.0;
          }
          __builtin_amdgcn_wave_barrier();
          if (hf == 0) myv[j] = gj_;
          __builtin_amdgcn_wave_barrier();
          const double nd = own_row_dot(nrow, myv + hf * NH);          // (G2 g)_j
          const double r2 = row16_sum_d(halve_pair_d<16>(gj_ * nd, gj_ * u0, lane));
          const double dnd = pick(r2, 0), dm = pick(r2, 16);
          const double s1 = fma(cp, fma(cp, a0, 2.0 * dm), dnd);       // r^T r
          const float s1f = (float)s1;
          rnn = __builtin_amdgcn_sqrtf(s1f < 0.f ? 0.f : s1f);
        }
        if (k >= 0) {                                                // closes iteration k: beta, residual norm, records
          beta = ((float)rz < a.eps) ? 0.0 : (double)((float)rzn * inv_rz);  // :39-42
          if (rhs_zero) rnn = 0.f;                                   // :299
          rn = rnn;
          if (rec && (rec_first || rec_last)) a.resid_rec[(size_t)k * a.B + bc] = rn;
        } else {
          beta = 0.0;
          rn = rnn;
          if (wig == 0 && t == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
          close_flags = ((rn < a.stop_after) ? 1u : 0u) | (delicate ? 4u : 0u);
          if (delicate && rec && wave == 0) atomicOr(a.err + 4, 1);  // (CgCtrl::rs_redo, for the host-launched control step)
        }
        // (NaN after the first product, linear_cg.py:199-200: the residual norm is NaN exactly when the coordinates are)
        if (k == 0 && (rnn != rnn || rzn != rzn)) close_flags |= 2u;
        conv = (own || rhs_zero) ? (rn < a.stop_after) : false;      // :300
        rz = rzn;
        // p = z + beta p (:268, :46);  p.Ap = sum lam (c + beta q)^2 + q'^2 tau2
        pApE = fma(beta, fma(beta, pApE, 2.0 * clq), clc);
        qj = fma(beta, qj, cj);
        qp = fma(beta, qp, cp);
        const double pAp = fma(qp * qp, tau2, pApE);
        alpha = ((float)pAp < a.eps) ? 0.0 : (double)((float)rz * __builtin_amdgcn_rcpf((float)pAp));  // :254-257
        if (conv) alpha = 0.0;                                       // :260
      }
      if (stamp) ts_b = wall_clock64();
      if (rec && wave == last_owner) {
        a.rhs_norm[bc] = nrm;
        a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
        a.rz[bc] = (float)rz;
        a.alpha[bc] = last_alpha;
        a.beta[bc] = (float)beta;
        a.resid_norm[bc] = rn;
        a.has_conv[bc] = conv ? 1 : 0;
        if (a.close_gran) {
          const unsigned long long gr =
              ((unsigned long long)(0x80000000u | close_flags) << 32) | (unsigned long long)__float_as_uint(rn);
          __hip_atomic_store(a.close_gran + b, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // y = Tin (eta - xi g0): lane i walks column i of TinT (its half of the eigen-indices; consecutive lanes read
      // consecutive doubles, the vector is a broadcast) -- Tin itself never has to be on chip
      __builtin_amdgcn_wave_barrier();
      if (hf == 0) myv[j] = fma(-etap, e0, etaj);
      __builtin_amdgcn_wave_barrier();
      xi = etap;
      {
        const double* col = mat_s + (size_t)(hf * NH) * MLD + jr;
        double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
        for (int q = 0; q < NH; q += 2) {
          const double2 ee = *reinterpret_cast<const double2*>(myv + hf * NH + q);
          a0_ = fma(col[(size_t)q * MLD], ee.x, a0_);
          a1_ = fma(col[(size_t)(q + 1) * MLD], ee.y, a1_);
        }
        double lo_, up_;
        halves_d(a0_ + a1_, lo_, up_);
        yj = lo_ + up_;
      }
      yj = live ? yj : 0.0;
      if (stamp) {  // (printed as wg-wait / publish / poll: start of the chain, the iterations, y)
        a.dbg[5] = ts_a - a.dbg[2];
        a.dbg[6] = ts_b - ts_a;
        a.dbg[7] = wall_clock64() - ts_b;
      }
      __syncthreads();  // the only barrier of the phase: the next member's matrices may replace these from here on
    } else
    {
      const int iw = (jr / H) * 2 * H + (jr % H);
      double a0 = rs.res[2 * RC + 1], s = rs.res[2 * RC];
      nrm = sqrtf((float)a0);                             // rhs.norm(2, dim=-2)          :177
      const bool rhs_zero = nrm < a.eps;                  // :178
      if (rhs_zero) nrm = 1.0f;                           // :179
      const double inv = 1.0 / (double)nrm;
      const double w0 = live ? rs.res[iw] * inv : 0.0, u0 = live ? rs.res[iw + H] * inv : 0.0;
      s *= inv * inv;
      a0 *= inv * inv;
      // this wave's matrix row (half of it per half-wave) against a vector the wave has just written to its own LDS copy
      auto own_row_dot = [&](const double* mrow, const double* vec) {
        double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
        for (int q = 0; q < NH; q += 2) {
          const double2 xa = *reinterpret_cast<const double2*>(mrow + q);
          const double2 xx = *reinterpret_cast<const double2*>(vec + q);
          a0_ = fma(xa.x, xx.x, a0_);
          a1_ = fma(xa.y, xx.y, a1_);
        }
        double lo_, up_;
        halves_d(a0_ + a1_, lo_, up_);
        return lo_ + up_;
      };
      double* myv = rs.gv[wave];
      double Fw0 = 0.0, EFw0 = 0.0;
      {  // F w0, E F w0
        if (hf == 0) myv[j] = w0;
        __builtin_amdgcn_wave_barrier();
        if (wave == 1 || wave == 2) {
          double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
          for (int q = 0; q < NH; q += 2) {
            const double2 xx = *reinterpret_cast<const double2*>(myv + hf * NH + q);
            const double fx = mk_d(__float_as_uint(frow[q / 2].x), __float_as_uint(frow[q / 2].y));
            const double fy = mk_d(__float_as_uint(frow[q / 2].z), __float_as_uint(frow[q / 2].w));
            a0_ = fma(fx, xx.x, a0_);
            a1_ = fma(fy, xx.y, a1_);
          }
          double lo_, up_;
          halves_d(a0_ + a1_, lo_, up_);
          if (hf == 0) rs.out[1][wave][j] = lo_ + up_;
        }
        __syncthreads();
        Fw0 = live ? rs.out[1][1][j] : 0.0;
        EFw0 = live ? rs.out[1][2][j] : 0.0;
      }
      double gj = 0.0, hj = 0.0, Ehj = 0.0, tj = 0.0;
      double rho = 1.0, pi = 0.0, rz = 0.0, dpp = 0.0, tt = 0.0, hw0 = 0.0;
      double alpha = 0.0, beta = 0.0;
      float rn = 0.f, last_alpha = 0.f;
      bool conv = false;
      unsigned close_flags = 0u;
      const size_t bc = (size_t)b;
      const double* mrow = mat_s + (size_t)(wave * RC + jr) * MLD + hf * NH;
      for (int k = -1; k < a.iters; ++k) {
        double Eg = 0.0, FEg = 0.0, EFEg = 0.0, G2g = 0.0;
        if (k >= 0) {  // x += alpha p (:31);  r -= alpha A p (:264), A p = pi r0 + C (h + t)
          last_alpha = (float)alpha;
          xi = fma(alpha, pi, xi);
          yj = fma(alpha, hj, yj);
          rho = fma(-alpha, pi, rho);
          gj = fma(-alpha, hj + tj, gj);
          if (hf == 0) myv[j] = gj;
          __builtin_amdgcn_wave_barrier();
          const double mine = own_row_dot(mrow, myv + hf * NH);
          const int par = k & 1;
          if (hf == 0) rs.out[par][wave][j] = mine;
          __syncthreads();
          if (live) {
            Eg = rs.out[par][0][j];
            FEg = rs.out[par][1][j];
            EFEg = rs.out[par][2][j];
            G2g = rs.out[par][3][j];
          }
        }
        const double wj = fma(rho, w0, Eg);
        const double vj = fma(rho, Fw0, FEg);
        const double Evj = fma(rho, EFw0, EFEg);
        const double tz = wj - Evj;                        // (C^T z)_j
        // eleven inner products over the components: the lower half-wave forms six, the upper five; ONE reduce-scatter of
        // eight values per half, the totals come back as wave-uniform scalars (v_readlane)
        const bool lo_h = hf == 0;
        double pr[8];
        pr[0] = lo_h ? gj * w0 : vj * Evj;
        pr[1] = lo_h ? gj * wj : vj * tj;
        pr[2] = lo_h ? gj * u0 : tz * tz;
        pr[3] = lo_h ? gj * G2g : tz * tj;
        pr[4] = lo_h ? wj * vj : vj * w0;
        pr[5] = lo_h ? gj * Ehj : 0.0;
        pr[6] = 0.0;
        pr[7] = 0.0;
        double red8;
        {
          double v4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v4[q] = halve_pair_d<16>(pr[q], pr[q + 4], lane);
          halving_steps_d<4, 8, 4>(v4, lane);  // bits 8, 4 (halving), 2, 1 (sums): lane l holds component (l >> 2) & 7
          red8 = v4[0];
        }
        const auto pick = [&](int ln) {
          return mk_d((unsigned)__builtin_amdgcn_readlane((int)lo_w(red8), ln),
                      (unsigned)__builtin_amdgcn_readlane((int)hi_w(red8), ln));
        };
        const double gw0 = pick(0), gw = pick(4), gu0 = pick(8), gG2g = pick(12), wv = pick(16), gEh = pick(20);
        const double vEv = pick(32), vt = pick(36), tztz = pick(40), tzt = pick(44), vw0 = pick(48);
        const double s2 = fma(rho, fma(rho, s, gw0), gw);            // r^T D^-1 r
        const double s1 = fma(rho, fma(rho, a0, 2.0 * gu0), gG2g);   // r^T r
        const double rzn = s2 - wv;                                  // residual_inner_prod :215 / :35-36
        const float s1f = (float)s1;                                 // (tiny negative by rounding -> 0; NaN stays NaN)
        float rnn = __builtin_amdgcn_sqrtf(s1f < 0.f ? 0.f : s1f);   // :298 / :204
        if (k >= 0) {                                                // closes iteration k: beta, residual norm, records
          beta = ((float)rz < a.eps) ? 0.0 : (double)((float)rzn * __builtin_amdgcn_rcpf((float)rz));  // :39-42
          if (rhs_zero) rnn = 0.f;                                   // :299
          rn = rnn;
          if (wig == 0 && t == 0) a.resid_rec[(size_t)k * a.B + bc] = rn;
        } else {
          beta = 0.0;
          rn = rnn;
          if (wig == 0 && t == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
          close_flags = (rn < a.stop_after) ? 1u : 0u;
        }
        if (k == 0 && rnn != rnn) close_flags |= 2u;  // (NaN after the first product, linear_cg.py:199-200)
        conv = rn < a.stop_after;                                    // :300
        rz = rzn;
        const double dzz = fma(-2.0, wv, s2) + vEv;                  // sum d z^2
        const double rp = fma(rho, fma(pi, s, hw0), fma(pi, gw0, gEh));  // r^T p_old
        const double dzp = rp - vt;
        dpp = fma(beta, fma(beta, dpp, 2.0 * dzp), dzz);             // sum d p_new^2
        tt = fma(beta, fma(beta, tt, 2.0 * tzt), tztz);              // |C^T p_new|^2
        tj = fma(beta, tj, tz);                                      // C^T p_new
        pi = fma(beta, pi, rho);                                     // p = z + beta p (:268, :46)
        hj = fma(beta, hj, gj - vj);
        Ehj = fma(beta, Ehj, Eg - Evj);
        hw0 = fma(beta, hw0, gw0 - vw0);
        const double pAp = tt + dpp;
        alpha = ((float)pAp < a.eps) ? 0.0 : (double)((float)rz * __builtin_amdgcn_rcpf((float)pAp));  // :254-257
        if (conv) alpha = 0.0;                                       // :260
      }
      if (wig == 0 && t == 0) {
        a.rhs_norm[bc] = nrm;
        a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
        a.rz[bc] = (float)rz;
        a.alpha[bc] = last_alpha;
        a.beta[bc] = (float)beta;
        a.resid_norm[bc] = rn;
        a.has_conv[bc] = conv ? 1 : 0;
        if (a.close_gran) {  // this member's line of the closing step: one never-torn 8-byte store
          const unsigned long long gr =
              ((unsigned long long)(0x80000000u | close_flags) << 32) | (unsigned long long)__float_as_uint(rn);
          __hip_atomic_store(a.close_gran + b, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!(a.prefetch & 2)) __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[3] = wall_clock64();
    // ---- x = nrm D^-1 (xi r0 + C y) = D^-1 (xi b + C (nrm y)), in fp64 (for small diagonals the two terms cancel); every
    // wave reads its own copy of y ----
    {
      const float4* win = stage_s + wave * (64 * (RC / 4));
#pragma unroll
      for (int i = 0; i < RC / 4; ++i) {
        const float4 c4 = win[i * 64 + lane];
        Cr[R4_NR - 1][2 * i] = f32x2{c4.x, c4.y}; Cr[R4_NR - 1][2 * i + 1] = f32x2{c4.z, c4.w};
      }
      double* myv = rs.gv[wave];
      if (hf == 0) myv[j] = yj * (double)nrm;
      __builtin_amdgcn_wave_barrier();
      double acc[R4_NR];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) acc[q] = xi * (double)bq[q];
      // (eight coordinates of y at a time: with all 32 in registers at once the compiler spills four doubles here)
#pragma unroll
      for (int i0 = 0; i0 < RC; i0 += 8) {
#pragma unroll
        for (int i = i0; i < i0 + 8; i += 2) {
          const double2 y2 = *reinterpret_cast<const double2*>(&myv[i]);
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) {
            acc[q] = fma((double)Cr[q][i >> 1].x, y2.x, acc[q]);
            acc[q] = fma((double)Cr[q][i >> 1].y, y2.y, acc[q]);
          }
        }
        // (pins the partial sums here: otherwise the products sink into the four guarded stores below, each of which
        //  then wants all of y in registers)
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) asm volatile("" : "+v"(acc[q]));
      }
      int tw = t;
      asm volatile("" : "+v"(tw));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q)
        if (tw + R4_TPB * q < nv) a.xout[(size_t)b * a.N + row0 + tw + R4_TPB * q] = (float)(acc[q] * (double)diq[q]);  // :335
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[4] = wall_clock64();
    b = b_next;
  }
  if (a.close_gran && wig == 0) cg_close_solve(a, ngroups, t);
}

// =====================================================================================================================
// Several columns (inv_quad_logdet: 16 probes + the right-hand side, with the alpha / beta records of the Lanczos
// tridiagonals, linear_cg.py:311-332): the same iteration per column, in THREE streaming launches instead of a resident one
//   k_rs_part   w0 | u0 = C^T [D^-1 B | B] for all columns on the fp64 matrix cores (no cross-lane reduction: the matrix
//               cores sum over the rows), s = sum b^2 / d, a0 = sum b^2 on the vector ALU; partials per row slice
//   k_rs_iter   one wave per (member, column): the iterations on R + 1 coordinates (the single-wave form of the algebra
//               above: four R x R products with g per iteration from LDS), alpha / beta / residual records, (xi, nrm y)
//   k_rs_apply  x = D^-1 (xi b + C (nrm y)) for all columns, fp64
// C is read twice, the right-hand sides twice, x written once; nothing has to be co-resident.
// =====================================================================================================================
constexpr int RSP_TR = 64;   // rows per tile of k_rs_part (16 per wave)
constexpr int RSP_CMAX = 32; // columns per launch
constexpr int RSP_SQ = 16;   // row classes of the s / a0 sums (threads per column)

// VL: the LAST column (inv_quad_logdet: the right-hand side behind 16 probes) runs on the vector ALU -- 2 c = 34 operand
// columns would open a third 16-column tile for two columns (24 instead of 16 matrix instructions per wave and tile:
// 330 -> 265 us at the cfg3 shape); the matrix cores then see cm = c - 1 columns, twice.
template <int RC, int NT, bool VL>
__global__ __launch_bounds__(kThreads) void k_rs_part(const float* __restrict__ C, const float* __restrict__ rhs,
                                                       const float* __restrict__ dinv, int dinv_mode, int N, int c,
                                                       int rows_per, double* __restrict__ part, double* __restrict__ sq) {
  constexpr int MT = RC > 16 ? 2 : 1;
  constexpr int TR = RSP_TR;
  constexpr int LD = 16 * MT + 1;  // C tile row stride (columns beyond RC stay zero)
  constexpr int WB = 16 * NT;      // B tile row stride: [b (c columns) | b again (c columns) | zeros]
  __shared__ __attribute__((aligned(16))) float ctile[2][TR * LD];
  __shared__ __attribute__((aligned(16))) float btile[2][TR * WB];
  __shared__ float dtile[2][TR];
  __shared__ float ltile[2][VL ? TR : 1];  // the last column (VL)
  __shared__ double sqred[RSP_SQ][RSP_CMAX][2];
  // (the cross-wave reduction at the end reuses the tiles)
  static_assert(sizeof(float) * 2 * TR * WB >= sizeof(double) * 4 * 64 * 4 || sizeof(float) * 2 * TR * LD >= sizeof(double) * 4 * 64 * 4,
                "reduction buffer inside a tile");
  double (*red)[64][4] = (sizeof(float) * 2 * TR * LD >= sizeof(double) * 4 * 64 * 4)
                             ? reinterpret_cast<double (*)[64][4]>(&ctile[0][0])
                             : reinterpret_cast<double (*)[64][4]>(&btile[0][0]);
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int t = threadIdx.x, wave = t >> 6, l = t & 63;
  const int a = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float* Cb = C + (size_t)b * N * RC;
  const float* Bb = rhs + (size_t)b * N * c;
  const bool dfull = dinv_mode == LO_DIAG_FULL;
  const float dconst = dfull ? 0.f : dinv[b];
  const float* db = dfull ? dinv + (size_t)b * N : nullptr;
  constexpr int RQ = RC / 4;
  constexpr int CP = (TR * RQ + kThreads - 1) / kThreads;  // 16-byte pieces of C per thread and tile
  constexpr int BP = TR * RSP_CMAX / kThreads;             // floats of the right-hand sides per thread and tile (at c = 32)
  // element e = i * 256 + t of a tile's right-hand sides (rows x c, contiguous) lands at row e / c, column e % c: the
  // same for every tile -- one division per element for the whole kernel
  const int cm = VL ? c - 1 : c;  // columns on the matrix cores
  int bdst[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int e = i * kThreads + t;
    bdst[i] = (VL && e % c == cm) ? -1 - e / c : (e / c) * WB + (e % c);  // (VL: the last column goes to ltile[row])
  }
  for (int e = t; e < 2 * TR * WB; e += kThreads) (&btile[0][0])[e] = 0.f;  // (the zero columns are never written again)
  for (int e = t; e < 2 * TR * LD; e += kThreads) (&ctile[0][0])[e] = 0.f;
  __syncthreads();
  float4 pc[CP];
  float pb[BP];
  float pd = 0.f;
  auto issue = [&](int base) {
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const int e = i * kThreads + t;
      const int row = e / RQ, q = e % RQ;
      pc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < TR * RQ && base + row < r1) pc[i] = *reinterpret_cast<const float4*>(Cb + (size_t)(base + row) * RC + 4 * q);
    }
    const int nb = min(TR, r1 - base) * c;  // the tile's right-hand sides are contiguous: rows x c
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int e = i * kThreads + t;
      pb[i] = (e < nb) ? Bb[(size_t)base * c + e] : 0.f;
    }
    pd = 0.f;
    if (t < TR && base + t < r1) pd = dfull ? db[base + t] : dconst;
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const int e = i * kThreads + t;
      if (e < TR * RQ) {
        const int row = e / RQ, q = e % RQ;
        float* dst = &ctile[buf][row * LD + 4 * q];
        dst[0] = pc[i].x; dst[1] = pc[i].y; dst[2] = pc[i].z; dst[3] = pc[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int e = i * kThreads + t;
      if (e < TR * c) {
        if (VL && bdst[i] < 0) {
          ltile[buf][-1 - bdst[i]] = pb[i];
        } else {
          btile[buf][bdst[i]] = pb[i];
          btile[buf][bdst[i] + cm] = pb[i];
        }
      }
    }
    if (t < TR) dtile[buf][t] = pd;
  };
  f64x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) acc[mi][nj] = f64x4{0.0, 0.0, 0.0, 0.0};
  // lane (a, kk) feeds column jc = 16 nj + a of the B operand: columns [0, cm) carry b dinv, [cm, 2 cm) carry b
  bool scaled[NT];
#pragma unroll
  for (int nj = 0; nj < NT; ++nj) scaled[nj] = 16 * nj + a < cm;
  double lw[MT], lu[MT];  // VL: C^T (dinv o b_last) and C^T b_last of the lane's rows (kk + 4 e) and components 16 mi + a
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) lw[mi] = lu[mi] = 0.0;
  // s = sum b^2 dinv, a0 = sum b^2: thread (col = t % c, rsub = t / c < nsq) walks the rows rsub, rsub + nsq, ... of its
  // column -- ALL threads share the work (with four row classes the 4 c threads of the first wave did 16 rows each on the
  // fp64 pipe the matrix instructions also run on, and every tile's barrier waited for that wave)
  const int nsq = min(RSP_SQ, kThreads / c);
  const int scol = t % c, rsub = t / c;
  double s_acc = 0.0, a_acc = 0.0;
  int buf = 0;
  issue(r0);
  commit(0);
  __syncthreads();
  for (int base = r0; base < r1; base += TR) {
    const bool more = base + TR < r1;
    if (more) issue(base + TR);
    float avf[4][MT], bvf[4][NT], dvf[4], lvf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // all operands of the wave's 16 rows first (branch-free), then the products
      const int row = 16 * wave + 4 * e + kk;
      dvf[e] = dtile[buf][row];
      lvf[e] = VL ? ltile[buf][row] : 0.f;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) avf[e][mi] = ctile[buf][row * LD + 16 * mi + a];
#pragma unroll
      for (int nj = 0; nj < NT; ++nj) bvf[e][nj] = btile[buf][row * WB + 16 * nj + a];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double dv = (double)dvf[e];
      double av[MT], bv[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) av[mi] = (double)avf[e][mi];
#pragma unroll
      for (int nj = 0; nj < NT; ++nj) bv[nj] = (double)bvf[e][nj] * (scaled[nj] ? dv : 1.0);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nj = 0; nj < NT; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mi], bv[nj], acc[mi][nj], 0, 0, 0);
      if constexpr (VL) {
        const double bl = (double)lvf[e], bld = bl * dv;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          lw[mi] = fma(av[mi], bld, lw[mi]);
          lu[mi] = fma(av[mi], bl, lu[mi]);
        }
      }
    }
    if (rsub < nsq) {
      for (int row = rsub; row < TR; row += nsq) {
        const double v = (double)((VL && scol == cm) ? ltile[buf][row] : btile[buf][row * WB + scol]);
        const double vv = v * v;
        a_acc += vv;
        s_acc = fma(vv, (double)dtile[buf][row], s_acc);
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  double* pp = part + ((size_t)b * S + s) * RC * 2 * c;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][l][r] = acc[mi][nj][r];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = (red[0][l][r] + red[1][l][r]) + (red[2][l][r] + red[3][l][r]);
          const int i = 16 * mi + 4 * r + kk, jc = 16 * nj + a;  // D[4 r + l / 16][l % 16]
          // operand column jc -> result column: [0, cm) stay, [cm, 2 cm) move behind the c scaled ones
          if (i < RC && jc < 2 * cm) pp[(size_t)i * 2 * c + (jc < cm ? jc : jc - cm + c)] = v;
        }
      }
    }
  if constexpr (VL) {  // the last column: sum over the four row classes kk (lane bits 4, 5), then over the waves
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      lw[mi] = bfly_add_d<32>(bfly_add_d<16>(lw[mi]));
      lu[mi] = bfly_add_d<32>(bfly_add_d<16>(lu[mi]));
    }
    __syncthreads();
    if (kk == 0) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        red[wave][a][2 * mi] = lw[mi];
        red[wave][a][2 * mi + 1] = lu[mi];
      }
    }
    __syncthreads();
    if (wave == 0 && kk == 0) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int i = 16 * mi + a;
        if (i < RC) {
          pp[(size_t)i * 2 * c + cm] = (red[0][a][2 * mi] + red[1][a][2 * mi]) + (red[2][a][2 * mi] + red[3][a][2 * mi]);
          pp[(size_t)i * 2 * c + c + cm] =
              (red[0][a][2 * mi + 1] + red[1][a][2 * mi + 1]) + (red[2][a][2 * mi + 1] + red[3][a][2 * mi + 1]);
        }
      }
    }
  }
  if (rsub < nsq) {
    sqred[rsub][scol][0] = s_acc;
    sqred[rsub][scol][1] = a_acc;
  }
  __syncthreads();
  if (t < c) {
    double* sp = sq + ((size_t)b * S + s) * 2 * c;
    double ss = 0.0, aa = 0.0;
    for (int i = 0; i < nsq; ++i) {
      ss += sqred[i][t][0];
      aa += sqred[i][t][1];
    }
    sp[t] = ss;
    sp[c + t] = aa;
  }
}

struct RsColsArgs {
  const float* C; const float* rhs; const float* dinv; int dinv_mode;
  const double* RS;
  int64_t B; int N; int c; int S; int rows_per;
  int iters; float eps, stop_after;
  double* part; double* sq; double* coef;  // [B,S,RC,2c] | [B,S,2c] | [B,c,RC+1]
  float* xout;
  float* ab_rec; float* resid_rec; int* init_conv;
  float *rhs_norm, *rz, *alpha, *beta, *resid_norm;
  int *rhs_is_zero, *has_conv;
};

template <int RC>
__global__ __launch_bounds__(kThreads) void k_rs_iter(RsColsArgs a) {
  constexpr int MLD = RC + 2;
  constexpr int NH = RC / 2;
  __shared__ __attribute__((aligned(16))) double mat_s[4 * RC * MLD];  // E | F E | E F E | G2
  __shared__ __attribute__((aligned(16))) double gv_s[4][32];
  const int64_t b = blockIdx.y;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int col = 4 * blockIdx.x + wave;
  const int c = a.c;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.RS + (size_t)b * 6 * RC * RC);
    for (int e = t; e < 4 * RC * RC / 2; e += kThreads) {
      const int mi = e / (RC / 2), jj = e % (RC / 2);
      *reinterpret_cast<f32x4*>(&mat_s[mi * MLD + 2 * jj]) = src[e];
    }
  }
  __syncthreads();
  if (col >= c) return;
  const int j = lane & 31, hf = lane >> 5;
  const bool live = j < RC;
  const int jr = live ? j : RC - 1;
  // the column's reduction: partials of the row slices in fixed order
  double w0 = 0.0, u0 = 0.0, s = 0.0, a0 = 0.0;
  for (int sl = 0; sl < a.S; ++sl) {
    const double* pp = a.part + (((size_t)b * a.S + sl) * RC + jr) * 2 * c;
    w0 += pp[col];
    u0 += pp[c + col];
    const double* sp = a.sq + ((size_t)b * a.S + sl) * 2 * c;
    s += sp[col];
    a0 += sp[c + col];
  }
  float nrm = sqrtf((float)a0);                       // rhs.norm(2, dim=-2)          :177
  const bool rhs_zero = nrm < a.eps;                  // :178
  if (rhs_zero) nrm = 1.0f;                           // :179
  const double inv = 1.0 / (double)nrm;
  w0 = live ? w0 * inv : 0.0;
  u0 = live ? u0 * inv : 0.0;
  s *= inv * inv;
  a0 *= inv * inv;
  double* myv = gv_s[wave];
  auto row_dot = [&](const double* mrow, const double* vec) {
    double a0_ = 0.0, a1_ = 0.0;
#pragma unroll
    for (int q = 0; q < NH; q += 2) {
      const double2 xa = *reinterpret_cast<const double2*>(mrow + q);
      const double2 xx = *reinterpret_cast<const double2*>(vec + q);
      a0_ = fma(xa.x, xx.x, a0_);
      a1_ = fma(xa.y, xx.y, a1_);
    }
    double lo_, up_;
    halves_d(a0_ + a1_, lo_, up_);
    return lo_ + up_;
  };
  double Fw0, EFw0;
  {  // F w0, E F w0: rows of F / E F straight from memory (once per column)
    if (hf == 0) myv[j] = w0;
    __builtin_amdgcn_wave_barrier();
    const double* Frow = a.RS + ((size_t)b * 6 + 4) * RC * RC + (size_t)jr * RC + hf * NH;
    const double* EFrow = a.RS + ((size_t)b * 6 + 5) * RC * RC + (size_t)jr * RC + hf * NH;
    Fw0 = row_dot(Frow, myv + hf * NH);
    EFw0 = row_dot(EFrow, myv + hf * NH);
    if (!live) { Fw0 = 0.0; EFw0 = 0.0; }
  }
  double gj = 0.0, hj = 0.0, Ehj = 0.0, tj = 0.0, yj = 0.0;
  double rho = 1.0, pi = 0.0, xi = 0.0, rz = 0.0, dpp = 0.0, tt = 0.0, hw0 = 0.0;
  double alpha = 0.0, beta = 0.0;
  float rn = 0.f, last_alpha = 0.f;
  bool conv = false;
  const size_t bc = (size_t)b * c + col;
  const size_t nbc = (size_t)a.B * c;
  const double* m0 = mat_s + (size_t)(0 * RC + jr) * MLD + hf * NH;
  const double* m1 = mat_s + (size_t)(1 * RC + jr) * MLD + hf * NH;
  const double* m2 = mat_s + (size_t)(2 * RC + jr) * MLD + hf * NH;
  const double* m3 = mat_s + (size_t)(3 * RC + jr) * MLD + hf * NH;
  for (int k = -1; k < a.iters; ++k) {
    double Eg = 0.0, FEg = 0.0, EFEg = 0.0, G2g = 0.0;
    if (k >= 0) {  // x += alpha p (:31);  r -= alpha A p (:264), A p = pi r0 + C (h + t)
      last_alpha = (float)alpha;
      xi = fma(alpha, pi, xi);
      yj = fma(alpha, hj, yj);
      rho = fma(-alpha, pi, rho);
      gj = fma(-alpha, hj + tj, gj);
      __builtin_amdgcn_wave_barrier();
      if (hf == 0) myv[j] = gj;
      __builtin_amdgcn_wave_barrier();
      const double* xv = myv + hf * NH;
      double e0 = 0.0, e1 = 0.0, f0 = 0.0, f1 = 0.0, h0 = 0.0, h1 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
      for (int q = 0; q < NH; q += 2) {
        const double2 xx = *reinterpret_cast<const double2*>(xv + q);
        const double2 x0 = *reinterpret_cast<const double2*>(m0 + q);
        const double2 x1 = *reinterpret_cast<const double2*>(m1 + q);
        const double2 x2 = *reinterpret_cast<const double2*>(m2 + q);
        const double2 x3 = *reinterpret_cast<const double2*>(m3 + q);
        e0 = fma(x0.x, xx.x, e0); e1 = fma(x0.y, xx.y, e1);
        f0 = fma(x1.x, xx.x, f0); f1 = fma(x1.y, xx.y, f1);
        h0 = fma(x2.x, xx.x, h0); h1 = fma(x2.y, xx.y, h1);
        g0 = fma(x3.x, xx.x, g0); g1 = fma(x3.y, xx.y, g1);
      }
      double lo_, up_;
      halves_d(e0 + e1, lo_, up_); Eg = lo_ + up_;
      halves_d(f0 + f1, lo_, up_); FEg = lo_ + up_;
      halves_d(h0 + h1, lo_, up_); EFEg = lo_ + up_;
      halves_d(g0 + g1, lo_, up_); G2g = lo_ + up_;
      if (!live) { Eg = 0.0; FEg = 0.0; EFEg = 0.0; G2g = 0.0; }
    }
    const double wj = fma(rho, w0, Eg);
    const double vj = fma(rho, Fw0, FEg);
    const double Evj = fma(rho, EFw0, EFEg);
    const double tz = wj - Evj;                        // (C^T z)_j
    const bool lo_h = hf == 0;
    double pr[8];
    pr[0] = lo_h ? gj * w0 : vj * Evj;
    pr[1] = lo_h ? gj * wj : vj * tj;
    pr[2] = lo_h ? gj * u0 : tz * tz;
    pr[3] = lo_h ? gj * G2g : tz * tj;
    pr[4] = lo_h ? wj * vj : vj * w0;
    pr[5] = lo_h ? gj * Ehj : 0.0;
    pr[6] = 0.0;
    pr[7] = 0.0;
    double red8;
    {
      double v4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v4[q] = halve_pair_d<16>(pr[q], pr[q + 4], lane);
      halving_steps_d<4, 8, 4>(v4, lane);  // bits 8, 4 (halving), 2, 1 (sums): lane l holds component (l >> 2) & 7
      red8 = v4[0];
    }
    const auto pick = [&](int ln) {
      return mk_d((unsigned)__builtin_amdgcn_readlane((int)lo_w(red8), ln),
                  (unsigned)__builtin_amdgcn_readlane((int)hi_w(red8), ln));
    };
    const double gw0 = pick(0), gw = pick(4), gu0 = pick(8), gG2g = pick(12), wv = pick(16), gEh = pick(20);
    const double vEv = pick(32), vt = pick(36), tztz = pick(40), tzt = pick(44), vw0 = pick(48);
    const double s2 = fma(rho, fma(rho, s, gw0), gw);            // r^T D^-1 r
    const double s1 = fma(rho, fma(rho, a0, 2.0 * gu0), gG2g);   // r^T r
    const double rzn = s2 - wv;                                  // residual_inner_prod :215 / :35-36
    const float s1f = (float)s1;
    float rnn = __builtin_amdgcn_sqrtf(s1f < 0.f ? 0.f : s1f);   // :298 / :204
    if (k >= 0) {
      beta = ((float)rz < a.eps) ? 0.0 : (double)((float)rzn * __builtin_amdgcn_rcpf((float)rz));  // :39-42
      if (rhs_zero) rnn = 0.f;                                   // :299
      rn = rnn;
      if (lane == 0) {
        a.resid_rec[(size_t)k * nbc + bc] = rn;
        if (a.ab_rec) {
          a.ab_rec[2 * ((size_t)k * nbc + bc)] = (float)alpha;
          a.ab_rec[2 * ((size_t)k * nbc + bc) + 1] = (float)beta;
        }
      }
    } else {
      beta = 0.0;
      rn = rnn;
      if (lane == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
    }
    conv = rn < a.stop_after;                                    // :300
    rz = rzn;
    const double dzz = fma(-2.0, wv, s2) + vEv;
    const double rp = fma(rho, fma(pi, s, hw0), fma(pi, gw0, gEh));
    const double dzp = rp - vt;
    dpp = fma(beta, fma(beta, dpp, 2.0 * dzp), dzz);
    tt = fma(beta, fma(beta, tt, 2.0 * tzt), tztz);
    tj = fma(beta, tj, tz);
    pi = fma(beta, pi, rho);
    hj = fma(beta, hj, gj - vj);
    Ehj = fma(beta, Ehj, Eg - Evj);
    hw0 = fma(beta, hw0, gw0 - vw0);
    const double pAp = tt + dpp;
    alpha = ((float)pAp < a.eps) ? 0.0 : (double)((float)rz * __builtin_amdgcn_rcpf((float)pAp));  // :254-257
    if (conv) alpha = 0.0;                                       // :260
  }
  double* cf = a.coef + bc * (RC + 1);
  if (lane == 0) cf[0] = xi;
  if (hf == 0 && live) cf[1 + j] = yj * (double)nrm;
  if (lane == 0) {
    a.rhs_norm[bc] = nrm;
    a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
    a.rz[bc] = (float)rz;
    a.alpha[bc] = last_alpha;
    a.beta[bc] = (float)beta;
    a.resid_norm[bc] = rn;
    a.has_conv[bc] = conv ? 1 : 0;
  }
}

// x[b, row, col] = dinv[row] (xi_col b[row, col] + sum_j C[row, j] yn[col][j]).  A pass of a workgroup covers NQ row sets
// of 256 rows (NQ = 2 up to 17 columns: the broadcast reads of yn serve two rows per thread); a wave fetches its 64 rows
// of C as consecutive 16-byte pieces (whole cache lines per instruction) and turns them into one row per lane through
// its own LDS window; the right-hand sides / results of a pass travel through LDS as well (rows x c floats contiguous).
constexpr int RSA_BT = 512 * 17;  // floats of the right-hand-side tile
template <int RC, int NQ>
__global__ __launch_bounds__(kThreads) void k_rs_apply(RsColsArgs a) {
  constexpr int CH = RC / 4;
  __shared__ float bt[RSA_BT];
  __shared__ float4 win_s[4 * 64 * CH];
  __shared__ __attribute__((aligned(16))) double yv[RSP_CMAX][RC + 2];  // [col][yn_0 .. yn_RC-1 | xi | pad]
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int c = a.c, N = a.N;
  const int ldb = c;  // (row stride of the staged tile = c: conflict-free for odd c, 2-way .. 16-way for even c)
  const int r0 = s * a.rows_per, r1 = min(N, r0 + a.rows_per);
  for (int e = t; e < c * (RC + 1); e += kThreads) {
    const int col = e / (RC + 1), q = e % (RC + 1);
    yv[col][q == 0 ? RC : q - 1] = a.coef[((size_t)b * c + col) * (RC + 1) + q];
  }
  const bool dfull = a.dinv_mode == LO_DIAG_FULL;
  const float* Cb = a.C + (size_t)b * N * RC;
  float4* win = win_s + wave * (64 * CH);
  constexpr int ROWS = NQ * kThreads;
  for (int base = r0; base < r1; base += ROWS) {
    const int nr = min(ROWS, r1 - base);
    __syncthreads();
    for (int e = t; e < nr * c; e += kThreads) bt[e] = a.rhs[((size_t)b * N + base) * c + e];  // (row stride c: no division)
    double cr[NQ][RC];
    double dv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int wrow = base + kThreads * q + 64 * wave;  // first row of this wave's block
      float4 pc[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int gch = 64 * i + lane;  // 16-byte piece of the wave's 64 x RC block
        const int rw = gch / CH, ck = gch % CH;
        pc[i] = *reinterpret_cast<const float4*>(Cb + (size_t)min(wrow + rw, N - 1) * RC + 4 * ck);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int gch = 64 * i + lane;
        const int rw = gch / CH, ck = gch % CH;
        win[rw * CH + (ck ^ ((rw ^ (rw >> 3)) & (CH - 1)))] = pc[i];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float4 c4 = win[lane * CH + (i ^ ((lane ^ (lane >> 3)) & (CH - 1)))];
        cr[q][4 * i] = (double)c4.x; cr[q][4 * i + 1] = (double)c4.y;
        cr[q][4 * i + 2] = (double)c4.z; cr[q][4 * i + 3] = (double)c4.w;
      }
      __builtin_amdgcn_wave_barrier();
      const int row = wrow + lane;
      dv[q] = (row < r1) ? (double)(dfull ? a.dinv[(size_t)b * N + row] : a.dinv[b]) : 0.0;
    }
    __syncthreads();
    for (int col = 0; col < c; ++col) {
      const double xi = yv[col][RC];
      double acc[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = xi * (double)bt[(t + kThreads * q) * ldb + col];
#pragma unroll
      for (int i = 0; i < RC; i += 2) {
        const double2 y2 = *reinterpret_cast<const double2*>(&yv[col][i]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          acc[q] = fma(cr[q][i], y2.x, acc[q]);
          acc[q] = fma(cr[q][i + 1], y2.y, acc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) bt[(t + kThreads * q) * ldb + col] = (float)(acc[q] * dv[q]);  // :335 (own rows only)
    }
    __syncthreads();
    for (int e = t; e < nr * c; e += kThreads) a.xout[((size_t)b * N + base) * c + e] = bt[e];
  }
}

Split rspace_cols_split(int64_t B, int64_t N) {
  // row slices per member: at least ~1024 workgroups, slices of a multiple of 512 rows
  int S = 1;
  while ((int64_t)B * S < 1024 && (N + S - 1) / S > 1024) S *= 2;
  int rows = (int)(((N + S - 1) / S + 511) / 512 * 512);
  S = (int)((N + rows - 1) / rows);
  return Split{S, rows};
}

size_t rspace_cols_ws_doubles(int64_t B, int64_t N, int RC, int c) {
  const Split sp = rspace_cols_split(B, N);
  return (size_t)B * sp.S * RC * 2 * c + (size_t)B * sp.S * 2 * c + (size_t)B * c * (RC + 1) + 64;
}

bool rspace_cols_eligible(int RC, int64_t N, int64_t c) {
  return (RC == 8 || RC == 16 || RC == 32) && c >= 1 && c <= RSP_CMAX && N >= 256;
}

template <int RC>
static int rspace_cols_go(const OnchipArgs& a, double* ws, hipStream_t st) {
  const Split sp = rspace_cols_split(a.B, a.N);
  RsColsArgs r;
  r.C = a.C; r.rhs = a.rhs; r.dinv = a.dinv; r.dinv_mode = a.dinv_mode; r.RS = a.RS;
  r.B = a.B; r.N = a.N; r.c = a.c; r.S = sp.S; r.rows_per = sp.rows;
  r.iters = a.iters; r.eps = a.eps; r.stop_after = a.stop_after;
  r.part = ws;
  r.sq = r.part + (size_t)a.B * sp.S * RC * 2 * a.c;
  r.coef = r.sq + (size_t)a.B * sp.S * 2 * a.c;
  r.xout = a.xout; r.ab_rec = a.ab_rec; r.resid_rec = a.resid_rec; r.init_conv = a.init_conv;
  r.rhs_norm = a.rhs_norm; r.rz = a.rz; r.alpha = a.alpha; r.beta = a.beta; r.resid_norm = a.resid_norm;
  r.rhs_is_zero = a.rhs_is_zero; r.has_conv = a.has_conv;
  dim3 block(kThreads);
  // the last column on the vector ALU when it alone would open another 16-column tile (c = 9, 17, 25: 2 c = 16 k + 2)
  const bool vl = a.c > 1 && (2 * a.c - 2 + 15) / 16 < (2 * a.c + 15) / 16 && !getenv("LO_RS_NO_VL");
  const int nt = vl ? (2 * a.c - 2 + 15) / 16 : (2 * a.c + 15) / 16;
  LO_PROF_BEGIN("rs_part", st);
  dim3 gp(sp.S, (unsigned)a.B);
#define LO_RSP(NT_, VL_) \
  hipLaunchKernelGGL((k_rs_part<RC, NT_, VL_>), gp, block, 0, st, r.C, r.rhs, r.dinv, r.dinv_mode, r.N, r.c, r.rows_per, r.part, r.sq)
  if (vl) {
    if (nt == 1) LO_RSP(1, true);
    else if (nt == 2) LO_RSP(2, true);
    else LO_RSP(3, true);
  } else {
    if (nt == 1) LO_RSP(1, false);
    else if (nt == 2) LO_RSP(2, false);
    else if (nt == 3) LO_RSP(3, false);
    else LO_RSP(4, false);
  }
#undef LO_RSP
  LO_PROF_END(st);
  LO_PROF_BEGIN("rs_iter", st);
  hipLaunchKernelGGL((k_rs_iter<RC>), dim3((unsigned)((a.c + 3) / 4), (unsigned)a.B), block, 0, st, r);
  LO_PROF_END(st);
  LO_PROF_BEGIN("rs_apply", st);
  if (a.c <= 17) hipLaunchKernelGGL((k_rs_apply<RC, 2>), gp, block, 0, st, r);
  else hipLaunchKernelGGL((k_rs_apply<RC, 1>), gp, block, 0, st, r);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// All c columns of a result-only solve (a.x == nullptr) with the R-space form at hand; ws: rspace_cols_ws_doubles.
int rspace_cols_launch(int RC, const OnchipArgs& a, double* ws, hipStream_t st) {
  if (!a.RS || a.x || !a.xout || !ws || !rspace_cols_eligible(RC, a.N, a.c)) return LO_ERR_UNSUPPORTED;
  if (RC == 32) return rspace_cols_go<32>(a, ws, st);
  if (RC == 16) return rspace_cols_go<16>(a, ws, st);
  return rspace_cols_go<8>(a, ws, st);
}

// (worst case over the group sizes: groups of one)
size_t rspace_gbuf_bytes(int nworkgroups) { return (size_t)nworkgroups * rs_group_granules(1) * sizeof(unsigned long long); }

bool rspace_eligible(int RC, int64_t N, int64_t c) {
  return (RC == 8 || RC == 16 || RC == 32) && c == 1 && N >= 256 && N <= (int64_t)64 * R4_ROWS;
}

thread_local bool tls_rspace_resident_ran = false;

thread_local bool tls_rspace_diag_ran = false;

template <int RC, int GW, bool DG>
static int rspace_go(const OnchipArgs& a, int nwg, hipStream_t st) {
  int per_cu = 0;
  if (LO_OCCUPANCY_CACHED(per_cu, (k_cg_rspace<RC, GW, DG>), R4_TPB, 0) != hipSuccess || per_cu < 2)
    return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_onchip", st);  // (one scope for the resident single-column kernels: lo_cg_last_executed().rspace tells them apart)
  ResidentLaunch guard(st);
  OnchipArgs a2 = a;
  {  // wave-priority mask of the latency-critical phases (1: iterations, 2: x pass as well, 4: all-reduce as well)
    const char* e = getenv("LO_RS_PRIO");
    a2.prefetch = e ? atoi(e) : 1;
  }
  hipLaunchKernelGGL((k_cg_rspace<RC, GW, DG>), dim3(2 * nwg), dim3(R4_TPB), 0, st, a2);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  tls_rspace_resident_ran = true;
  tls_rspace_diag_ran = DG;
  return LO_OK;
}

// One column, result only (a.x == nullptr), a.F / a.RS present.  Same launch geometry as onchip5_launch.
int rspace_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st) {
  if (!a.RS || a.x || a.c != 1 || a.ab_rec || !a.xout) return LO_ERR_UNSUPPORTED;
  const bool dg = a.RSD != nullptr && !getenv("LO_RS_NO_DIAG");  // the diagonal form (lo_eigform.hip) when the cache carries it
  if (dg) {  // ... in the chunk-per-lane layout (lo_rspace3.hip) for the groups it takes
    const int rc3 = rspace3_launch(RC, a, nwg, st);
    if (rc3 != LO_ERR_UNSUPPORTED) {
      if (rc3 == LO_OK) {
        tls_rspace_resident_ran = true;
        tls_rspace_diag_ran = true;
      }
      return rc3;
    }
  }
  // (the diagonal form of groups of up to 32 workgroups is k_cg_rspace3's: k_cg_rspace<.., true> is instantiated for the
  //  groups of 64 only -- members of 32768 < N <= 65536 rows)
#define LO_RS0(C_, G_) return rspace_go<C_, G_, false>(a, nwg, st)
#define LO_RS1(C_, G_) return dg ? rspace_go<C_, G_, true>(a, nwg, st) : rspace_go<C_, G_, false>(a, nwg, st)
#define LO_RS(C_)                                                                                    \
  switch (a.GW) {                                                                                    \
    case 1: LO_RS0(C_, 1);                                                                           \
    case 2: LO_RS0(C_, 2);                                                                           \
    case 4: LO_RS0(C_, 4);                                                                           \
    case 8: LO_RS0(C_, 8);                                                                           \
    case 16: LO_RS0(C_, 16);                                                                         \
    case 32: LO_RS0(C_, 32);                                                                         \
    default: LO_RS1(C_, 64);                                                                         \
  }
  if (RC == 32) {
    LO_RS(32);
  } else if (RC == 16) {
    LO_RS(16);
  } else if (RC == 8) {
    LO_RS(8);
  }
#undef LO_RS
#undef LO_RS1
#undef LO_RS0
  return LO_ERR_UNSUPPORTED;
}

}  // namespace lo

// lo_cg_onchip4.hip -- second generation of the operator-resident preconditioned CG (see lo_cg_onchip.hip for the
// algorithm, the granule hand-off and the reference citations; the arithmetic is identical).
//
// What limits the first generation is not HBM and not the hand-off latency but VALU issue: with one operator row
// per thread, the per-thread overhead of every inner product (wave reduce-scatter, scalar butterflies, ~250
// instructions) is paid once per ROW, a member occupies 8 CUs, and while a group waits for a hand-off (two per
// iteration, ~1 us each) its CUs idle.  Here a thread owns FOUR rows and TWO workgroups share a CU:
//   * workgroup = 256 threads x 4 rows = 1024 rows; a member of N <= 8192 rows is a group of 8 workgroups but only
//     4 CUs' worth of resources, so 64 members are in flight instead of 32 (groups of 16 / 32 for N <= 16384 /
//     32768); the two workgroups resident on a CU belong to different members, so one computes while the other
//     waits for its hand-off;
//   * the 4 C rows of a thread live in VGPRs (4 x 32 floats -- the register file of a CU is 512 KiB, four times
//     its LDS), the 4 Q rows in LDS (1024 x 16 floats = 64 KiB per workgroup, 16-byte slots XOR-swizzled so that
//     the ds_read_b128 of 16 consecutive rows is bank-conflict free without padding);
//   * the products of the 4 rows are summed in registers BEFORE the wave reduce-scatter, so the reduction cost
//     per row drops 4x;
//   * one all-reduce = wave reduce-scatter -> LDS -> barrier -> the first wave sums the 4 wave partials, publishes
//     the granules, polls the whole group's granules and sums them -> LDS -> barrier (2 barriers instead of 4).
// (512 threads x 2 rows per thread -- 4 waves per SIMD to hide the reduction latency -- was measured 10 % slower:
// the per-thread reduction overhead doubles and the 128-VGPR budget spills.)
// Requires 2 resident workgroups per CU (256 VGPRs per lane, 66 KiB LDS each): checked on the host with the
// occupancy API, otherwise the first generation runs.
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"
#include "lo_group_reduce.h"
#include "lo_cg_close.h"

namespace lo {

// MC: several right-hand-side columns and / or recorded alpha, beta (the single-column instantiation keeps the
// column count a compile-time 1)
template <int RC, int RK, int GW, bool MC>
__global__ __launch_bounds__(R4_TPB, 2 * R4_TPB / 256) void k_cg_onchip4(OnchipArgs a) {
  constexpr int NQ = RK / 4;
  __shared__ R4Shared sh;
  __shared__ float4 q_s[R4_ROWS * NQ];  // Q rows of this workgroup, swizzled 16-byte slots
  __shared__ float x_s[R4_ROWS], d_s[R4_ROWS], dinv_s[R4_ROWS];  // per-row x, d, 1/d (VGPR budget: 256 with 2 WGs per CU)
  constexpr int gw = GW;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;  // block b runs on XCD b % 8: keep a group behind one L2 (speed only)
  const int groups_per_xcd = (gridDim.x / 8) / gw;
  const int grp = xcd * groups_per_xcd + jx / gw;
  const int wig = jx % gw;
  const int ngroups = groups_per_xcd * 8;
  if (jx / gw >= groups_per_xcd) return;
  const int t = threadIdx.x;
  R4Group g;
  g.gslot = a.gbuf + (size_t)grp * 2 * gw * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  g.tag = 0;
  g.err = a.err;
  g.same_xcd = false;
  {  // placement check through the agent-scope path: plain-store hand-off only when the whole group shares an XCD;
     // sum and sum of squares of the XCC ids agree with gw * id and gw * id^2 only if all ids are equal
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t < 64) {
      sh.red[0][0] = (float)xcc;
      sh.red[0][1] = (float)(xcc * xcc);
    }
    if (t < 2 * (R4_WAVES - 1)) sh.red[1 + t / 2][t % 2] = 0.f;
    r4_group_sum<GW>(sh, 2, g);
    const float fx = (float)xcc;
    g.same_xcd = (sh.res[0] == gw * fx) && (sh.res[1] == gw * fx * fx) && (a.allow_l2_handoff != 0);
    __syncthreads();
  }

  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));  // rows of this workgroup
  // Members are handed out dynamically: a group takes member `grp` first and then draws the next index from a
  // shared counter.  Groups that start late (their workgroups waited for a CU that another kernel -- e.g. the RCCL
  // all-gather of the previous solve -- was using) simply take fewer members instead of delaying the whole launch.
  int64_t b = grp;
  while (b < a.B) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    // ---- load: C rows -> VGPRs (each thread walks its own 4 rows; the rows were pulled into L2 by the previous
    // member's prefetch), Q rows -> LDS (coalesced, swizzled), x / d / 1/d -> LDS ----
    float Cr[R4_NR][RC];
    bool valid[R4_NR];
    // thread index the optimiser cannot see through: keeps the address arithmetic of the load phase from being
    // hoisted out of the member loop (it would stay live in VGPRs during the iterations and force spills)
    int tl = t;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const int lr = tl + R4_TPB * q;
      valid[q] = lr < nv;
      const size_t grow = (size_t)b * a.N + row0 + lr;
      float dq = 0.f, diq = 0.f;
      if (valid[q]) {
        const float4* cp = reinterpret_cast<const float4*>(a.C + grow * RC);
#pragma unroll
        for (int i = 0; i < RC / 4; ++i) {
          const float4 c4 = cp[i];
          Cr[q][4 * i] = c4.x; Cr[q][4 * i + 1] = c4.y; Cr[q][4 * i + 2] = c4.z; Cr[q][4 * i + 3] = c4.w;
        }
        dq = (a.d_mode == LO_DIAG_FULL) ? a.d[grow] : (a.d_mode == LO_DIAG_CONST ? a.d[b] : 0.f);
        diq = (a.dinv_mode == LO_DIAG_FULL) ? a.dinv[grow] : a.dinv[b];
      } else {
#pragma unroll
        for (int i = 0; i < RC; ++i) Cr[q][i] = 0.f;
      }
      d_s[lr] = dq;
      dinv_s[lr] = diq;
    }
    {  // Q rows -> LDS, coalesced and swizzled, QB float4 per thread in flight at a time (next to the loads of C that
       // are still landing in their 128 registers; nothing else is live in this phase)
      const float4* qsrc = reinterpret_cast<const float4*>(a.Q + ((size_t)b * a.N + row0) * RK);
      constexpr int QB = (R4_NR * NQ >= 8) ? 8 : R4_NR * NQ;  // (16 at once measured slower)
#pragma unroll
      for (int i0 = 0; i0 < R4_NR * NQ; i0 += QB) {
        float4 v[QB];
#pragma unroll
        for (int i = 0; i < QB; ++i) {
          const int f = (i0 + i) * R4_TPB + tl;
          v[i] = (f / NQ < nv) ? qsrc[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
          const int f = (i0 + i) * R4_TPB + tl;
          q_s[q_slot<NQ>(f / NQ, f % NQ)] = v[i];
        }
      }
    }
    __syncthreads();
    if (stamp) a.dbg[1] = wall_clock64();

    // The right-hand-side columns are solved one after the other against the resident rows (each column is an
    // independent recurrence in the reference: every scalar of linear_cg.py carries a trailing column dimension).
    const int nc = MC ? a.c : 1;  // row stride of the vectors; this launch solves columns [col0, col0 + ncols)
    const int cfirst = MC ? a.col0 : 0, clast = MC ? a.col0 + a.ncols : 1;
    for (int col = cfirst; col < clast; ++col) {
    const size_t bc = (size_t)b * nc + col;
    // ---- initialisation (linear_cg.py:177-215) ----
    float sc[2];
    float r[R4_NR], p[R4_NR], z[R4_NR];
    sc[0] = 0.f;
    int tc = t;  // (opaque, like tl: the addresses of the column's loads must not live across the iterations)
    asm volatile("" : "+v"(tc));
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const int lr = tc + R4_TPB * q;
      r[q] = (lr < nv) ? a.rhs[((size_t)b * a.N + row0 + lr) * nc + col] : 0.f;
      x_s[lr] = 0.f;
      sc[0] = fmaf(r[q], r[q], sc[0]);
    }
    r4_allreduce_scalars<GW>(sh, sc, 1, g);
    float nrm = sqrtf(sh.res[0]);                           // rhs.norm(2, dim=-2)          :177
    const bool rhs_zero = nrm < a.eps;                      // :178
    if (rhs_zero) nrm = 1.0f;                               // :179
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) r[q] = r[q] / nrm;      // :182 (x0 = 0 -> residual = rhs)
    // Q^T r, ||r||^2, sum r^2/d; z = r/d - Q (Q^T r)  (precondition_closure :135-140)
    float rr, rz, rn;
    bool conv;
    auto precond = [&](float& rz_out) {
      float u[RK];
#pragma unroll
      for (int j = 0; j < RK; ++j) u[j] = 0.f;
      float s2[2] = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = t + R4_TPB * q;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const float4 q4 = q_s[q_slot<NQ>(lr, i)];
          u[4 * i] = fmaf(q4.x, r[q], u[4 * i]);
          u[4 * i + 1] = fmaf(q4.y, r[q], u[4 * i + 1]);
          u[4 * i + 2] = fmaf(q4.z, r[q], u[4 * i + 2]);
          u[4 * i + 3] = fmaf(q4.w, r[q], u[4 * i + 3]);
        }
        s2[0] = fmaf(r[q], r[q], s2[0]);
        s2[1] = fmaf(dinv_s[lr] * r[q], r[q], s2[1]);
      }
      r4_allreduce<GW, RK>(sh, [&](int j) { return u[j]; }, s2, 2, g);
      rr = sh.res[RK];
      float uu = 0.f;
#pragma unroll
      for (int j = 0; j < RK; ++j) uu = fmaf(sh.res[j], sh.res[j], uu);
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = t + R4_TPB * q;
        float zq = dinv_s[lr] * r[q];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const float4 q4 = q_s[q_slot<NQ>(lr, i)];
          const float4 u4 = *reinterpret_cast<const float4*>(&sh.res[4 * i]);
          zq = fmaf(-q4.x, u4.x, zq);
          zq = fmaf(-q4.y, u4.y, zq);
          zq = fmaf(-q4.z, u4.z, zq);
          zq = fmaf(-q4.w, u4.w, zq);
        }
        z[q] = zq;
      }
      // r.z = sum r o (r/d - Q u) = sum r^2/d - ||Q^T r||^2   (residual_inner_prod :215 / :35-36)
      rz_out = sh.res[RK + 1] - uu;
    };
    precond(rz);
    conv = sqrtf(rr) < a.stop_after;                        // :204-205
    if (wig == 0 && t == 0) a.init_conv[bc] = conv ? 1 : 0;
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) p[q] = z[q];
    float beta = 0.f, alpha = 0.f;
    rn = sqrtf(rr);

    if (stamp && col == cfirst) {
      a.dbg[2] = wall_clock64();
      g.dbg = a.dbg;
    }
    for (int k = 0; k < a.iters; ++k) {
      if (k > 0) {
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) p[q] = fmaf(p[q], beta, z[q]);  // p.mul_(beta).add_(z)  :46
      }
      {  // t = C^T p and sum d p^2
        sc[0] = 0.f;
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) sc[0] = fmaf(d_s[t + R4_TPB * q] * p[q], p[q], sc[0]);
        r4_allreduce<GW, RC>(
            sh,
            [&](int c) {
              float v = Cr[0][c] * p[0];
#pragma unroll
              for (int q = 1; q < R4_NR; ++q) v = fmaf(Cr[q][c], p[q], v);
              return v;
            },
            sc, 1, g);
      }
      // A p = C t + d o p (added_diag_linear_operator.py:72-76); p.Ap = ||C^T p||^2 + sum d p^2 (:250-251)
      float y[R4_NR];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) y[q] = d_s[t + R4_TPB * q] * p[q];
      float tt = 0.f;
#pragma unroll
      for (int i = 0; i < RC; i += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(&sh.res[i]);
        tt = fmaf(t4.x, t4.x, tt);
        tt = fmaf(t4.y, t4.y, tt);
        tt = fmaf(t4.z, t4.z, tt);
        tt = fmaf(t4.w, t4.w, tt);
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) {
          y[q] = fmaf(Cr[q][i], t4.x, y[q]);
          y[q] = fmaf(Cr[q][i + 1], t4.y, y[q]);
          y[q] = fmaf(Cr[q][i + 2], t4.z, y[q]);
          y[q] = fmaf(Cr[q][i + 3], t4.w, y[q]);
        }
      }
      const float pAp = tt + sh.res[RC];
      alpha = (pAp < a.eps) ? 0.f : rz / pAp;               // :254-257
      if (conv) alpha = 0.f;                                // :260
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        r[q] = fmaf(-alpha, y[q], r[q]);                    // :264
        x_s[t + R4_TPB * q] = fmaf(alpha, p[q], x_s[t + R4_TPB * q]);  // :31
      }
      float rzn;
      precond(rzn);                                         // z = P^{-1} r, ||r||^2, r.z
      beta = (rz < a.eps) ? 0.f : rzn / rz;                 // :39-42
      rz = rzn;
      rn = sqrtf(rr);                                       // :298
      if (rhs_zero) rn = 0.f;                               // :299
      conv = rn < a.stop_after;                             // :300
      if (wig == 0 && t == 0) {
        a.resid_rec[(size_t)k * a.B * nc + bc] = rn;
        if (MC && a.ab_rec) {  // masked alpha and beta of this iteration: the tridiagonal recurrence is replayed afterwards
          a.ab_rec[2 * ((size_t)k * a.B * nc + bc)] = alpha;
          a.ab_rec[2 * ((size_t)k * a.B * nc + bc) + 1] = beta;
        }
      }
    }

    if (stamp && col == cfirst) a.dbg[3] = wall_clock64();
    g.dbg = nullptr;
    // ---- write the state back in the streaming engine's layout ----
    int tw = t;
    asm volatile("" : "+v"(tw));
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      if (tw + R4_TPB * q < nv) {
        const size_t o = ((size_t)b * a.N + row0 + tw + R4_TPB * q) * nc + col;
        a.x[o] = x_s[t + R4_TPB * q];
        if (a.xout) a.xout[o] = x_s[t + R4_TPB * q] * nrm;  // final when the stop rule holds at the floor (:335)
        a.r[o] = r[q];
        a.p[o] = p[q];
        if (a.z) a.z[o] = z[q];  // no z buffer in the unpreconditioned engine (z = r)
      }
    }
    if (wig == 0 && t == 0) {
      a.rhs_norm[bc] = nrm;
      a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
      a.rz[bc] = rz;
      a.alpha[bc] = alpha;
      a.beta[bc] = beta;
      a.resid_norm[bc] = rn;
      a.has_conv[bc] = conv ? 1 : 0;
    }
    }  // columns
    __syncthreads();  // q_s / sh reuse by the next member
    if (stamp) a.dbg[4] = wall_clock64();
    // next member: drawn by the group's first workgroup, handed to the others through the all-reduce path
    // (one contributor, the rest add zeros: exact for indices < 2^24)
    if (t < R4_WAVES) sh.red[t][0] = 0.f;
    __syncthreads();
    if (wig == 0 && t == 0) sh.red[0][0] = (float)(ngroups + atomicAdd(a.next_member, 1));
    r4_group_sum<GW>(sh, 1, g);
    b = (int64_t)sh.res[0];
    __syncthreads();
  }
}

bool onchip4_eligible(int RC, int RK, int64_t N, int64_t c) {
  const bool rc_ok = (RC == 8 || RC == 16 || RC == 32);
  const bool rk_ok = (RK == 4 || RK == 8 || RK == 16);
  return rc_ok && rk_ok && c >= 1 && c <= 64 && N >= 1024 && N <= (int64_t)R4_MAXGW * R4_ROWS;
}

int onchip4_group_size(int64_t N) { return N <= 8 * (int64_t)R4_ROWS ? 8 : (N <= 16 * (int64_t)R4_ROWS ? 16 : 32); }
// Root-form kernel: the smallest group that holds the member (1024 rows per workgroup).  Small members used to take 8
// workgroups with mostly idle threads: N = 1024 .. 4096 ran at the N = 8192 time per solve; with 1 / 2 / 4 workgroups
// 8 / 4 / 2 x more members are resident at a time (and a group of one needs no hand-off at all).
int onchip5_group_size(int64_t N) {
  for (int gw = 1; gw < 64; gw *= 2)
    if (N <= (int64_t)gw * R4_ROWS) return gw;
  return 64;  // (up to 65536 rows: 8 members resident at a time, two-hop lane-parallel all-reduce)
}

template <int RC, int RK, int GW, bool MC>
static int onchip4_go(const OnchipArgs& a, int nwg, hipStream_t st) {
  // the spin-waiting groups need ALL workgroups resident: two per CU
  int per_cu = 0;
  const bool half = getenv("LO_OC_HALF") != nullptr;  // debugging: one workgroup per CU
  if (LO_OCCUPANCY_CACHED(per_cu, (k_cg_onchip4<RC, RK, GW, MC>), R4_TPB, 0) != hipSuccess ||
      per_cu < (half ? 1 : 2))
    return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_onchip", st);
  ResidentLaunch guard(st);
  hipLaunchKernelGGL((k_cg_onchip4<RC, RK, GW, MC>), dim3(half ? nwg : 2 * nwg), dim3(R4_TPB), 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// nwg = number of CUs used (multiple of 64); 2 * nwg workgroups are launched.  LO_ERR_UNSUPPORTED when two
// workgroups do not fit on a CU (the caller then runs the first generation).
int onchip4_launch(int RC, int RK, const OnchipArgs& a, int nwg, hipStream_t st) {
  const bool mc = a.c > 1 || a.ab_rec != nullptr;
#define LO_OC_G(C_, K_, G_) (mc ? onchip4_go<C_, K_, G_, true>(a, nwg, st) : onchip4_go<C_, K_, G_, false>(a, nwg, st))
#define LO_OC(C_, K_) return a.GW == 8 ? LO_OC_G(C_, K_, 8) : (a.GW == 16 ? LO_OC_G(C_, K_, 16) : LO_OC_G(C_, K_, 32))
  if (RC == 32 && RK == 16) LO_OC(32, 16);
  else if (RC == 32 && RK == 8) LO_OC(32, 8);
  else if (RC == 32 && RK == 4) LO_OC(32, 4);
  else if (RC == 16 && RK == 16) LO_OC(16, 16);
  else if (RC == 16 && RK == 8) LO_OC(16, 8);
  else if (RC == 16 && RK == 4) LO_OC(16, 4);
  else if (RC == 8 && RK == 16) LO_OC(8, 16);
  else if (RC == 8 && RK == 8) LO_OC(8, 8);
  else if (RC == 8 && RK == 4) LO_OC(8, 4);
#undef LO_OC
#undef LO_OC_G
  return LO_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// Third generation of the serial-column kernel: ROOT-FORM preconditioner (lo_precond_desc.F / EF) and ONE group
// all-reduce per iteration.  With P^-1 r = (r - C F w) / d, w = C^T (r / d), the iteration needs no second tall matrix:
// the reduction delivers w (RC values), s1 = sum r^2, s2 = sum r^2 / d, rp = sum r o p_old, and the first wave derives
//     v = F w,  E v = (E F) w,  r.z = s2 - w.v,  C^T p_new = (w - E v) + beta C^T p_old,
//     sum d p_new^2 = (s2 - 2 w.v + v.E v) + 2 beta (rp - v . C^T p_old) + beta^2 sum d p_old^2,   p.Ap = |C^T p|^2 + sum d p^2
// (tests/proto/proto_root_form.py) from two RC x RC matrices in LDS before the others leave the all-reduce.  Per row the work
// is three passes over the thread's C rows in VGPRs (z = (r - C v) / d, A p = C t + d p, the partials of w): the same
// 12 RC FMAs per row as before, half the hand-offs.  Without a preconditioner F = 0: v = 0, z = r / 1.
struct alignas(16) R5Post {
  float v[32];
  float t[32];
  float w[32];  // (w-recurrence mode: this wave's copy of w = C^T (r / d) for the broadcast reads of the next F w)
  float alpha, beta, rn, pad;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: one column, three passes over C per iteration.  MODE 1 ("MC"): several columns and / or recorded alpha, beta.
// MODE 2 ("WR"): one column, no tridiagonals, w = C^T D^-1 r carried by RECURRENCE -- r' = r - alpha (C t + d o p) with
// t = C^T p gives  w' = w - alpha (E t + t),  E = C^T D^-1 C  (tests/proto/proto_w_recurrence.py): the third pass over the
// rows of C and 32 of the 35 values of the per-iteration all-reduce disappear; the rows still deliver the three scalars
// {sum r^2, sum r^2 / d, sum r o p}.  Identical in exact arithmetic; solutions as close to the fp64 iteration as the
// three-pass form (1.5e-6 vs 1.7e-6 at the headline spectrum), but the CG COEFFICIENTS of the converging iterations
// follow the fp64 ones to 1e-3 instead of 5e-6 -- so the mode is never used when tridiagonals are recorded.
template <int RC, int GW, int MODE>
__global__ __launch_bounds__(R4_TPB, 2 * R4_TPB / 256) void k_cg_onchip5(OnchipArgs a) {
  constexpr bool MC = MODE == 1;
  constexpr bool WR = MODE == 2;
  constexpr int FLD = RC + 4;  // LDS row stride of F / EF (16-byte aligned rows, conflict-free float4 reads per lane)
  __shared__ R4Shared sh;
  __shared__ R5Post post[R4_WAVES];
  __shared__ __attribute__((aligned(16))) float f_s[RC * FLD];
  __shared__ __attribute__((aligned(16))) float ef_s[RC * FLD];
  __shared__ __attribute__((aligned(16))) float e_s[WR ? RC * FLD : 4];
  __shared__ float x_s[R4_ROWS], d_s[R4_ROWS], dinv_s[R4_ROWS];
  __shared__ float4 stage_s[R4_WAVES * 64 * (RC / 4)];  // per-wave transposition window of the member load
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63;
  R4Group g;
  g.gslot = a.gbuf + (size_t)grp * 2 * (GW == 64 ? GW + 1 : GW) * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  g.tag = 0;
  g.err = a.err;
  g.same_xcd = false;
  {
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
  const bool pre = a.F != nullptr;
  int64_t b = grp;
  while (b < a.B) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    // the C rows of the thread as register PAIRS: the three passes over them are v_pk_fma_f32 (two columns per
    // instruction; the kernel is bound by vector-ALU issue)
    f32x2 Cr[R4_NR][RC / 2];
    int tl = t;
    asm volatile("" : "+v"(tl));
    // Every load of the member is issued before anything waits (vmcnt counts in order: one early use would serialise
    // the rows): branch-free, the padding rows read a clamped valid row and are zeroed afterwards.
    const bool d_any = a.d_mode != LO_DIAG_NONE, d_full = a.d_mode == LO_DIAG_FULL;
    const bool di_full = a.dinv_mode == LO_DIAG_FULL;
    float dq[R4_NR], diq[R4_NR];
    // C: a wave fetches its 64 consecutive rows as 64 * RC / 4 CONSECUTIVE 16-byte chunks (lane l takes chunk 64 i + l:
    // whole cache lines per instruction instead of 16 bytes out of 64 different lines), parks them in its own LDS
    // window (chunk slot XOR-swizzled by the row) and reads its row back -- no barrier, the window belongs to the wave
    constexpr int CH = RC / 4;
    const int wv = tl >> 6, ln = tl & 63;
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int g = 64 * i + ln;  // chunk of the wave's block
        const int rw = g / CH, ck = g % CH;
        const size_t grow_c = (size_t)b * a.N + min(row0 + R4_TPB * q + 64 * wv + rw, a.N - 1);
        const float4 c4 = *reinterpret_cast<const float4*>(a.C + grow_c * RC + 4 * ck);
        Cr[q][2 * i] = f32x2{c4.x, c4.y}; Cr[q][2 * i + 1] = f32x2{c4.z, c4.w};
      }
    }
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const int lr = tl + R4_TPB * q;
      const size_t grow = (size_t)b * a.N + min(row0 + lr, a.N - 1);
      const float* dp = d_any ? (d_full ? a.d + grow : a.d + b) : a.C;
      const float* ip = pre ? (di_full ? a.dinv + grow : a.dinv + b) : a.C;
      dq[q] = *dp;
      diq[q] = *ip;
    }
    constexpr int NF = (RC * RC + R4_TPB - 1) / R4_TPB;  // F, EF -> LDS (zero without a preconditioner)
    float fv[NF], ev[NF], e0v[WR ? NF : 1];
    {
      const float* Fp = pre ? a.F + (size_t)b * RC * RC : a.C;
      const float* Ep = pre ? a.EF + (size_t)b * RC * RC : a.C;
      const float* E0p = (WR && pre) ? a.E + (size_t)b * RC * RC : a.C;
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int e = min(tl + R4_TPB * u, RC * RC - 1);
        fv[u] = Fp[e];
        ev[u] = Ep[e];
        if (WR) e0v[u] = E0p[e];
      }
    }
#pragma unroll
    for (int q = 0; q < R4_NR; ++q) {
      const int lr = tl + R4_TPB * q;
      const bool valid = lr < nv;
      float4* win = stage_s + wv * (64 * CH);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int g = 64 * i + ln;
        const int rw = g / CH, ck = g % CH;
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
      d_s[lr] = (valid && d_any) ? dq[q] : 0.f;
      dinv_s[lr] = valid ? (pre ? diq[q] : 1.0f) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = tl + R4_TPB * u;
      if (e < RC * RC) {
        const int i = e / RC, j = e % RC;
        f_s[i * FLD + j] = pre ? fv[u] : 0.f;
        ef_s[i * FLD + j] = pre ? ev[u] : 0.f;
        if (WR) e_s[i * FLD + j] = pre ? e0v[u] : 0.f;
      }
    }
    __syncthreads();
    if (stamp) a.dbg[1] = wall_clock64();

    const int nc = MC ? a.c : 1;
    const int cfirst = MC ? a.col0 : 0, clast = MC ? a.col0 + a.ncols : 1;
    int64_t b_next = a.B;  // set by the member's last reduction
    int drawn = 0;
    // the hand-out counter is read one phase ahead of the reduction that carries it: the atomic's round trip hides
    // behind the vector updates instead of delaying the publication of the group's first workgroup
    auto draw = [&](int k, int col) {
      if (k == a.iters - 1 && col == clast - 1 && wig == 0 && t == 0) drawn = atomicAdd(a.next_member, 1);
    };
    for (int col = cfirst; col < clast; ++col) {
      const size_t bc = (size_t)b * nc + col;
      float r[R4_NR], p[R4_NR];
      float sc[4];
      sc[0] = 0.f;
      int tc = t;
      asm volatile("" : "+v"(tc));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q)  // (all four loads in flight together: clamped address, selected below)
        r[q] = a.rhs[((size_t)b * a.N + min(row0 + tc + R4_TPB * q, a.N - 1)) * nc + col];
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        const int lr = tc + R4_TPB * q;
        r[q] = (lr < nv) ? r[q] : 0.f;
        p[q] = 0.f;
        x_s[lr] = 0.f;
      }
      // The norm of the right-hand side rides on the FIRST reduction (its s1 = sum r^2 is the squared norm): that
      // reduction runs on the raw column and its results are scaled afterwards -- w by 1 / norm, s1 and s2 by
      // 1 / norm^2 -- instead of a separate all-reduce in front of it (1.4 us per member and column).  Same range as the
      // norm itself: sum r^2 has to be finite in fp32.
      float nrm = 1.0f, inv0 = 1.0f;
      bool rhs_zero = false;
      // first-wave state of the recurrences (uniform scalars replicated in the lanes; t_old = C^T p_old in lane j < RC)
      float t_old = 0.f, tt_old = 0.f, dpp = 0.f, rz = 0.f, alpha = 0.f, beta = 0.f, rn = 0.f;
      float w_reg = 0.f;  // (WR: w_j in lane j of both halves, carried by the recurrence)
      unsigned close_flags = 0u;  // bit 0: converged before the first iteration, bit 1: NaN after the first product
      bool conv = false;
      // one reduction: w = C^T (r / d), s1, s2, rp; then (first wave) the small algebra; k = -1 marks the initial one
      auto reduce_and_post = [&](int k) {
        sc[0] = 0.f; sc[1] = 0.f; sc[2] = 0.f;
        float rd[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) {
          const int lr = t + R4_TPB * q;
          rd[q] = r[q] * dinv_s[lr];
          sc[0] = fmaf(r[q], r[q], sc[0]);
          if (pre) {
            sc[1] = fmaf(rd[q], r[q], sc[1]);
            sc[2] = fmaf(r[q], p[q], sc[2]);
          } else {  // z = r: the "d" of the recurrences is the operator's diagonal itself
            const float dr = d_s[lr] * r[q];
            sc[1] = fmaf(dr, r[q], sc[1]);
            sc[2] = fmaf(dr, p[q], sc[2]);
          }
        }
        long long cr0 = 0;
        if (g.dbg && t == 0) cr0 = wall_clock64();
        // the member's LAST reduction also carries the next member of the group (dynamic hand-out: drawn by the
        // group's first workgroup, exact below 2^24) -- no all-reduce of its own at the end of the member
        const bool draws = (k == a.iters - 1) && (col == clast - 1);
        sc[3] = (draws && wig == 0 && t == 0) ? (float)(ngroups + drawn) : 0.f;  // (requested ahead, see draw())
        const bool wrec = WR && k >= 0;  // w comes from the recurrence: only the scalars are reduced
        float tot[4] = {0.f, 0.f, 0.f, 0.f};  // the reduced scalars s1 | s2 | rp | (hand-out)
        if (wrec) {
          if constexpr (GW <= 16) {  // one barrier, the totals come back in registers
            if (draws) {  // (two call sites: the scalar count stays a compile-time constant in the common one)
              r4_allreduce_small<GW, 4>(sh, sc, tot, g);
              b_next = (int64_t)tot[3];
            } else {
              r4_allreduce_small<GW, 3>(sh, sc, tot, g);
            }
          } else {
            if (draws) {
              r4_allreduce_scalars4<GW, 4>(sh, sc, g);
              b_next = (int64_t)sh.res[3];
            } else {
              r4_allreduce_scalars4<GW, 3>(sh, sc, g);
            }
            tot[0] = sh.res[0]; tot[1] = sh.res[1]; tot[2] = sh.res[2];
          }
        } else {
          f32x2 wp[RC / 2];  // column partials of w, two columns per instruction
#pragma unroll
          for (int j = 0; j < RC / 2; ++j) {
            f32x2 v = Cr[0][j] * f32x2{rd[0], rd[0]};
#pragma unroll
            for (int q = 1; q < R4_NR; ++q) v = __builtin_elementwise_fma(Cr[q][j], f32x2{rd[q], rd[q]}, v);
            wp[j] = v;
          }
          const auto gen = [&](int c) { return (c & 1) ? wp[c >> 1].y : wp[c >> 1].x; };
          if (draws) {
            r4_allreduce<GW, RC>(sh, gen, sc, 4, g);
            b_next = (int64_t)sh.res[RC + 3];
          } else {
            r4_allreduce<GW, RC>(sh, gen, sc, 3, g);
          }
          tot[0] = sh.res[RC]; tot[1] = sh.res[RC + 1]; tot[2] = sh.res[RC + 2];
        }
        if (g.dbg && t == 0) g.dbg[9] += wall_clock64() - cr0;  // partials of w + reduce-scatter + group all-reduce
        long long cp0 = 0;
        if (g.dbg && t == 0) cp0 = wall_clock64();
        {  // ---- the small algebra, redundantly in every wave (no further barrier): sh.res holds w | s1 | s2 | rp ----
          const int j = lane & 31;
          float mv = 0.f;  // lanes 0-31: (F w)_j, lanes 32-63: (E F w)_j
          if (pre && j < RC) {
            const float* row = (lane < 32 ? f_s : ef_s) + j * FLD;
            const float* wsrc = wrec ? post[t >> 6].w : sh.res;  // (this wave's own copy / the all-reduce's result)
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RC; q += 4) {
              const float4 m4 = *reinterpret_cast<const float4*>(row + q);
              const float4 w4 = *reinterpret_cast<const float4*>(&wsrc[q]);
              a01 = __builtin_elementwise_fma(f32x2{m4.x, m4.y}, f32x2{w4.x, w4.y}, a01);
              a23 = __builtin_elementwise_fma(f32x2{m4.z, m4.w}, f32x2{w4.z, w4.w}, a23);
            }
            mv = (a01.x + a01.y) + (a23.x + a23.y);
          }
          // v_permlane32_swap of a value with itself: first result = the lower-half lane's value in both halves,
          // second = the upper-half lane's: every lane gets (v_j, (E v)_j) with one instruction
          float sc1 = 1.0f, sc2 = 1.0f;  // scaling of the raw first reduction (k < 0), 1 afterwards
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
          const bool own = lane < 32 && j < RC;
          const float s1 = tot[0] * sc2, s2 = tot[1] * sc2, rp = tot[2];
          const float zj = wj - evj;                          // (C^T z)_j
          // five independent sums (their butterflies interleave); |C^T p_new|^2 from the expansion so that it does not
          // wait for beta: |zc + beta t|^2 = |zc|^2 + 2 beta zc.t + beta^2 |t|^2
          // (w, v, E v and C^T p_old are replicated in the two half-waves: each half sums its own 32 lanes)
          const bool live = j < RC;
          const float wv = lanes32_sum(live ? wj * vj : 0.f);
          const float vev = lanes32_sum(live ? vj * evj : 0.f);
          const float vt = lanes32_sum(live ? vj * t_old : 0.f);
          const float zz = lanes32_sum(live ? zj * zj : 0.f);
          const float zt = lanes32_sum(live ? zj * t_old : 0.f);
          const float rzn = pre ? s2 - wv : s1;              // residual_inner_prod :215 / :35-36
          float rnn = __builtin_amdgcn_sqrtf(s1);            // :298 / :204
          if (k >= 0) {                                      // closes iteration k: beta, residual norm, records
            beta = (rz < a.eps) ? 0.f : rzn * __builtin_amdgcn_rcpf(rz);  // :39-42
            if (rhs_zero) rnn = 0.f;                         // :299
            rn = rnn;
            if (wig == 0 && t == 0) {
              a.resid_rec[(size_t)k * a.B * nc + bc] = rn;
              if (MC && a.ab_rec) {
                a.ab_rec[2 * ((size_t)k * a.B * nc + bc)] = alpha;
                a.ab_rec[2 * ((size_t)k * a.B * nc + bc) + 1] = beta;
              }
            }
          } else {
            beta = 0.f;
            rn = rnn;
            if (wig == 0 && t == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
            close_flags = (rn < a.stop_after) ? 1u : 0u;
          }
          if (k == 0 && rnn != rnn) close_flags |= 2u;  // (NaN after the first product, linear_cg.py:199-200)
          conv = rn < a.stop_after;                          // :300
          rz = rzn;
          const float dzz = pre ? fmaf(-2.f, wv, s2) + vev : s2;
          const float dzp = rp - vt;
          dpp = fmaf(beta, fmaf(beta, dpp, 2.f * dzp), dzz);
          tt_old = fmaf(beta, fmaf(beta, tt_old, 2.f * zt), zz);  // |C^T p_new|^2
          t_old = fmaf(beta, t_old, zj);                     // C^T p_new (lanes j < RC of both halves)
          const float pAp = tt_old + dpp;
          alpha = (pAp < a.eps) ? 0.f : rz * __builtin_amdgcn_rcpf(pAp);  // :254-257
          if (conv) alpha = 0.f;                             // :260
          R5Post& mine = post[t >> 6];                       // this wave's own copy: written and read by the same wave
          if (own) {
            mine.v[j] = vj;
            mine.t[j] = t_old;
          }
          if (WR) {
            // w of the NEXT reduction by recurrence: r' = r - alpha (C t + d o p)  =>  w' = w - alpha (E t + t).
            // (E t)_j: each half-wave sums half of row j of E against the t this wave has just written (LDS operations of
            // a wave execute in order), the halves are joined by one v_permlane32_swap
            __builtin_amdgcn_wave_barrier();
            float et = 0.f;
            if (pre && j < RC) {
              const int q0 = (lane < 32) ? 0 : RC / 2;
              const float* row = e_s + j * FLD + q0;
              f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
              for (int q = 0; q < RC / 2; q += 4) {
                const float4 m4 = *reinterpret_cast<const float4*>(row + q);
                const float4 t4 = *reinterpret_cast<const float4*>(&mine.t[q0 + q]);
                a01 = __builtin_elementwise_fma(f32x2{m4.x, m4.y}, f32x2{t4.x, t4.y}, a01);
                a23 = __builtin_elementwise_fma(f32x2{m4.z, m4.w}, f32x2{t4.z, t4.w}, a23);
              }
              et = (a01.x + a01.y) + (a23.x + a23.y);
            }
            const auto se = __builtin_amdgcn_permlane32_swap(__float_as_uint(et), __float_as_uint(et), false, false);
            const float etj = __uint_as_float(se[0]) + __uint_as_float(se[1]);
            w_reg = fmaf(-alpha, etj + t_old, wj);
            if (own) mine.w[j] = w_reg;
          }
        }
        if (g.dbg && t == 0) g.dbg[10] += wall_clock64() - cp0;  // small algebra
      };
      draw(-1, col);
      reduce_and_post(-1);
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) r[q] = r[q] / nrm;     // :182 (the reduction above saw the raw column)
      if (stamp && col == cfirst) {
        a.dbg[2] = wall_clock64();
        g.dbg = a.dbg;
      }
      float last_alpha = 0.f;
      for (int k = 0; k < a.iters; ++k) {
        long long c0 = 0;
        if (g.dbg && t == 0) c0 = wall_clock64();
        draw(k, col);
        const float al = alpha, be = beta;  // (identical in every wave: formed from the same all-reduced values)
        const R5Post& mine = post[t >> 6];
        last_alpha = al;
        // p = beta p + (r - C v) / d  (:268, :46);  x += alpha p (:31);  r -= alpha (C t + d p) (:264)
        f32x2 cv2[R4_NR], y2[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) { cv2[q] = f32x2{0.f, 0.f}; y2[q] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < RC; i += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(&mine.v[i]);
          const float4 t4 = *reinterpret_cast<const float4*>(&mine.t[i]);
          const f32x2 va{v4.x, v4.y}, vb{v4.z, v4.w}, ta{t4.x, t4.y}, tb{t4.z, t4.w};
#pragma unroll
          for (int q = 0; q < R4_NR; ++q) {
            cv2[q] = __builtin_elementwise_fma(Cr[q][i / 2], va, cv2[q]);
            cv2[q] = __builtin_elementwise_fma(Cr[q][i / 2 + 1], vb, cv2[q]);
            y2[q] = __builtin_elementwise_fma(Cr[q][i / 2], ta, y2[q]);
            y2[q] = __builtin_elementwise_fma(Cr[q][i / 2 + 1], tb, y2[q]);
          }
        }
        float cv[R4_NR], y[R4_NR];
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) { cv[q] = cv2[q].x + cv2[q].y; y[q] = y2[q].x + y2[q].y; }
#pragma unroll
        for (int q = 0; q < R4_NR; ++q) {
          const int lr = t + R4_TPB * q;
          p[q] = fmaf(be, p[q], (r[q] - cv[q]) * dinv_s[lr]);
          x_s[lr] = fmaf(al, p[q], x_s[lr]);
          r[q] = fmaf(-al, fmaf(d_s[lr], p[q], y[q]), r[q]);
        }
        if (g.dbg && t == 0) g.dbg[8] += wall_clock64() - c0;  // vector updates (three passes over the C rows incl. below)
        reduce_and_post(k);
      }
      if (stamp && col == cfirst) a.dbg[3] = wall_clock64();
      g.dbg = nullptr;
      // ---- write the state back in the streaming engine's layout (z = (r - C v) / d from the last reduction) ----
      int tw = t;
      asm volatile("" : "+v"(tw));
#pragma unroll
      for (int q = 0; q < R4_NR; ++q) {
        if (tw + R4_TPB * q < nv) {
          const size_t o = ((size_t)b * a.N + row0 + tw + R4_TPB * q) * nc + col;
          const int lr = t + R4_TPB * q;
          // (a.x == nullptr: the caller expects the stop rule to hold at the floor and wants the result only; if it
          //  does not hold, the launch is repeated WITH the continuation state -- lo_cg.hip)
          if (a.xout) a.xout[o] = x_s[lr] * nrm;             // :335
          if (a.x) {
            a.x[o] = x_s[lr];
            a.r[o] = r[q];
            a.p[o] = p[q];
          }
          if (a.x && a.z) {
            float cvq = 0.f;
#pragma unroll
            for (int i = 0; i < RC; ++i) cvq = fmaf((i & 1) ? Cr[q][i >> 1].y : Cr[q][i >> 1].x, post[t >> 6].v[i], cvq);
            a.z[o] = (r[q] - cvq) * dinv_s[lr];
          }
        }
      }
      if (wig == 0 && t == 0) {  // (thread 0 is a first-wave lane: its copies of the scalars are the current ones)
        a.rhs_norm[bc] = nrm;
        a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
        a.rz[bc] = rz;
        a.alpha[bc] = last_alpha;
        a.beta[bc] = beta;
        a.resid_norm[bc] = rn;
        a.has_conv[bc] = conv ? 1 : 0;
        if (!MC && a.close_gran) {  // this member's line of the closing step: one never-torn 8-byte store
          const unsigned long long gr =
              ((unsigned long long)(0x80000000u | close_flags) << 32) | (unsigned long long)__float_as_uint(rn);
          __hip_atomic_store(a.close_gran + b, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
    }  // columns
    if (stamp) a.dbg[4] = wall_clock64();
    b = b_next;
  }
  // ---- closing step: the first workgroup of the group that finishes LAST does what k_cg_ctrl_onchip does ----
  if (!MC && a.close_gran && wig == 0) cg_close_solve(a, ngroups, t);
}

bool onchip5_eligible(int RC, int64_t N, int64_t c) {
  return (RC == 8 || RC == 16 || RC == 32) && c >= 1 && c <= 64 && N >= 256 && N <= (int64_t)64 * R4_ROWS;
}

template <int RC, int GW, int MODE>
static int onchip5_go(const OnchipArgs& a, int nwg, hipStream_t st) {
  int per_cu = 0;
  if (LO_OCCUPANCY_CACHED(per_cu, (k_cg_onchip5<RC, GW, MODE>), R4_TPB, 0) != hipSuccess ||
      per_cu < 2)
    return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_onchip", st);
  ResidentLaunch guard(st);
  hipLaunchKernelGGL((k_cg_onchip5<RC, GW, MODE>), dim3(2 * nwg), dim3(R4_TPB), 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// a.F / a.EF [B, RC, RC] (or nullptr: no preconditioner).  Same launch geometry as onchip4_launch.
int onchip5_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st) {
  const bool mc = a.c > 1 || a.ab_rec != nullptr;
  // one column without recorded coefficients and the preconditioner's E = C^T D^-1 C at hand: w by recurrence
  // (LO_OC_NO_WREC restores the three-pass iteration; the tests compare the two)
  // ONLY for the result-only first pass (a.x == nullptr): the recurrence is as accurate as the three-pass iteration
  // while CG converges, but on ill-conditioned systems (diagonals ~1e-3: E ~ 1e5) w stagnates at 1e-4 of its start --
  // such solves miss the stop rule at the floor (evaluated on sum r^2 of the ACTUAL rows), and their repeat with the
  // continuation state runs the three-pass iteration.
  const bool wr = !mc && a.F && a.EF && a.E && !a.x && !getenv("LO_OC_NO_WREC");
  // the same result-only pass with the fp64 Gram matrices of the operator at hand: the iterations run on R + 1
  // coordinates (lo_rspace.hip), one all-reduce per member (LO_OC_NO_RSPACE restores the w-recurrence kernel)
  if (wr && a.RS && a.xout && rspace_eligible(RC, a.N, a.c) && !getenv("LO_OC_NO_RSPACE")) {
    const int rc = rspace_launch(RC, a, nwg, st);
    if (rc != LO_ERR_UNSUPPORTED) return rc;
  }
#define LO_O5_G(C_, G_)                                \
  (mc ? onchip5_go<C_, G_, 1>(a, nwg, st)              \
      : (wr ? onchip5_go<C_, G_, 2>(a, nwg, st) : onchip5_go<C_, G_, 0>(a, nwg, st)))
#define LO_O5(C_)                                                                                    \
  switch (a.GW) {                                                                                    \
    case 1: return LO_O5_G(C_, 1);                                                                   \
    case 2: return LO_O5_G(C_, 2);                                                                   \
    case 4: return LO_O5_G(C_, 4);                                                                   \
    case 8: return LO_O5_G(C_, 8);                                                                   \
    case 16: return LO_O5_G(C_, 16);                                                                 \
    case 32: return LO_O5_G(C_, 32);                                                                 \
    default: return LO_O5_G(C_, 64);                                                                 \
  }
  if (RC == 32) {
    LO_O5(32);
  } else if (RC == 16) {
    LO_O5(16);
  } else if (RC == 8) {
    LO_O5(8);
  }
#undef LO_O5
#undef LO_O5_G
  return LO_ERR_UNSUPPORTED;
}

}  // namespace lo

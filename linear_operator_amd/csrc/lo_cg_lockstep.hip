// lo_cg_lockstep.hip -- operator-resident preconditioned CG that advances SIXTEEN right-hand-side columns of a member in
// lockstep on the matrix cores (third generation of the resident kernels; see lo_cg_onchip.hip / lo_cg_onchip4.hip for
// the granule hand-off and the reference citations).
//
// Why: the second generation solves the columns of a member one after the other, so an inv_quad_logdet call with 16
// probe columns (BASELINE cfg3) pays 16 x 21 x 2 group all-reduces per member and runs at ~2 % of its traffic floor.
// Every scalar of linear_cg.py:245-332 carries a trailing column dimension, i.e. the columns are independent
// recurrences that may advance together.  With 16 columns the four products of an iteration
//     W = C^T (r/d)   U = Q^T r        (contraction over the rows, "reduce")
//     r -= C (alpha T)   p += Q (-U)   (contraction over the rank, "expand")
// are 16-column GEMMs: v_mfma_f32_16x16x4_f32 (exact fp32, 64 flop/clk/SIMD).  The bound of this kernel is the fp32
// matrix rate: 2 x 8192 x (2 x 32 + 2 x 16) x 16 flop per member and iteration.
//
// Layout.  A member of N <= 8192 rows is a group of 8 workgroups x 512 threads (8 waves, 2 per SIMD, one workgroup per
// CU); a wave owns 128 rows = 8 blocks of 16.  Lane l = (kk = l >> 4, n = l & 15).
//   * the vectors r, p, x live in registers in the MFMA accumulator layout D[row = 4 kk + i][col = n] per block
//     (32 VGPRs each) -- which is also the B-operand layout B[k = kk][n] of a reduce product whose k-step i covers the
//     rows 4 kk + i, and the layout in which the expand products accumulate (r and p ARE the accumulators);
//   * C (the workgroup's 1024 x 32 rows, 128 KiB) lives in LDS, 16-byte slots XOR-swizzled, and is read in both
//     operand layouts: 4-byte reads C[row 4kk+i][16h + n] for the reduce product, 16-byte reads C[row n][16h+4kk..+3] for
//     the expand product (k-step e covers the ranks 16h + 4kk + e, which makes the D layout of T = C^T p its B layout);
//   * Q (1024 x 16) is held twice in registers, once per operand layout (2 x 32 VGPRs); d and 1/d in LDS.
//
// One all-reduce per iteration (instead of two): the reduction delivers u = Q^T r, w = C^T (r/d), s1 = sum r^2,
// s2 = sum r^2/d and rp = sum r o p_old per column, and everything else follows from per-member 32 x 16 / 16 x 16
// matrices H = C^T Q, G = Q^T D Q formed once at load:
//     r.z = s2 - |u|^2          C^T p_new = (w - H u) + beta C^T p_old        Q^T D p_new = (u - G u) + beta Q^T D p_old
//     sum d p_new^2 = dzz + 2 beta dzp + beta^2 sum d p_old^2,   dzz = s2 - 2|u|^2 + u^T G u,   dzp = rp - u^T (Q^T D p_old)
//     p.Ap = |C^T p|^2 + sum d p^2
// identical to the reference's iteration in exact arithmetic; in fp32 the solutions and the Lanczos coefficients agree
// with the fp64 iteration as closely as the reference's own fp32 arithmetic does (tests/proto/proto_single_reduction.py).
// Without a preconditioner (z = r): u, H, G vanish, s2 -> sum r^2, dzz -> sum d r^2, dzp -> sum d r p.
//
// Cross-wave sums go through LDS in two stages (waves 4-7 store, waves 0-3 add and store, every thread sums four
// partials of one or two payload entries in fixed order), then the group all-reduce through tagged 8-byte granules.
// All sums are taken in a fixed order: bitwise reproducible.
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"

namespace lo {

constexpr int LS_WROWS = 128;   // rows per wave
constexpr int LS_NBLK = 8;      // 16-row blocks per wave
constexpr int LS_NC = 16;       // columns advanced together
constexpr int LS_NVP = 832;     // payload slots allocated per workgroup (>= NV of every instantiation)
constexpr unsigned LS_MAXSPIN = 1u << 20;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// float index of C[row][col] inside the swizzled LDS image (rows of RC floats, 16-byte slots).
// RC = 32: slot ^ ((row >> 1) & 7).  Measured alternative slot ^ (row & 7): SQ_LDS_BANK_CONFLICT drops from 25 % of the
// LDS-active cycles to 0.2 % (the two kk of a half-wave then use different halves of the banks in the 4-byte reads of
// the reduce product) -- and the kernel gets 4.5 % SLOWER on the same box (4.00 vs 3.83 ms, three runs each): the
// kernel is not LDS-bound (the cause of the slowdown was not isolated).  Kept as measured.
template <int RC>
__device__ __forceinline__ int c_idx(int row, int col) {
  if constexpr (RC == 32) return row * 32 + ((((col >> 2) ^ ((row >> 1) & 7))) << 2) + (col & 3);
  else return row * 16 + ((((col >> 2) ^ ((row >> 2) & 3))) << 2) + (col & 3);
}

// sum over the four kk lane groups: every lane ends with the total of its column n
__device__ __forceinline__ float kk_sum(float v) {
  v = bfly_add<16>(v);
  v = bfly_add<32>(v);
  return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}

// value the optimiser cannot see through: address arithmetic derived from it stays inside the phase that uses it
// instead of being hoisted out of the iteration loop (where it would occupy VGPRs the resident state needs)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// granules in front of the per-CU arrival counters: (workgroups / GW) groups x 2 parities x (GW + 1) arrays
__host__ __device__ inline size_t ngroups_total_slots(unsigned nwg, int GW) {
  return (size_t)(nwg / GW) * 2 * (GW + 1) * LS_NVP;
}

struct LsGroup {
  unsigned long long* gslot;  // [2][GW][LS_NVP] granules of this group
  int wig;
  unsigned tag;
  int* err;
  bool same_xcd;
};

// part[0..NP)[0..cnt) hold the second-stage partials of this workgroup.  Entry e is summed (fixed order) and published
// as a {tag, value} granule; the entries of all workgroups of the group are then summed in fixed order -> res[e],
// bitwise identical in all workgroups.  Starts and ends with a barrier.
//   GW <= 8 : every workgroup polls the entry of all GW workgroups itself (one hand-off; GW = 1 / 2 / 4 for members of
//             up to 1024 / 2048 / 4096 rows, so that small members do not occupy eight mostly idle workgroups);
//   GW == 16: reduce-scatter + all-gather -- workgroup j sums the entries [j per, (j+1) per) of all 16 workgroups and
//             publishes the totals, everybody then reads the totals (two hand-offs, 8 x fewer loads: the payload of a
//             16-column iteration is 816 values).
template <int GW, int NP, int TPB>
__device__ __forceinline__ void ls_group_sum(float (*part)[LS_NVP], float* res, int cnt, LsGroup& g) {
  const int t = threadIdx.x;
  const unsigned tag = ++g.tag;
  __syncthreads();
  unsigned long long* slot = g.gslot + (size_t)(tag & 1u) * (GW + 1) * LS_NVP;  // GW partial arrays | one totals array
  auto publish = [&](unsigned long long* dst, float v) {
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    if (g.same_xcd) __hip_atomic_store(dst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(dst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // sum over NW consecutive granules (stride LS_NVP) of entry e, waiting for their tags
  constexpr int NG = GW < 8 ? GW : 8;  // granules of one entry gathered at a time
  auto gather8 = [&](const unsigned long long* src, int e) -> float {
    float vals[NG];
    unsigned spin = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int w = 0; w < NG; ++w) {
        const unsigned long long x =
            __hip_atomic_load(src + (size_t)w * LS_NVP + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && ((unsigned)(x >> 32) == tag);
        vals[w] = __uint_as_float((unsigned)(x & 0xffffffffull));
      }
      if (ok) break;
      if (++spin > LS_MAXSPIN ||
          ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        atomicExch(g.err, 1);  // timed out, or another workgroup already did: give up at once
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NG; ++w) tot += vals[w];
    return tot;
  };
  for (int e = t; e < cnt; e += TPB) {
    float sum = part[0][e];
#pragma unroll
    for (int q = 1; q < NP; ++q) sum += part[q][e];
    publish(slot + (size_t)g.wig * LS_NVP + e, sum);
  }
  if constexpr (GW <= 8) {
    for (int e = t; e < cnt; e += TPB) res[e] = gather8(slot, e);
  } else {
    static_assert(GW == 16, "group size");
    unsigned long long* tot = slot + (size_t)GW * LS_NVP;
    const int per = (cnt + GW - 1) / GW;
    const int lo = g.wig * per, hi = min(cnt, lo + per);
    for (int e = lo + t; e < hi; e += TPB)  // (fixed order: workgroups 0-7, then 8-15)
      publish(tot + e, gather8(slot, e) + gather8(slot + (size_t)8 * LS_NVP, e));
    for (int e = t; e < cnt; e += TPB) {
      unsigned spin = 0;
      unsigned long long x;
      for (;;) {
        x = __hip_atomic_load(tot + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == tag) break;
        if (++spin > LS_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(g.err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      res[e] = __uint_as_float((unsigned)(x & 0xffffffffull));
    }
  }
  __syncthreads();
}

// RC: padded rank of C (16 or 32).  PRE: Woodbury preconditioner z = r/d' - Q (Q^T r) with Q [.., 16] (zero padded);
// !PRE: z = r.  GW: workgroups per member.  NW: waves per workgroup -- 8 (1024 rows, one workgroup per CU, GW = 8) or
// 4 (512 rows, TWO workgroups per CU that belong to different members, GW = 16: while one waits for its hand-off the
// other keeps the matrix cores busy).
// DBG: phase timers (wall_clock64) of one work item; a separate instantiation so that the production kernel does not
// carry their registers.
template <int RC, bool PRE, int GW, int NW, bool DBG>
__global__ __launch_bounds__(NW * 64, 2) void k_cg_lockstep(OnchipArgs a) {
  constexpr int LS_TPB = NW * 64;
  constexpr int LS_ROWS = NW * LS_WROWS;        // rows per workgroup
  constexpr int NP = NW / 2;                    // second-stage partials of the cross-wave sum
  constexpr int NH = RC / 16;                   // 16-row blocks of T = C^T p
  constexpr int OFF_U = NH * 256;               // payload: W | U | three scalars per column
  constexpr int OFF_S = OFF_U + (PRE ? 256 : 0);
  constexpr int NV = OFF_S + 3 * LS_NC;
  constexpr int NVX = NV + 1;  // + the next work item of the dynamic hand-out (non-zero in an item's last reduction only)
  static_assert(NVX <= LS_NVP, "payload");
  __shared__ __attribute__((aligned(16))) float c_s[LS_ROWS * RC];
  __shared__ __attribute__((aligned(16))) float d_s[LS_ROWS];
  __shared__ __attribute__((aligned(16))) float dinv_s[LS_ROWS];
  __shared__ float part[NP][LS_NVP];
  __shared__ float res[LS_NVP];
  __shared__ float h_s[NH * 4 * 64];            // H = C^T Q in the A-operand order [(h, s)][lane]

  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;  // block b runs on XCD b % 8: keep a group behind one L2 (speed only)
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), kk = lane >> 4, n = lane & 15;
  if (t < NP) part[t][NV] = 0.f;  // the hand-out slot of the payload (first barrier: the placement check below)
  LsGroup g;
  g.gslot = a.gbuf + (size_t)grp * 2 * (GW + 1) * LS_NVP;
  g.wig = wig;
  g.tag = 0;
  g.err = a.err;
  g.same_xcd = false;
  {  // placement check through the agent-scope path (see lo_cg_onchip4.hip)
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t < 2) {
      part[0][t] = t == 0 ? (float)xcc : (float)(xcc * xcc);
#pragma unroll
      for (int q = 1; q < NP; ++q) part[q][t] = 0.f;
    }
    ls_group_sum<GW, NP, LS_TPB>(part, res, 2, g);
    const float fx = (float)xcc;
    g.same_xcd = (res[0] == GW * fx) && (res[1] == GW * fx * fx) && (a.allow_l2_handoff != 0);
    __syncthreads();
  }

  if constexpr (NW == 4) {
    // Two workgroups share a CU and both are deterministic loops of equal length: left alone they stay in phase --
    // both in their matrix-core phases (halving each other's rate), then both waiting for their hand-offs (pipes idle).
    // The first workgroup to arrive on a CU raises its wave priority: it gets the matrix cores whenever it wants them
    // and finishes its compute phase at full rate, the other one computes while the first waits -- the complementary
    // schedule establishes itself.  (CU identity from HW_REG_HW_ID[15:8] + XCC id; counters zeroed with the granules.)
    __shared__ int cu_slot;
    if (t == 0) {
      const unsigned hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);   // HW_REG_HW_ID
      const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;
      int* counters = reinterpret_cast<int*>(a.gbuf + (size_t)ngroups_total_slots(gridDim.x, GW));
      cu_slot = atomicAdd(counters + ((xcc << 8) | ((hwid >> 8) & 0xffu)), 1);
    }
    __syncthreads();
    if ((cu_slot & 1) == 0) __builtin_amdgcn_s_setprio(3);
  }

  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));  // rows of this workgroup
  const int lrow0 = w * (LS_NBLK * 16);          // first row of this wave inside the workgroup
  const int ldc = a.c;
  const int nchunk = (a.ncols + LS_NC - 1) / LS_NC;
  const int64_t nitems = a.B * nchunk;
  int64_t item = grp;
  int64_t b_loaded = -1;
  f32x4 qb[LS_NBLK];      // Q[row n of the block][4 kk .. 4 kk + 3]     (A operand of the expand product)
  float qa[LS_NBLK][4];   // Q[row 4 kk + i of the block][n]             (A operand of the reduce product)
  f32x4 ga = {0.f, 0.f, 0.f, 0.f};  // G = Q^T D Q in D layout = (symmetric) its A-operand layout

  while (item < nitems) {
    const int64_t b = item / nchunk;
    const int ch = (int)(item - b * nchunk);
    const int cbase = a.col0 + ch * LS_NC;                 // first column of this chunk
    const int ncol = min(LS_NC, a.col0 + a.ncols - cbase); // live columns
    const bool col_ok = n < ncol;
    const size_t brow = (size_t)b * a.N + row0;
    const bool stamp = DBG && a.dbg && b == a.dbg_member && ch == 0 && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();

    // ---- rhs columns -> registers (D layout); sum of squares for the normalisation (linear_cg.py:177) ----
    f32x4 r[LS_NBLK], p[LS_NBLK], x[LS_NBLK];
    float ss = 0.f;
    const int Ll = opaque(lane), kl = Ll >> 4, nl = Ll & 15;  // (the load phase's addresses must not live across the iterations)
    const int tl = opaque(t);
#pragma unroll
    for (int blk = 0; blk < LS_NBLK; ++blk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lr = lrow0 + blk * 16 + 4 * kl + i;
        float v = 0.f;
        if (nl < ncol && lr < nv) v = a.rhs[(brow + lr) * ldc + cbase + nl];
        r[blk][i] = v;
        ss = fmaf(v, v, ss);
      }
      p[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
      x[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x4 accw[NH], accu;
    if (b != b_loaded) {
      // ---- operator rows of a new member: C -> LDS (coalesced, swizzled), d, 1/d -> LDS, Q -> registers ----
      __syncthreads();  // (previous item's readers of c_s / d_s are done)
      // Q (registers) and d, 1/d are requested BEFORE the C rows: everything is in flight together and the first wait
      // (vmcnt counts in order) is the one in front of the first LDS stores of C -- three round trips per member
      // instead of eight
      constexpr int ND = LS_ROWS / LS_TPB;
      const bool d_any = a.d_mode != LO_DIAG_NONE, d_full = a.d_mode == LO_DIAG_FULL;
      float dq[ND], diq[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        const size_t grow = (size_t)b * a.N + min(row0 + j * LS_TPB + tl, a.N - 1);
        const float* dp = d_any ? (d_full ? a.d + grow : a.d + b) : a.C;
        const float* ip = PRE ? ((a.dinv_mode == LO_DIAG_FULL) ? a.dinv + grow : a.dinv + b) : a.C;
        dq[j] = *dp;
        diq[j] = *ip;
      }
      if constexpr (PRE) {
        const int RK = a.RK;  // floats per row of Q (<= 16)
#pragma unroll
        for (int blk = 0; blk < LS_NBLK; ++blk) {
          const int lrb = lrow0 + blk * 16 + nl;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (lrb < nv && 4 * kl < RK) {
            const float4 q4 = *reinterpret_cast<const float4*>(a.Q + (brow + lrb) * RK + 4 * kl);
            v = f32x4{q4.x, q4.y, q4.z, q4.w};
          }
          qb[blk] = v;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int lra = lrow0 + blk * 16 + 4 * kl + i;
            qa[blk][i] = (lra < nv && nl < RK) ? a.Q[(brow + lra) * RK + nl] : 0.f;
          }
        }
      }
      {
        const int SPG = a.RCg / 4;   // 16-byte slots per row in HBM (<= SPR: narrower roots are zero-padded here)
        const float4* csrc = reinterpret_cast<const float4*>(a.C + brow * a.RCg);
        constexpr int SPR = RC / 4;  // 16-byte slots per row of the LDS image
#pragma unroll
        for (int j0 = 0; j0 < LS_ROWS * SPR / LS_TPB; j0 += 8) {
          float4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int f = (j0 + j) * LS_TPB + tl;
            v[j] = (f / SPR < nv && f % SPR < SPG) ? csrc[(f / SPR) * SPG + f % SPR] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int f = (j0 + j) * LS_TPB + tl;
            *reinterpret_cast<float4*>(&c_s[c_idx<RC>(f / SPR, 4 * (f % SPR))]) = v[j];
          }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          const int lr = j * LS_TPB + tl;
          const bool valid = lr < nv;
          d_s[lr] = (valid && d_any) ? dq[j] : 0.f;
          dinv_s[lr] = valid ? (PRE ? diq[j] : 1.0f) : 0.f;
        }
      }
      __syncthreads();
      // ---- H = C^T Q and G = Q^T D Q (reduce products with B = Q, d o Q), together with the rhs norms ----
#pragma unroll
      for (int h = 0; h < NH; ++h) accw[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      accu = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (PRE) {
        const int L = opaque(lane), kq = L >> 4, nq = L & 15;
        const float* cw = c_s + lrow0 * RC;
        const float* dw = d_s + lrow0 + 4 * kq;
        int rb[4][NH];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < NH; ++h) rb[i][h] = c_idx<RC>(4 * kq + i, 16 * h + nq);
#pragma unroll
        for (int blk = 0; blk < LS_NBLK; ++blk) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(dw + blk * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int h = 0; h < NH; ++h) accw[h] = mfma4(cw[blk * 16 * RC + rb[i][h]], qa[blk][i], accw[h]);
            accu = mfma4(qa[blk][i], d4[i] * qa[blk][i], accu);
          }
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h) accw[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      accu = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // two-stage cross-wave sum of a payload held in (accw, accu, three per-column scalars), then the group all-reduce
    long long* tdbg = nullptr;  // phase timers of the stamped work item (iteration loop only)
    float draw_f = 0.f;         // thread 0 of the group's first workgroup: the next work item, in the item's last reduction
    auto allreduce = [&](float s0, float s1, float s2) {
      long long c0 = 0;
      if (DBG && tdbg) c0 = wall_clock64();
      s0 = kk_sum(s0);
      s1 = kk_sum(s1);
      s2 = kk_sum(s2);
      __syncthreads();  // (measured: without this re-alignment of the waves the iteration is 10 % slower)
      if (w >= NP) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) part[w - NP][(h * 4 + i) * 64 + lane] = accw[h][i];
        if constexpr (PRE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) part[w - NP][OFF_U + i * 64 + lane] = accu[i];
        }
        if (kk == 0) {
          part[w - NP][OFF_S + n] = s0;
          part[w - NP][OFF_S + 16 + n] = s1;
          part[w - NP][OFF_S + 32 + n] = s2;
        }
      }
      __syncthreads();
      if (w < NP) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) part[w][(h * 4 + i) * 64 + lane] += accw[h][i];
        if constexpr (PRE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) part[w][OFF_U + i * 64 + lane] += accu[i];
        }
        if (kk == 0) {
          part[w][OFF_S + n] += s0;
          part[w][OFF_S + 16 + n] += s1;
          part[w][OFF_S + 32 + n] += s2;
        }
      }
      if (DBG && tdbg) {
        const long long c1 = wall_clock64();
        tdbg[8] += c1 - c0;  // cross-wave stage (incl. waiting for the slowest wave)
        c0 = c1;
      }
      if (t == 0) part[0][NV] = draw_f;  // (part[1 ..][NV] stay zero)
      ls_group_sum<GW, NP, LS_TPB>(part, res, NVX, g);
      if (DBG && tdbg) tdbg[9] += wall_clock64() - c0;  // publish + poll + sum
    };

    allreduce(ss, 0.f, 0.f);
    if (b != b_loaded) {
      if constexpr (PRE) {
        // H (D layout in res) -> A-operand order in h_s: lane (kk, m) needs H[16 h + m][4 kk + s]
        // res index of H[rho][kap] (rho = 16 h + 4 kk' + i', kap = n'): (h * 4 + (rho & 3)) * 64 + ((rho >> 2) & 3) * 16 + kap
        for (int o = t; o < NH * 4 * 64; o += LS_TPB) {  // o = (h * 4 + s) * 64 + lane'
          const int hs = o >> 6, l2 = o & 63, h = hs >> 2, s = hs & 3, k2 = l2 >> 4, m = l2 & 15;
          h_s[o] = res[(h * 4 + (m & 3)) * 64 + ((m >> 2) & 3) * 16 + 4 * k2 + s];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = res[OFF_U + i * 64 + lane];
      }
      b_loaded = b;
    }
    float nrm = sqrtf(res[OFF_S + n]);                    // rhs.norm(2, dim=-2)          :177
    const bool rhs_zero = nrm < a.eps;                    // :178
    if (rhs_zero) nrm = 1.0f;                             // :179
    {
      const float inv = 1.0f / nrm;
#pragma unroll
      for (int blk = 0; blk < LS_NBLK; ++blk) r[blk] = r[blk] * inv;  // :182 (x0 = 0 -> residual = rhs)
    }
    __syncthreads();  // h_s visible; res consumed
    if (stamp) a.dbg[1] = wall_clock64();

    // ---- the reduction of an iteration: u = Q^T r, w = C^T (r/d), s1 = sum r^2, s2 = sum r^2/d (sum d r^2 without
    //      preconditioner), rp = sum r o p (sum d r p) ----
    auto reduce = [&]() {
#pragma unroll
      for (int h = 0; h < NH; ++h) accw[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      accu = f32x4{0.f, 0.f, 0.f, 0.f};
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      long long tc0 = 0;
      if (DBG && tdbg) tc0 = wall_clock64();
      const int L = opaque(lane), kq = L >> 4, nq = L & 15;
      const float* cw = c_s + lrow0 * RC;
      const float* ew = (PRE ? dinv_s : d_s) + lrow0 + 4 * kq;
      int rb[4][NH];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h) rb[i][h] = c_idx<RC>(4 * kq + i, 16 * h + nq);
#pragma unroll
      for (int blk = 0; blk < LS_NBLK; ++blk) {
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(ew + blk * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rv = r[blk][i];
          const float rd = PRE ? rv * e4[i] : rv;  // B operand of w = C^T z-part
#pragma unroll
          for (int h = 0; h < NH; ++h) accw[h] = mfma4(cw[blk * 16 * RC + rb[i][h]], rd, accw[h]);
          if constexpr (PRE) accu = mfma4(qa[blk][i], rv, accu);
          s0 = fmaf(rv, rv, s0);
          if constexpr (PRE) {
            s1 = fmaf(rd, rv, s1);
            s2 = fmaf(rv, p[blk][i], s2);
          } else {
            const float dr = e4[i] * rv;
            s1 = fmaf(dr, rv, s1);
            s2 = fmaf(dr, p[blk][i], s2);
          }
        }
      }
      if (DBG && tdbg) tdbg[7] += wall_clock64() - tc0;  // reduce products
      allreduce(s0, s1, s2);
    };

    int64_t item_next = nitems;
    if (a.iters == 0) {  // (no iteration follows: this is the item's last reduction)
      if (t == 0 && wig == 0) draw_f = (float)(ngroups + atomicAdd(a.next_member, 1));
      reduce();
      item_next = (int64_t)res[NV];
      draw_f = 0.f;
    } else {
      reduce();
    }
    float rz, beta = 0.f, alpha = 0.f, dpp = 0.f, rn;
    f32x4 tb[NH], gd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < NH; ++h) tb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      float uu = 0.f;
      if constexpr (PRE) {
        f32x4 ud;
#pragma unroll
        for (int i = 0; i < 4; ++i) ud[i] = res[OFF_U + i * 64 + lane];
        uu = kk_sum(dot4(ud, ud));
      }
      rz = PRE ? res[OFF_S + 16 + n] - uu : res[OFF_S + n];  // residual_inner_prod :215
      rn = sqrtf(res[OFF_S + n]);
    }
    bool conv = rn < a.stop_after;                          // :204-205
    const bool rec = (wig == 0 && w == 0 && kk == 0 && col_ok);
    const size_t bc = (size_t)b * ldc + cbase + n;
    if (rec) a.init_conv[bc] = conv ? 1 : 0;
    if (stamp) {
      a.dbg[2] = wall_clock64();
      if constexpr (DBG) tdbg = a.dbg;
    }

    for (int k = 0; k < a.iters; ++k) {
      long long c0 = 0;
      if (DBG && tdbg) c0 = wall_clock64();
      const bool last_it = k == a.iters - 1;
      int drawn = 0;  // (the counter is read at the top of the last iteration: its round trip hides behind the products)
      if (last_it && t == 0 && wig == 0) drawn = atomicAdd(a.next_member, 1);
      // ---- search direction: p = z + beta p with z = r/d - Q u formed on the fly (:268, :46); the small
      //      recurrences for C^T p, Q^T D p and sum d p^2 ----
      {
        f32x4 ud = {0.f, 0.f, 0.f, 0.f};
        float dzz, dzp;
        if constexpr (PRE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ud[i] = res[OFF_U + i * 64 + lane];
          f32x4 gu = {0.f, 0.f, 0.f, 0.f};
          f32x4 hu[NH];
#pragma unroll
          for (int h = 0; h < NH; ++h) hu[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int h = 0; h < NH; ++h) hu[h] = mfma4(h_s[(h * 4 + s) * 64 + lane], ud[s], hu[h]);
            gu = mfma4(ga[s], ud[s], gu);
          }
          const float uu = kk_sum(dot4(ud, ud));
          const float ugu = kk_sum(dot4(ud, gu));
          const float ug = kk_sum(dot4(ud, gd));
          dzz = fmaf(-2.f, uu, res[OFF_S + 16 + n]) + ugu;
          dzp = res[OFF_S + 32 + n] - ug;
#pragma unroll
          for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) tb[h][i] = fmaf(beta, tb[h][i], res[(h * 4 + i) * 64 + lane] - hu[h][i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) gd[i] = fmaf(beta, gd[i], ud[i] - gu[i]);
        } else {
          dzz = res[OFF_S + 16 + n];
          dzp = res[OFF_S + 32 + n];
#pragma unroll
          for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) tb[h][i] = fmaf(beta, tb[h][i], res[(h * 4 + i) * 64 + lane]);
        }
        dpp = fmaf(beta, fmaf(beta, dpp, 2.f * dzp), dzz);
        const float* ew = dinv_s + lrow0 + 4 * (opaque(lane) >> 4);
#pragma unroll
        for (int blk = 0; blk < LS_NBLK; ++blk) {
          if constexpr (PRE) {
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(ew + blk * 16);
            f32x4 acc;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(beta, p[blk][i], e4[i] * r[blk][i]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma4(qb[blk][s], -ud[s], acc);
            p[blk] = acc;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) p[blk][i] = fmaf(beta, p[blk][i], r[blk][i]);
          }
        }
      }
      if (DBG && tdbg) {
        const long long c1 = wall_clock64();
        tdbg[5] += c1 - c0;  // search direction
        c0 = c1;
      }
      // ---- p.Ap = |C^T p|^2 + sum d p^2 ; alpha (:250-260) ----
      {
        float tt = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) tt += dot4(tb[h], tb[h]);
        const float pAp = kk_sum(tt) + dpp;
        alpha = (pAp < a.eps) ? 0.f : rz / pAp;             // :254-257
        if (conv) alpha = 0.f;                              // :260
      }
      // ---- x += alpha p (:31);  r -= alpha (C t + d o p) (:264): r is the accumulator of the expand product ----
      {
        f32x4 nb[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) nb[h] = tb[h] * (-alpha);
        const int L = opaque(lane), kq = L >> 4, nq = L & 15;
        const float* cw = c_s + lrow0 * RC;
        const float* dw = d_s + lrow0 + 4 * kq;
        int eb[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) eb[h] = c_idx<RC>(nq, 16 * h + 4 * kq);
#pragma unroll
        for (int blk = 0; blk < LS_NBLK; ++blk) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(dw + blk * 16);
          f32x4 acc;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[blk][i] = fmaf(alpha, p[blk][i], x[blk][i]);
            acc[i] = fmaf(-alpha * d4[i], p[blk][i], r[blk][i]);
          }
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cw + blk * 16 * RC + eb[h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma4(c4[e], nb[h][e], acc);
          }
          r[blk] = acc;
        }
      }
      if (DBG && tdbg) {
        const long long c1 = wall_clock64();
        tdbg[6] += c1 - c0;  // alpha, x and r updates
        c0 = c1;
      }
      if (last_it && t == 0 && wig == 0) draw_f = (float)(ngroups + drawn);
      reduce();
      if (last_it) {
        item_next = (int64_t)res[NV];
        draw_f = 0.f;
      }
      {
        float uu = 0.f;
        if constexpr (PRE) {
          f32x4 ud;
#pragma unroll
          for (int i = 0; i < 4; ++i) ud[i] = res[OFF_U + i * 64 + lane];
          uu = kk_sum(dot4(ud, ud));
        }
        const float rzn = PRE ? res[OFF_S + 16 + n] - uu : res[OFF_S + n];  // :35-36
        beta = (rz < a.eps) ? 0.f : rzn / rz;               // :39-42
        rz = rzn;
        rn = sqrtf(res[OFF_S + n]);                         // :298
        if (rhs_zero) rn = 0.f;                             // :299
        conv = rn < a.stop_after;                           // :300
      }
      if (rec) {
        const size_t o = (size_t)k * a.B * ldc + bc;
        a.resid_rec[o] = rn;
        if (a.ab_rec) {  // masked alpha and beta of this iteration: the tridiagonal recurrence is replayed afterwards
          a.ab_rec[2 * o] = alpha;
          a.ab_rec[2 * o + 1] = beta;
        }
      }
    }
    if (stamp) a.dbg[3] = wall_clock64();
    tdbg = nullptr;

    // ---- write the state back in the streaming engine's layout (p is the direction of the last iteration; the
    //      streaming loop forms z + beta p itself) ----
    {
      f32x4 ud = {0.f, 0.f, 0.f, 0.f};
      if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ud[i] = res[OFF_U + i * 64 + lane];
      }
      const int L = opaque(lane), kq = L >> 4, nq = L & 15;
      const float* ew = dinv_s + lrow0 + 4 * kq;
      const bool cok = nq < ncol;
#pragma unroll
      for (int blk = 0; blk < LS_NBLK; ++blk) {
        f32x4 z = r[blk];
        if constexpr (PRE) {
          const f32x4 e4 = *reinterpret_cast<const f32x4*>(ew + blk * 16);
          z = z * e4;
#pragma unroll
          for (int s = 0; s < 4; ++s) z = mfma4(qb[blk][s], -ud[s], z);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lr = lrow0 + blk * 16 + 4 * kq + i;
          if (cok && lr < nv) {
            const size_t o = (brow + lr) * ldc + cbase + nq;
            if (a.xout) a.xout[o] = x[blk][i] * nrm;  // final when the stop rule holds at the floor (:335)
            if (a.x) {  // (nullptr: result only, see k_cg_onchip5)
              a.x[o] = x[blk][i];
              a.r[o] = r[blk][i];
              a.p[o] = p[blk][i];
              if (a.z) a.z[o] = z[i];
            }
          }
        }
      }
    }
    if (rec) {
      a.rhs_norm[bc] = nrm;
      a.rhs_is_zero[bc] = rhs_zero ? 1 : 0;
      a.rz[bc] = rz;
      a.alpha[bc] = alpha;
      a.beta[bc] = beta;
      a.resid_norm[bc] = rn;
      a.has_conv[bc] = conv ? 1 : 0;
    }
    if (stamp) a.dbg[4] = wall_clock64();
    // next work item: drawn by the group's first workgroup one phase ahead and carried by the item's last reduction
    // (one contributor, the rest add zeros: exact for indices < 2^24) -- no all-reduce of its own
    item = item_next;
    __syncthreads();
  }
}

size_t lockstep_gbuf_bytes(int ngroups, int GW) {  // granules + 4096 per-CU arrival counters
  return (size_t)ngroups * 2 * (GW + 1) * LS_NVP * sizeof(unsigned long long) + 4096 * sizeof(int);
}

bool lockstep_eligible(int RC, int RK, bool pre, int64_t N, int64_t ncols) {
  const bool rc_ok = (RC == 8 || RC == 16 || RC == 32);
  const bool rk_ok = !pre || (RK == 4 || RK == 8 || RK == 16);
  return rc_ok && rk_ok && ncols >= 1 && N >= 256 && N <= 8 * (int64_t)8 * LS_WROWS;  // (N < 1024: a group of one)
}

template <int RC, bool PRE, int GW, int NW, bool DBG = false>
static int lockstep_go(const OnchipArgs& a, int ncu, hipStream_t st) {
  // the spin-waiting groups need ALL workgroups resident: one (NW = 8) or two (NW = 4) per CU
  constexpr int per_cu_needed = NW == 4 ? 2 : 1;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_lockstep<RC, PRE, GW, NW, DBG>, NW * 64, 0) !=
          hipSuccess ||
      per_cu < per_cu_needed)
    return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_lockstep", st);
  ResidentLaunch guard(st);
  hipLaunchKernelGGL((k_cg_lockstep<RC, PRE, GW, NW, DBG>), dim3(per_cu_needed * ncu), dim3(NW * 64), 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// 8: one 1024-row workgroup per CU (default).  16 (LO_LS_V2=1, N > 4096): two 512-row workgroups per CU that belong to
// different members.  Measured equal (3.78 vs 3.83 ms at cfg3): a wave's matrix-core stream is not software-pipelined
// against its LDS operand reads, so one workgroup alone reaches ~55 % of the matrix rate and the second workgroup's
// compute phase cannot fill the first one's hand-off wait any better than the second wave per SIMD already does.
int lockstep_group_size(int64_t N) {
  if (N > 4096 && getenv("LO_LS_V2")) return 16;
  if (getenv("LO_OC_GW8")) return 8;
  return N <= 1024 ? 1 : (N <= 2048 ? 2 : (N <= 4096 ? 4 : 8));
}

// ncu = number of CUs used (multiple of 64).  a.GW selects the variant (8: one 1024-row workgroup per CU; 16: two
// 512-row workgroups per CU); a.RK = floats per row of Q (PRE) or 0.  LO_ERR_UNSUPPORTED when the workgroups do not
// fit (the caller falls back to the other variant / the serial-column kernels).
int lockstep_launch(int RC, bool pre, const OnchipArgs& a, int ncu, hipStream_t st) {
  const bool v2 = a.GW == 16;
#define LO_LS(C_, P_)                                                      \
  (v2 ? lockstep_go<C_, P_, 16, 4>(a, ncu, st)                             \
      : (a.GW == 8 ? lockstep_go<C_, P_, 8, 8>(a, ncu, st)                 \
                   : (a.GW == 4 ? lockstep_go<C_, P_, 4, 8>(a, ncu, st)    \
                                : (a.GW == 2 ? lockstep_go<C_, P_, 2, 8>(a, ncu, st) : lockstep_go<C_, P_, 1, 8>(a, ncu, st)))))
#ifdef LO_DEBUG_KERNELS  // (phase timers: two more instantiations that spill 46 registers; make CXXFLAGS+=-DLO_DEBUG_KERNELS)
  if (RC == 32 && pre && a.dbg) return v2 ? lockstep_go<32, true, 16, 4, true>(a, ncu, st) : lockstep_go<32, true, 8, 8, true>(a, ncu, st);
#endif
  if (RC == 32) return pre ? LO_LS(32, true) : LO_LS(32, false);
  if (RC == 16 || RC == 8) return pre ? LO_LS(16, true) : LO_LS(16, false);
#undef LO_LS
  return LO_ERR_UNSUPPORTED;
}

}  // namespace lo

// lo_pivchol_onchip.hip -- operator-resident pivoted Cholesky for the low-rank-root + diagonal operator
// (PivotedCholesky.forward, linear_operator/functions/_pivoted_cholesky.py:14-105, row source
// RootLinearOperator._get_indices root_linear_operator.py:37-50 / _diagonal :22-28).
//
// The streaming engine (lo_pivchol.hip) re-reads C (4NR bytes) and the m finished rows of L (4mN bytes) from HBM
// for every pivot.  Here a group of GW workgroups (256 threads x 4 rows; GW = smallest power of two that holds the member)
// owns one batch member for ALL pivots: a thread keeps its four rows of C in registers (as two PAIRS of rows: a register
// pair per column, so the chains of a pivot run as packed products and sums), their running diagonal and position in the
// permutation as well, and their L entries L[0..m-1][i] in LDS (pair rows, swizzled 16-byte slots); two workgroups of
// different members share a CU.  HBM traffic per member: C once (4NR) + L once (4 max_rank N).
// Per pivot the group needs ONE exchange (8-byte {value, tag} granules, the hand-off of lo_group_reduce.h): every
// workgroup publishes its best candidate (diagonal value, position, row), that row's C row and L entries, and its partial
// of the error 1-norm; everybody then picks the same winner and has all it needs for the Schur update of its own rows.
// (The first generation -- one row per thread, C in LDS, L in registers, 1.0 ms at the headline shape against 0.58 -- was
// removed in round 6; nothing had selected it since round 2.)
//
// What a pivot costs (variants of this file with one piece removed, timed on one box; 512 x 8192 x 32, rank
// 15: 566 us = 100 us of HBM time + 120 pivots of 3.9 us per workgroup): C.C chain 0.43, L.L chain 0.52, the group
// exchange 0.62 (0.52 of it the wait for the slowest workgroup), the owner lanes' LDS stores of their C rows 0.52, wave
// argmax 0.17, winner 0.20, division 0.08 -- and 1.3 us are left when all of these are gone.  No piece dominates; a
// member alone on its CUs takes 59 us, two members per CU 71 us each.
//
// Every operation that feeds a pivot decision is the same individually rounded, fixed-order arithmetic as
// lo_pivchol.hip / oracle.pivoted_cholesky (file compiled with -ffp-contract=off), so L and the permutation
// are bit-identical to the streaming engine and to the CPU path.
//
// The batch-global stopping rule (:57: one shared m, stop when max_b error <= tol) cannot be evaluated inside
// a kernel whose groups work on different members at different times, so the kernel takes all `rank` pivots and
// records the error after each; k_po_perm then finds the reference's m* (po_rank) and rebuilds the permutation
// from the recorded swaps 0..m*-1 and clears the rows of L the reference would not have written.
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#pragma clang fp contract(off)

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int PO_GW = 8;
constexpr int PO_SLOT = 72;   // granules per workgroup and parity (header + 32 C entries + up to 32 L entries); also the
                              // words of a wave's candidate in LDS (header + the 2 x 32 words of its row PAIR)
constexpr int PO_MAXR = 16;   // up to here: 4 slots of 16 bytes per row (64 KB of L rows, two workgroups per CU)
constexpr int P4_MAXR = 32;   // above 16: 8 slots of 16 bytes per row (128 KB, one workgroup per CU)
constexpr int PO_HDR = 4;     // value, position, (k_pc_onchip4, in LDS only: the candidate's row inside its workgroup), error partial
constexpr unsigned PO_MAXSPIN = 1u << 20;  // ~0.5 s of polling: co-residency was lost (never seen on a dedicated GPU)
constexpr int PO_INVALID = 0x7fffffff;

struct PoArgs {
  const float* C;  // [B, N, RC]
  int64_t B;
  int N, RW, rank, max_rank;
  float* L;        // [B, max_rank, N]
  float* err_rec;  // [rank, B]  sum |diag| over the positions >= m, taken before pivot m
  float* orig;     // [B] max of the initial diagonal (:43)
  int* swaps;      // [B, max_rank] position exchanged with position m at pivot m
  unsigned long long* gbuf;  // [ngroups][2][PO_GW][PO_SLOT]
  int* err;
  int allow_l2_handoff;
  int prio;        // k_pc_onchip4: wave-priority mask (LO_PC_PRIO): 1 = the candidate reduction / exchange / winner part of
                   // a pivot runs at raised priority, the instruction-dense Schur update at normal priority
  long long* dbg;  // optional phase timers (wall_clock64 ticks) of member 0 / workgroup 0, or nullptr
};

__device__ __forceinline__ bool po_better(float ov, int oj, float mv, int mj) {
  // FIRST maximal position wins (torch.max on CPU, :61-63)
  return oj != PO_INVALID && (mj == PO_INVALID || ov > mv || (ov == mv && oj < mj));
}

// argmax butterfly step over lane bit M on (value, position): FIRST maximal position wins; positions of valid
// candidates are unique, so the unordered {own, partner} pair of bfly_i resolves identically in both lanes
template <int M>
__device__ __forceinline__ void po_amax_step(float& v, int& j) {
  int va, vb, ja, jb;
  bfly_i<M>(__float_as_int(v), va, vb);
  bfly_i<M>(j, ja, jb);
  const float fa = __int_as_float(va), fb = __int_as_float(vb);
  const bool tb = po_better(fb, jb, fa, ja);
  v = tb ? fb : fa;
  j = tb ? jb : ja;
}

// ---- 4 rows per thread, two workgroups per CU (same idea as lo_cg_onchip4.hip) ---------------------------------
// Workgroup = 256 threads x 4 rows; the 4 C rows of a thread live in VGPRs, the L rows of the workgroup in LDS
// (1024 x 16 floats, 16-byte slots XOR-swizzled), so the pivot index is a run-time value and the pivot loop is a
// plain loop.  A member is a group of GW = 8 (N <= 8192), 16 (N <= 16384) or 32 (N <= 32768) workgroups; two workgroups of
// different members share a CU, one updates its rows while the other waits for its exchange.  Arithmetic and
// operation order per row are those of the first generation, hence bit-identical results.
constexpr int P4_TPB = 256;
constexpr int P4_NR = 4;
constexpr int P4_WAVES = P4_TPB / 64;
constexpr int P4_ROWS = P4_TPB * P4_NR;

template <int GW>
struct alignas(16) P4Shared {
  unsigned part[PO_SLOT];            // (placement check only)
  unsigned part4[P4_WAVES][PO_SLOT]; // every wave's candidate: header, the register pairs of its C row (2 x 32 words)
  unsigned gath[GW][PO_SLOT];
};

// LQ = 16-byte slots per row: 4 (rank <= 16, rows 64 bytes apart, XOR over groups of 4 rows) or 8 (rank <= 32, rows 128
// bytes apart: two consecutive rows cover the 64 banks, XOR over pairs of rows)  [k_pc_onchip_rows]
template <int LQ>
__device__ __forceinline__ int l_slot(int r, int q) {
  if constexpr (LQ == 4) return r * 4 + (q ^ ((r >> 2) & 3));
  else return r * 8 + (q ^ ((r >> 1) & 7));
}

// k_pc_onchip4 keeps the L entries of a PAIR of a thread's rows side by side (rows t + 256 (2 p) and t + 256 (2 p + 1),
// p = 0, 1: the "pair row" pr = t + 256 p): slot jb of a pair row = {L[r0][2 jb], L[r1][2 jb], L[r0][2 jb + 1],
// L[r1][2 jb + 1]}, so one ds_read_b128 feeds two packed products.  NS = 16-byte slots per pair row: 8 (rank <= 16, pair
// rows 128 bytes apart) or 16 (rank <= 32, 256 bytes apart), XOR-swizzled so that 8 consecutive lanes cover the 64 banks.
template <int NS>
__device__ __forceinline__ int l_pslot(int pr, int jb) {
  if constexpr (NS == 8) return pr * 8 + (jb ^ ((pr >> 1) & 7));
  else return pr * 16 + (jb ^ (pr & 15));
}
// entry j of the workgroup's row lr (0 .. 1023)
template <int NS>
__device__ __forceinline__ float l_entry(const float4* l_s, int lr, int j) {
  const int q = lr >> 8, pr = (lr & 255) + 256 * (q >> 1);
  return reinterpret_cast<const float*>(&l_s[l_pslot<NS>(pr, j >> 1)])[2 * (j & 1) + (q & 1)];
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// packed products of a pair with ONE factor taken from the low / high half of a register pair (v_pk_mul_f32 with op_sel
// rounds each half like v_mul_f32: the mandated individually rounded product, two rows per instruction; the file is
// compiled with -ffp-contract=off, so the sums stay separate v_pk_add_f32)
template <int HI>
__device__ __forceinline__ f2 pk_mul_bcast(f2 a, f2 b) {
  const float s = HI ? b.y : b.x;
  return a * (f2){s, s};
}
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { return a + b; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { return a * b; }

// xor-4 partner on the DPP path (two row shifts under complementary bank masks) instead of the LDS-crossbar ds_swizzle
__device__ __forceinline__ int p4_xor4(int x) {
  const int a = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0, 2 read lane + 4
  return __builtin_amdgcn_update_dpp(a, x, 0x114, 0xf, 0xa, false);          // row_shr:4 -> banks 1, 3 read lane - 4
}
// wave_sum_fast (lo_device.h) with the xor-4 step on the DPP path: same operands, same order, same bits
__device__ __forceinline__ float p4_wave_sum(float v) {
  v = bfly_add<1>(v); v = bfly_add<2>(v);
  v = v + __int_as_float(p4_xor4(__float_as_int(v)));
  v = bfly_add<8>(v); v = bfly_add<16>(v); v = bfly_add<32>(v);
  return v;
}

// (value, position) of the wave's best candidate in every lane: FIRST maximal position wins (torch.max on CPU, :61-63).
// The value maximum runs as six cross-lane maxima; the position comes from the winner's lane (v_readlane) -- a tie of
// the maximum (exact float equality, rare) takes the smallest position among the tied lanes through a second reduction.
// The returned value is the winner's own bits (a maximum of +0 and -0 would not say which); returns the winner's lane
// (or -1: no candidate).
__device__ __forceinline__ int p4_wave_argmax(float& v, int& j) {
  float mx = (j == PO_INVALID) ? -INFINITY : v;
  mx = fmaxf(mx, xor_lane<1>(mx)); mx = fmaxf(mx, xor_lane<2>(mx));
  mx = fmaxf(mx, __int_as_float(p4_xor4(__float_as_int(mx))));
  mx = fmaxf(mx, xor_lane<8>(mx));
  {
    int a_, b_;
    bfly_i<16>(__float_as_int(mx), a_, b_);
    mx = fmaxf(__int_as_float(a_), __int_as_float(b_));
    bfly_i<32>(__float_as_int(mx), a_, b_);
    mx = fmaxf(__int_as_float(a_), __int_as_float(b_));
  }
  const bool eq = j != PO_INVALID && v == mx;
  const unsigned long long bal = __ballot(eq);
  if (bal == 0ull) {  // no candidate (or only NaNs): nothing to offer
    v = -INFINITY;
    j = PO_INVALID;
    return -1;
  }
  int lane_w = __ffsll((long long)bal) - 1;
  if (__popcll(bal) > 1) {  // tie: smallest position among the tied lanes
    int jm = eq ? j : PO_INVALID;
    jm = min(jm, xor_lane_i<1>(jm)); jm = min(jm, xor_lane_i<2>(jm)); jm = min(jm, p4_xor4(jm));
    jm = min(jm, xor_lane_i<8>(jm));
    int a_, b_;
    bfly_i<16>(jm, a_, b_);
    jm = min(a_, b_);
    bfly_i<32>(jm, a_, b_);
    jm = min(a_, b_);
    const unsigned long long bw = __ballot(eq && j == jm);
    lane_w = __ffsll((long long)bw) - 1;
  }
  v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_w));
  j = __builtin_amdgcn_readlane(j, lane_w);
  return lane_w;
}

// thread t < cnt publishes component t of the workgroup's candidate and fetches component t of every workgroup of the
// group into sh.gath.  WIN = false (placement check): the component is sh.part[t].  WIN = true (a pivot): the value comes
// from the four wave candidates in sh.part4 -- every publishing thread picks the workgroup's winner itself from the four
// headers, in the order of the butterfly the workgroup-level reduction used to run, so the candidate phase needs ONE
// barrier; the winner's C row comes from its wave's sh.part4, its L entries 0 .. m-1 straight from the L rows in LDS
// (header word 2 = the candidate's row inside the workgroup): no owner lane copies them first.
template <int GW, int NS, int RC, bool WIN>
__device__ __forceinline__ void p4_gather(P4Shared<GW>& sh, const float4* l_s, int cnt, unsigned long long* gslot_base,
                                          int wig, unsigned tag, int* err, bool same_xcd) {
  const int t = threadIdx.x;
  __syncthreads();  // sh.part / sh.part4 and the L entries of the last pivot complete
  if (t < cnt) {
    unsigned long long* slot = gslot_base + (size_t)(tag & 1u) * GW * PO_SLOT;
    unsigned myval;
    if constexpr (WIN) {
      float v0 = __uint_as_float(sh.part4[0][0]), v1 = __uint_as_float(sh.part4[1][0]);
      float v2 = __uint_as_float(sh.part4[2][0]), v3 = __uint_as_float(sh.part4[3][0]);
      int j0 = (int)sh.part4[0][1], j1 = (int)sh.part4[1][1], j2 = (int)sh.part4[2][1], j3 = (int)sh.part4[3][1];
      int w01 = 0, w23 = 2;
      if (po_better(v1, j1, v0, j0)) { v0 = v1; j0 = j1; w01 = 1; }
      if (po_better(v3, j3, v2, j2)) { v2 = v3; j2 = j3; w23 = 3; }
      const int wsel = po_better(v2, j2, v0, j0) ? w23 : w01;
      const float ge = (__uint_as_float(sh.part4[0][3]) + __uint_as_float(sh.part4[1][3])) +
                       (__uint_as_float(sh.part4[2][3]) + __uint_as_float(sh.part4[3][3]));  // (the butterfly's order)
      const int lr = (int)sh.part4[wsel][2] & (P4_ROWS - 1);  // (no candidate at all: any row, nobody reads it)
      if (t >= PO_HDR + RC) {
        myval = __float_as_uint(l_entry<NS>(l_s, lr, t - (PO_HDR + RC)));
      } else if (t >= PO_HDR) {
        myval = sh.part4[wsel][PO_HDR + 2 * (t - PO_HDR) + ((lr >> 8) & 1)];  // (pairs {row 2 p, row 2 p + 1} per column)
      } else {
        myval = (t == 3) ? __float_as_uint(ge) : sh.part4[wsel][t];
      }
    } else {
      myval = sh.part[t];
    }
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)myval;
    if (same_xcd)
      __hip_atomic_store(slot + (size_t)wig * PO_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)wig * PO_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spin = 0;
    if constexpr (GW <= 16) {
      unsigned vals[GW];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
          const unsigned long long x =
              __hip_atomic_load(slot + (size_t)w * PO_SLOT + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(x >> 32) == tag);
          vals[w] = (unsigned)(x & 0xffffffffull);
        }
        if (ok) break;
        if (++spin > PO_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(err, 1);  // timed out, or another workgroup already did: give up at once
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int w = 0; w < GW; ++w) sh.gath[w][t] = vals[w];
    } else {
      // large groups: poll the tag words (upper halves of the granules, all loads in flight together), then fetch
      // the value words -- final once the tags match, see lo_cg_onchip4.hip -- straight into LDS
      const unsigned* words = reinterpret_cast<const unsigned*>(slot);
      for (;;) {
        unsigned bad = 0;
#pragma unroll
        for (int w = 0; w < GW; ++w)
          bad |= __hip_atomic_load(words + 2 * ((size_t)w * PO_SLOT + t) + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) ^ tag;
        if (bad == 0) break;
        if (++spin > PO_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int h = 0; h < GW; h += 16) {
        unsigned v[16];
#pragma unroll
        for (int w = 0; w < 16; ++w)
          v[w] = __hip_atomic_load(words + 2 * ((size_t)(h + w) * PO_SLOT + t), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int w = 0; w < 16; ++w) sh.gath[h + w][t] = v[w];
      }
    }
  }
  __syncthreads();
}

// the member's rows of C: every request in flight before anything waits (vmcnt counts in order).  A wave fetches its 64
// consecutive rows as 64 * RC / 4 CONSECUTIVE 16-byte chunks (whole cache lines per instruction instead of 16 bytes out
// of 64 lines); padding rows read a clamped valid row and are zeroed after the transposition.
template <int RC>
__device__ __forceinline__ void p4_issue_loads(const float* C, int64_t b, int N, int row0, int tl,
                                               float4 (&raw)[P4_NR][RC / 4]) {
  constexpr int CH = RC / 4;
  const int wv = tl >> 6, ln = tl & 63;
#pragma unroll
  for (int q = 0; q < P4_NR; ++q) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int g = 64 * i + ln;
      const int rw = g / CH, ck = g % CH;
      const size_t grow = (size_t)b * N + min(row0 + P4_TPB * q + 64 * wv + rw, N - 1);
      raw[q][i] = *reinterpret_cast<const float4*>(C + grow * RC + 4 * ck);
    }
  }
}

// phase timers without registers carried between the stamps: a stamp adds the clock to the phase that ends here and
// subtracts it from the phase(s) that begin (the phase's sum of end - start builds up in memory)
__device__ __forceinline__ void p4_stamp(long long* dbg, int ends, int begins, int begins2 = -1) {
  const unsigned long long n = (unsigned long long)wall_clock64();
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dbg);
  if (ends >= 0) atomicAdd(d + ends, n);
  if (begins >= 0) atomicAdd(d + begins, 0ull - n);
  if (begins2 >= 0) atomicAdd(d + begins2, 0ull - n);
}

template <int RC, int GW, int LQ>
__global__ __launch_bounds__(P4_TPB, 2) void k_pc_onchip4(PoArgs a) {
  constexpr int NS = 2 * LQ;  // 16-byte slots per pair row
  __shared__ P4Shared<GW> sh;
  // L entries of this workgroup's rows (pair rows, swizzled 16-byte slots): 64 KB static (two workgroups per CU) for
  // rank <= 16, 128 KB dynamic (one workgroup per CU) for rank <= 32
  __shared__ float4 l_static[LQ == 4 ? P4_ROWS * 4 : 1];
  extern __shared__ float4 l_dynamic[];
  float4* const l_s = (LQ == 4) ? l_static : l_dynamic;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long* gslot = a.gbuf + (size_t)grp * 2 * GW * PO_SLOT;
  unsigned tag = 0;
  bool same_xcd = false;
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t == 0) sh.part[0] = xcc;
    p4_gather<GW, NS, RC, false>(sh, l_s, 1, gslot, wig, ++tag, a.err, false);
    bool same = true;
#pragma unroll
    for (int w = 1; w < GW; ++w) same = same && (sh.gath[w][0] == sh.gath[0][0]);
    same_xcd = same && (a.allow_l2_handoff != 0);
    __syncthreads();
  }
  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));
  constexpr int CH = RC / 4;

  float4 raw[P4_NR][CH];
  {
    int tl = t;
    asm volatile("" : "+v"(tl));
    if (grp < a.B) p4_issue_loads<RC>(a.C, grp, a.N, row0, tl, raw);
  }
  for (int64_t b = grp; b < a.B; b += ngroups) {
    const bool stamp = a.dbg && b == 64 && wig == 0 && t == 0;  // a member of the second round
    if (stamp) a.dbg[0] = wall_clock64();
    int tl = t;
    asm volatile("" : "+v"(tl));  // keeps the address arithmetic of a phase inside the member loop (VGPR budget)
    // Cq[p][k] = {C[r0][2 k], C[r1][2 k], C[r0][2 k + 1], C[r1][2 k + 1]}, r0 = row t + 256 (2 p), r1 = r0 + 256: the two
    // rows of a pair share a register pair per column, the chains of a pivot run as packed products and sums (two rows
    // per instruction); two columns share a 16-byte register group (the owner lane's LDS stores of a pivot row)
    f4 Cq[2][RC / 2];
    f2 dg2[2];
    int pos[P4_NR];
    // A wave parks the 64 x CH chunks of a row set in its own window of the (not yet used) L rows, chunk slot
    // XOR-swizzled by the row, and reads its row back
    const int wv = tl >> 6, ln = tl & 63;
#pragma unroll
    for (int q = 0; q < P4_NR; ++q) {
      const int lr = tl + P4_TPB * q;
      const bool valid = lr < nv;
      float4* win = l_s + wv * (64 * CH);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int g = 64 * i + ln;
        const int rw = g / CH, ck = g % CH;
        win[rw * CH + (ck ^ ((rw ^ (rw >> 3)) & (CH - 1)))] = raw[q][i];
      }
      __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order; this pins the compiler's order)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float4 c4 = win[ln * CH + (i ^ ((ln ^ (ln >> 3)) & (CH - 1)))];
        c4.x = valid ? c4.x : 0.f; c4.y = valid ? c4.y : 0.f; c4.z = valid ? c4.z : 0.f; c4.w = valid ? c4.w : 0.f;
        if (q & 1) {
          Cq[q >> 1][2 * i].y = c4.x; Cq[q >> 1][2 * i].w = c4.y; Cq[q >> 1][2 * i + 1].y = c4.z; Cq[q >> 1][2 * i + 1].w = c4.w;
        } else {
          Cq[q >> 1][2 * i].x = c4.x; Cq[q >> 1][2 * i].z = c4.y; Cq[q >> 1][2 * i + 1].x = c4.z; Cq[q >> 1][2 * i + 1].z = c4.w;
        }
      }
      __builtin_amdgcn_wave_barrier();  // the next row set reuses the window
      pos[q] = valid ? row0 + lr : PO_INVALID;
    }
    // (opaque: otherwise the 16-byte values read from the window are kept alive -- in scratch -- for the owner lane's
    //  16-byte LDS stores of a pivot row, which are rebuilt from the pairs instead)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < RC / 2; ++k) asm volatile("" : "+v"(Cq[p][k]));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f2 acc = Cq[p][0].lo * Cq[p][0].lo;  // (root ** 2).sum(-1), sequential in r (zero rows give 0)
      acc = acc + Cq[p][0].hi * Cq[p][0].hi;
#pragma unroll
      for (int k = 1; k < RC / 2; ++k) {
        acc = acc + Cq[p][k].lo * Cq[p][k].lo;
        acc = acc + Cq[p][k].hi * Cq[p][k].hi;
      }
      dg2[p] = acc;
    }
    float dg[P4_NR] = {dg2[0].x, dg2[0].y, dg2[1].x, dg2[1].y};
    __syncthreads();  // the windows are done: the L rows can be cleared
#pragma unroll
    for (int i = 0; i < 4 * LQ; ++i) l_s[tl + P4_TPB * i] = make_float4(0.f, 0.f, 0.f, 0.f);

    if (stamp) a.dbg[1] = wall_clock64();
    for (int m = 0; m < a.rank; ++m) {
      if (stamp) p4_stamp(a.dbg, -1, 4);
      if (a.prio & 1) __builtin_amdgcn_s_setprio(3);
      // ---- workgroup candidate: argmax of the running diagonal over the positions >= m, error 1-norm partial ----
      float bv = -INFINITY, es = 0.f;
      int bj = PO_INVALID, bq = 0;
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) {
        const bool cand = pos[q] != PO_INVALID && pos[q] >= m;
        if (cand) {
          es += fabsf(dg[q]);
          if (po_better(dg[q], pos[q], bv, bj)) {
            bv = dg[q];
            bj = pos[q];
            bq = q;
          }
        }
      }
      const int lane_w = p4_wave_argmax(bv, bj);
      es = p4_wave_sum(es);
      // every WAVE's candidate goes to LDS with its header and C row: the four owner lanes write side by side, and the
      // workgroup's winner is picked by the publishing threads after ONE barrier (p4_gather<.., true>)
      unsigned* const mine4 = sh.part4[wave];
      if (lane == 0) {
        mine4[0] = __float_as_uint(bv);   // (-inf, PO_INVALID when the wave has no candidate left)
        mine4[1] = (unsigned)bj;
        mine4[3] = __float_as_uint(es);
      }
      if (lane_w >= 0) {
        // (the owner lane writes the register PAIRS that hold the row -- both rows of the pair, 8-byte LDS stores straight
        //  from the registers; the publishing threads pick the row's half.  Which pair is a wave-uniform value, so exactly
        //  one of the two compile-time variants runs)
        const int bq_u = __builtin_amdgcn_readlane(bq, lane_w);
        f4* dst = reinterpret_cast<f4*>(&mine4[PO_HDR]);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if ((bq_u >> 1) == p && lane == lane_w) {
            mine4[2] = (unsigned)(t + P4_TPB * bq);
#pragma unroll
            for (int k = 0; k < RC / 2; ++k) dst[k] = Cq[p][k];
          }
        }
      }
      if (stamp) p4_stamp(a.dbg, 4, 5);
      p4_gather<GW, NS, RC, true>(sh, l_s, PO_HDR + RC + m, gslot, wig, ++tag, a.err, same_xcd);
      if (stamp) p4_stamp(a.dbg, 5, 6, 7);

      // ---- group winner (identical in all workgroups): lane l holds candidate l % GW ----
      float vb = __uint_as_float(sh.gath[lane % GW][0]);
      const int myj = (int)sh.gath[lane % GW][1];
      int jb = myj;
      float etot = __uint_as_float(sh.gath[lane % GW][3]);
      if constexpr (GW >= 2) {  // (lane l holds candidate l % GW: only the butterfly steps below GW combine distinct ones)
        etot = bfly_add<1>(etot);
        po_amax_step<1>(vb, jb);
      }
      if constexpr (GW >= 4) {
        etot = bfly_add<2>(etot);
        po_amax_step<2>(vb, jb);
      }
      if constexpr (GW >= 8) {
        etot = bfly_add<4>(etot);
        po_amax_step<4>(vb, jb);
      }
      if constexpr (GW >= 16) {
        etot = bfly_add<8>(etot);
        po_amax_step<8>(vb, jb);
      }
      if constexpr (GW == 32) {
        etot = bfly_add<16>(etot);
        po_amax_step<16>(vb, jb);
      }
      vb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vb)));
      jb = __builtin_amdgcn_readfirstlane(jb);
      etot = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(etot)));
      const unsigned long long wbal = __ballot(lane < GW && myj == jb);
      const int wb = wbal ? __ffsll((long long)wbal) - 1 : 0;
      if (wig == 0 && t == 0) {
        a.err_rec[(size_t)m * a.B + b] = etot;
        if (m == 0) a.orig[b] = vb;
        a.swaps[(size_t)b * a.max_rank + m] = jb;
      }
      const float piv = sqrtf(vb);  // :73-74
      const float* g = reinterpret_cast<const float*>(sh.gath[wb]) + PO_HDR;
      // the pivot row of C as register pairs {g[2 k], g[2 k + 1]}
      f2 g2[RC / 2];
#pragma unroll
      for (int i = 0; i < RC / 4; ++i) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + 4 * i);
        g2[2 * i] = (f2){g4.x, g4.y};
        g2[2 * i + 1] = (f2){g4.z, g4.w};
      }
      // Schur update of row m for the 4 rows of this thread: the products and sums of a row are a dependent chain in the
      // mandated order; the two rows of a pair share an instruction, the two pairs run in lockstep (computed for every
      // row, applied below only where the reference writes).  The L entries (of the pivot row: u; of this thread's rows:
      // sl) are requested two slots (four entries) at a time, a stage ahead of their use: the chains hide the LDS latency
      // and at most 72 registers hold operands in flight (VGPR budget: 128 of the 256 hold C).
      if (stamp) p4_stamp(a.dbg, 7, 8);
      if (!(a.prio & 2)) __builtin_amdgcn_s_setprio(0);
      f2 rowv[2], accs[2];
      int tp = t;
      asm volatile("" : "+v"(tp));  // (the 16 slot addresses are formed per pivot, not kept in registers across the loop)
      f2 u2[4];         // {u[2 k], u[2 k + 1]}, entries 0 .. 7
      float4 sl[2][4];  // slots 0 .. 3 of the two pair rows
#define LO_P4_FETCH(K0)                                                                      \
  if (2 * (K0) < m) {                                                                        \
    const float4 u4 = *reinterpret_cast<const float4*>(g + RC + 2 * (K0));                   \
    u2[K0] = (f2){u4.x, u4.y};                                                               \
    u2[(K0) + 1] = (f2){u4.z, u4.w};                                                         \
    sl[0][K0] = l_s[l_pslot<NS>(tp, K0)];                                                     \
    sl[1][K0] = l_s[l_pslot<NS>(tp + P4_TPB, K0)];                                            \
    if (2 * ((K0) + 1) < m) {                                                                \
      sl[0][(K0) + 1] = l_s[l_pslot<NS>(tp, (K0) + 1)];                                       \
      sl[1][(K0) + 1] = l_s[l_pslot<NS>(tp + P4_TPB, (K0) + 1)];                              \
    }                                                                                        \
  }
#define LO_P4_CHAIN(K)                                                                       \
  if (2 * (K) < m) {                                                                         \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                          \
      const f2 pr0 = pk_mul_bcast<0>((f2){sl[p][K].x, sl[p][K].y}, u2[K]);                   \
      accs[p] = ((K) == 0) ? pr0 : pk_add(accs[p], pr0);                                     \
    }                                                                                        \
    if (2 * (K) + 1 < m) {                                                                   \
      _Pragma("unroll") for (int p = 0; p < 2; ++p)                                          \
        accs[p] = pk_add(accs[p], pk_mul_bcast<1>((f2){sl[p][K].z, sl[p][K].w}, u2[K]));     \
    }                                                                                        \
  }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 2; ++p) rowv[p] = pk_mul_bcast<0>(Cq[p][0].lo, g2[0]);
#pragma unroll
      for (int r = 1; r < RC / 2; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          rowv[p] = pk_add(rowv[p], (r & 1) ? pk_mul_bcast<1>(Cq[p][r >> 1].hi, g2[r >> 1]) : pk_mul_bcast<0>(Cq[p][r >> 1].lo, g2[r >> 1]));
      __builtin_amdgcn_sched_barrier(0);
      LO_P4_FETCH(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = RC / 2; r < RC; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          rowv[p] = pk_add(rowv[p], (r & 1) ? pk_mul_bcast<1>(Cq[p][r >> 1].hi, g2[r >> 1]) : pk_mul_bcast<0>(Cq[p][r >> 1].lo, g2[r >> 1]));
      if (stamp) { asm volatile("" :: "v"(rowv[0]), "v"(rowv[1])); p4_stamp(a.dbg, 8, 9); }
#pragma unroll
      for (int p = 0; p < 2; ++p) accs[p] = (f2){0.f, 0.f};
      // :83-89, sequential in j
      __builtin_amdgcn_sched_barrier(0);
      LO_P4_FETCH(2)
      __builtin_amdgcn_sched_barrier(0);
      LO_P4_CHAIN(0)
      LO_P4_CHAIN(1)
      LO_P4_CHAIN(2)
      LO_P4_CHAIN(3)
#undef LO_P4_FETCH
#undef LO_P4_CHAIN
      // entries 8 ..: one slot (two entries) of the pivot row and of the two pair rows per step, requested a step ahead
      if (m > 8) {
        const int nk = (m + 1) >> 1;
        float2 un = *reinterpret_cast<const float2*>(g + RC + 8);
        float4 ln0 = l_s[l_pslot<NS>(tp, 4)], ln1 = l_s[l_pslot<NS>(tp + P4_TPB, 4)];
        for (int k = 4; k < nk; ++k) {
          const f2 uk = (f2){un.x, un.y};
          const float4 lq0 = ln0, lq1 = ln1;
          if (k + 1 < nk) {
            un = *reinterpret_cast<const float2*>(g + RC + 2 * (k + 1));
            ln0 = l_s[l_pslot<NS>(tp, k + 1)];
            ln1 = l_s[l_pslot<NS>(tp + P4_TPB, k + 1)];
          }
          accs[0] = pk_add(accs[0], pk_mul_bcast<0>((f2){lq0.x, lq0.y}, uk));
          accs[1] = pk_add(accs[1], pk_mul_bcast<0>((f2){lq1.x, lq1.y}, uk));
          if (2 * k + 1 < m) {
            accs[0] = pk_add(accs[0], pk_mul_bcast<1>((f2){lq0.z, lq0.w}, uk));
            accs[1] = pk_add(accs[1], pk_mul_bcast<1>((f2){lq1.z, lq1.w}, uk));
          }
        }
      }
      if (stamp) { asm volatile("" :: "v"(accs[0]), "v"(accs[1])); p4_stamp(a.dbg, 9, -1); }
      // write-back without divergent control flow: the four quotients are formed in lockstep (independent divisions),
      // the new L entry is ONE 4-byte LDS store into its 16-byte slot (no read-modify-write of the float4)
      const float rw4[P4_NR] = {rowv[0].x, rowv[0].y, rowv[1].x, rowv[1].y};
      const float ac4[P4_NR] = {accs[0].x, accs[0].y, accs[1].x, accs[1].y};
      float vq[P4_NR];
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) vq[q] = ((m > 0) ? rw4[q] - ac4[q] : rw4[q]) / piv;  // :91
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) {
        const int pq = pos[q];
        const bool live = pq != PO_INVALID;
        // permutation swap of positions m and jb (:67-70), tracked per row
        const int np = (pq == jb) ? m : ((pq == m) ? jb : pq);
        if (live) pos[q] = np;
        const float val = (np == m) ? piv : vq[q];  // the pivot row gets sqrt(max) (:73-74)
        if (live && np >= m)                         // already pivoted rows keep L[m] = 0
          reinterpret_cast<float*>(&l_s[l_pslot<NS>(tp + P4_TPB * (q >> 1), m >> 1)])[2 * (m & 1) + (q & 1)] = val;
        if (live && np > m) dg[q] = dg[q] - vq[q] * vq[q];  // :94-95
      }
      // sh.gath / sh.part4 are next written after the barrier that follows the candidate reduction
      if (stamp) p4_stamp(a.dbg, 6, -1);
    }
    __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[2] = wall_clock64();

    // the next member's rows are requested BEFORE this member's L rows leave (the registers of C are free from here on)
    int tn = t;
    asm volatile("" : "+v"(tn));
    {
      if (b + ngroups < a.B) {
        p4_issue_loads<RC>(a.C, b + ngroups, a.N, row0, tn, raw);
      } else {  // (defined on both paths: the old values must not stay live through the pivot loop)
#pragma unroll
        for (int q = 0; q < P4_NR; ++q)
#pragma unroll
          for (int i = 0; i < CH; ++i) raw[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // ---- L rows -> global, [max_rank, N] layout, consecutive threads = consecutive rows ----
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lr0 = tn + P4_TPB * (2 * p), lr1 = lr0 + P4_TPB;
      float* Lb = a.L + (size_t)b * a.max_rank * a.N + row0 + lr0;
#pragma unroll
      for (int h = 0; h < NS; h += 4) {
        float4 l4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) l4[k] = l_s[l_pslot<NS>(tn + P4_TPB * p, h + k)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m0 = 2 * (h + k);
          l4[k].x = (m0 < a.rank) ? l4[k].x : 0.f; l4[k].y = (m0 < a.rank) ? l4[k].y : 0.f;
          l4[k].z = (m0 + 1 < a.rank) ? l4[k].z : 0.f; l4[k].w = (m0 + 1 < a.rank) ? l4[k].w : 0.f;
        }
        if (lr0 < nv) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int m0 = 2 * (h + k);
            if (m0 < a.max_rank) Lb[(size_t)m0 * a.N] = l4[k].x;
            if (m0 + 1 < a.max_rank) Lb[(size_t)(m0 + 1) * a.N] = l4[k].z;
          }
        }
        if (lr1 < nv) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int m0 = 2 * (h + k);
            if (m0 < a.max_rank) Lb[(size_t)m0 * a.N + P4_TPB] = l4[k].y;
            if (m0 + 1 < a.max_rank) Lb[(size_t)(m0 + 1) * a.N + P4_TPB] = l4[k].w;
          }
        }
      }
    }
    __syncthreads();  // l_s / sh reuse by the next member
    if (stamp) a.dbg[3] = wall_clock64();
  }
}

// ---- the same resident factorisation for operators whose ROWS are fetched instead of formed from a resident root
// (round 4): DenseLinearOperator (row = K[pi, :], dense_linear_operator.py:47-50) and KroneckerProductLinearOperator of
// two dense factors (row[i] = K1[p1, i1] K2[p2, i2], kronecker_product_linear_operator.py:198-216).  The streaming engine
// re-read the m finished rows of L for every pivot and took two launches per pivot: 2.0 ms for the cfg4 shard (128 x
// 65536 rows, 15 pivots at 3.3 - 3.8 TB/s), 0.65 ms for the cfg5 shard, 0.4 ms for ONE dense operator of 4000 rows -- a
// chain of 30 launches in front of every solve.  Here, as in k_pc_onchip4, a member is a group of GW workgroups of 1024
// rows (GW up to 64: 65536 rows), a thread owns four rows with their running diagonal and permutation position in
// registers and their L entries in LDS; per pivot ONE exchange carries every workgroup's candidate (value, position,
// row index, error partial, its L entries 0 .. m-1); the winner's row of the operator is then fetched from memory
// (coalesced: consecutive threads = consecutive columns) and the Schur update runs in the streaming engine's operation
// order (-ffp-contract=off), so L and the permutation are bit-identical to it and to the CPU path.
constexpr int PR_SLOT = 40;  // granules per workgroup and parity: header + up to 32 L entries (+ pad)
constexpr int PR_SRC_DENSE = 1, PR_SRC_KRON = 2;

struct PrArgs {
  const float* A0;  // dense: K [B, N, N]; Kronecker: K1 [B, n1, n1]
  const float* A1;  // Kronecker: K2 [B, n2, n2]
  int n1, n2;
  int64_t B;
  int N, RW, rank, max_rank;
  float* L;        // [B, max_rank, N]
  float* err_rec;  // [rank, B]
  float* orig;     // [B]
  int* swaps;      // [B, max_rank]
  unsigned long long* gbuf;  // [ngroups][2][GW][PR_SLOT]
  int* err;
};

template <int GW>
struct alignas(16) PrShared {
  float wv[P4_WAVES];
  int wj[P4_WAVES];
  float we[P4_WAVES];
  int wi[P4_WAVES];
  unsigned part[PR_SLOT];
  unsigned gath[GW][PR_SLOT];
};

// One granule per polling thread (a thread that polls all GW workgroups' granules of its component issues GW loads per
// attempt: 7.3 us per pivot at GW = 64).  Phase 1: every workgroup publishes its whole payload, then thread (w, f) =
// (t / 4, t % 4) fetches header field f (value | position | row index | error partial) of workgroup w -- 4 GW <= 256
// threads, one load each.  Phase 2, after the winner is known: threads 0 .. m-1 fetch the WINNER's L entries only (the
// loads of the winner's row of the operator are issued first: their latency hides behind this poll).
__device__ __forceinline__ unsigned pr_poll(const unsigned long long* src, unsigned tag, int* err) {
  unsigned long long g = 0;
  unsigned spin = 0;
  for (;;) {
    g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(g >> 32) == tag) break;
    if (++spin > PO_MAXSPIN ||
        ((spin & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 1);  // timed out, or another workgroup already did: give up at once
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return (unsigned)(g & 0xffffffffull);
}

template <int GW>
__device__ __forceinline__ void pr_headers(PrShared<GW>& sh, int cnt, unsigned long long* slot, int wig, unsigned tag,
                                           int* err) {
  const int t = threadIdx.x;
  __syncthreads();  // sh.part complete
  if constexpr (GW == 1) {
    if (t < cnt) sh.gath[0][t] = sh.part[t];
  } else {
    if (t < cnt)
      __hip_atomic_store(slot + (size_t)wig * PR_SLOT + t, ((unsigned long long)tag << 32) | (unsigned long long)sh.part[t],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 4 * GW) {
      const int w = t >> 2, fld = t & 3;
      sh.gath[w][fld] = pr_poll(slot + (size_t)w * PR_SLOT + fld, tag, err);
    }
  }
  __syncthreads();
}

// L entries 0 .. m-1 (components PO_HDR ..) of workgroup wb -> sh.gath[wb]
template <int GW>
__device__ __forceinline__ void pr_winner(PrShared<GW>& sh, int m, unsigned long long* slot, int wb, unsigned tag, int* err) {
  const int t = threadIdx.x;
  if constexpr (GW > 1) {
    if (t < m) sh.gath[wb][PO_HDR + t] = pr_poll(slot + (size_t)wb * PR_SLOT + PO_HDR + t, tag, err);
  }
  __syncthreads();
}

template <int SRC, int GW, int LQ>
__global__ __launch_bounds__(P4_TPB, 2) void k_pc_onchip_rows(PrArgs a) {
  __shared__ PrShared<GW> sh;
  __shared__ float4 l_static[LQ == 4 ? P4_ROWS * 4 : 1];
  extern __shared__ float4 l_dynamic[];
  float4* const l_s = (LQ == 4) ? l_static : l_dynamic;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long* gslot = a.gbuf + (size_t)grp * 2 * GW * PR_SLOT;
  unsigned tag = 0;
  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));
  const int N = a.N;

  for (int64_t b = grp; b < a.B; b += ngroups) {
    float dg[P4_NR];
    int pos[P4_NR];
    const float* K0 = a.A0 + (size_t)b * (SRC == PR_SRC_DENSE ? (size_t)N * N : (size_t)a.n1 * a.n1);
    const float* K2 = (SRC == PR_SRC_KRON) ? a.A1 + (size_t)b * a.n2 * a.n2 : nullptr;
    // ---- initial diagonal (dense_linear_operator.py:37-40 / the product of the factors' diagonals), positions ----
#pragma unroll
    for (int q = 0; q < P4_NR; ++q) {
      const int lr = t + P4_TPB * q;
      const bool valid = lr < nv;
      const int i = row0 + lr;
      float v = 0.f;
      if (valid) {
        if (SRC == PR_SRC_DENSE) {
          v = K0[(size_t)i * N + i];
        } else {
          const int i1 = i / a.n2, i2 = i - i1 * a.n2;
          v = K0[(size_t)i1 * a.n1 + i1] * K2[(size_t)i2 * a.n2 + i2];
        }
      }
      dg[q] = v;
      pos[q] = valid ? i : PO_INVALID;
#pragma unroll
      for (int s4 = 0; s4 < LQ; ++s4) l_s[l_slot<LQ>(lr, s4)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int m = 0; m < a.rank; ++m) {
      // ---- workgroup candidate: argmax of the running diagonal over the positions >= m, error 1-norm partial ----
      float bv = -INFINITY, es = 0.f;
      int bj = PO_INVALID;
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) {
        const bool cand = pos[q] != PO_INVALID && pos[q] >= m;
        if (cand) {
          es += fabsf(dg[q]);
          if (po_better(dg[q], pos[q], bv, bj)) {
            bv = dg[q];
            bj = pos[q];
          }
        }
      }
      po_amax_step<1>(bv, bj); po_amax_step<2>(bv, bj); po_amax_step<4>(bv, bj);
      po_amax_step<8>(bv, bj); po_amax_step<16>(bv, bj); po_amax_step<32>(bv, bj);
      es = wave_sum_fast(es);
      if (lane == 0) {
        sh.wv[wave] = bv;
        sh.wj[wave] = bj;
        sh.we[wave] = es;
      }
      __syncthreads();
      float gv = sh.wv[lane & 3];
      int gj = sh.wj[lane & 3];
      float ge = sh.we[lane & 3];
      ge = bfly_add<1>(ge); ge = bfly_add<2>(ge);
      po_amax_step<1>(gv, gj); po_amax_step<2>(gv, gj);
      gv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gv)));
      gj = __builtin_amdgcn_readfirstlane(gj);
      ge = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ge)));
      // the owner of the candidate publishes it: header (value, position, row index, error) and its L entries 0..m-1
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) {
        if (pos[q] != PO_INVALID && pos[q] >= m && pos[q] == gj) {
          const int lr = t + P4_TPB * q;
          sh.part[0] = __float_as_uint(gv);
          sh.part[1] = (unsigned)gj;
          sh.part[2] = (unsigned)(row0 + lr);
          float4* dst = reinterpret_cast<float4*>(&sh.part[PO_HDR]);
          for (int j4 = 0; 4 * j4 < m; ++j4) dst[j4] = l_s[l_slot<LQ>(lr, j4)];
        }
      }
      if (t == 0) {
        if (gj == PO_INVALID) {
          sh.part[0] = __float_as_uint(-INFINITY);
          sh.part[1] = (unsigned)PO_INVALID;
          sh.part[2] = 0u;
        }
        sh.part[3] = __float_as_uint(ge);
      }
      ++tag;
      unsigned long long* slot = gslot + (size_t)(tag & 1u) * GW * PR_SLOT;
      pr_headers<GW>(sh, PO_HDR + m, slot, wig, tag, a.err);

      // ---- group winner (identical in all workgroups): lane l holds candidate l % GW ----
      float vb = __uint_as_float(sh.gath[lane % GW][0]);
      const int myj = (int)sh.gath[lane % GW][1];
      int jb = myj;
      float etot = __uint_as_float(sh.gath[lane % GW][3]);
      if constexpr (GW >= 2) { etot = bfly_add<1>(etot); po_amax_step<1>(vb, jb); }
      if constexpr (GW >= 4) { etot = bfly_add<2>(etot); po_amax_step<2>(vb, jb); }
      if constexpr (GW >= 8) { etot = bfly_add<4>(etot); po_amax_step<4>(vb, jb); }
      if constexpr (GW >= 16) { etot = bfly_add<8>(etot); po_amax_step<8>(vb, jb); }
      if constexpr (GW >= 32) { etot = bfly_add<16>(etot); po_amax_step<16>(vb, jb); }
      if constexpr (GW >= 64) { etot = bfly_add<32>(etot); po_amax_step<32>(vb, jb); }
      vb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vb)));
      jb = __builtin_amdgcn_readfirstlane(jb);
      etot = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(etot)));
      const unsigned long long wbal = __ballot(lane < GW && myj == jb);
      const int wb = wbal ? __ffsll((long long)wbal) - 1 : 0;
      const int pim = __builtin_amdgcn_readfirstlane((int)sh.gath[wb][2]);  // the winner's row of the operator
      if (wig == 0 && t == 0) {
        a.err_rec[(size_t)m * a.B + b] = etot;
        if (m == 0) a.orig[b] = vb;
        a.swaps[(size_t)b * a.max_rank + m] = jb;
      }
      const float piv = sqrtf(vb);  // :73-74
      const float* g = reinterpret_cast<const float*>(sh.gath[wb]) + PO_HDR;
      // ---- row pi_m of the operator at this thread's four columns ----
      float rowv[P4_NR], accs[P4_NR];
      if (SRC == PR_SRC_DENSE) {
        const float* kr = K0 + (size_t)pim * N + row0;
#pragma unroll
        for (int q = 0; q < P4_NR; ++q) rowv[q] = (t + P4_TPB * q < nv) ? kr[t + P4_TPB * q] : 0.f;
      } else {
        const int p1 = pim / a.n2, p2 = pim - p1 * a.n2;
        const float* k1r = K0 + (size_t)p1 * a.n1;
        const float* k2r = K2 + (size_t)p2 * a.n2;
#pragma unroll
        for (int q = 0; q < P4_NR; ++q) {
          const int i = min(row0 + t + P4_TPB * q, N - 1);
          const int i1 = i / a.n2, i2 = i - i1 * a.n2;
          rowv[q] = k1r[i1] * k2r[i2];
        }
      }
      pr_winner<GW>(sh, m, slot, wb, tag, a.err);  // (the row loads above are in flight)
      // ---- Schur update, :83-89: products and sums sequential in j (the four rows in lockstep) ----
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) accs[q] = 0.f;
      for (int j4 = 0; 4 * j4 < m; ++j4) {
        const float4 u4 = *reinterpret_cast<const float4*>(g + 4 * j4);
        const int j = 4 * j4;
        float4 l4[P4_NR];
#pragma unroll
        for (int q = 0; q < P4_NR; ++q) l4[q] = l_s[l_slot<LQ>(t + P4_TPB * q, j4)];
#pragma unroll
        for (int q = 0; q < P4_NR; ++q) accs[q] = (j == 0) ? u4.x * l4[q].x : accs[q] + u4.x * l4[q].x;
        if (j + 1 < m) {
#pragma unroll
          for (int q = 0; q < P4_NR; ++q) accs[q] = accs[q] + u4.y * l4[q].y;
        }
        if (j + 2 < m) {
#pragma unroll
          for (int q = 0; q < P4_NR; ++q) accs[q] = accs[q] + u4.z * l4[q].z;
        }
        if (j + 3 < m) {
#pragma unroll
          for (int q = 0; q < P4_NR; ++q) accs[q] = accs[q] + u4.w * l4[q].w;
        }
      }
      const int ms = m >> 2, me = m & 3;
      float vq[P4_NR];
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) vq[q] = ((m > 0) ? rowv[q] - accs[q] : rowv[q]) / piv;  // :91
#pragma unroll
      for (int q = 0; q < P4_NR; ++q) {
        const int pq = pos[q];
        const bool live = pq != PO_INVALID;
        const int np = (pq == jb) ? m : ((pq == m) ? jb : pq);  // permutation swap of positions m and jb (:67-70)
        if (live) pos[q] = np;
        const float val = (np == m) ? piv : vq[q];  // the pivot row gets sqrt(max) (:73-74)
        if (live && np >= m) reinterpret_cast<float*>(&l_s[l_slot<LQ>(t + P4_TPB * q, ms)])[me] = val;
        if (live && np > m) dg[q] = dg[q] - vq[q] * vq[q];  // :94-95
      }
    }
    // ---- L rows -> global, [max_rank, N] layout, consecutive threads = consecutive rows ----
#pragma unroll
    for (int q = 0; q < P4_NR; ++q) {
      const int lr = t + P4_TPB * q;
      if (lr < nv) {
        float* Lb = a.L + (size_t)b * a.max_rank * N + row0 + lr;
#pragma unroll
        for (int j4 = 0; j4 < LQ; ++j4) {
          const float4 l4 = l_s[l_slot<LQ>(lr, j4)];
          const int m0 = 4 * j4;
          if (m0 < a.max_rank) Lb[(size_t)m0 * N] = (m0 < a.rank) ? l4.x : 0.f;
          if (m0 + 1 < a.max_rank) Lb[(size_t)(m0 + 1) * N] = (m0 + 1 < a.rank) ? l4.y : 0.f;
          if (m0 + 2 < a.max_rank) Lb[(size_t)(m0 + 2) * N] = (m0 + 2 < a.rank) ? l4.z : 0.f;
          if (m0 + 3 < a.max_rank) Lb[(size_t)(m0 + 3) * N] = (m0 + 3 < a.rank) ? l4.w : 0.f;
        }
      }
    }
    __syncthreads();  // l_s / sh reuse by the next member
  }
}

// m* = number of pivots the reference takes: pivot 0 always, pivot m >= 1 while max_b error_{m-1} > tol (:57, :99).
// Wave w evaluates the pivots m = w + 1, w + 5, ... (max over the members with wave butterflies, no barriers inside);
// thread 0 then takes the first failing m.
// (Evaluated by EVERY workgroup of k_po_perm -- 14 x B error values from L2 -- instead of a one-workgroup kernel in
// front of it: one launch and its gap less on a path where every launch is ~1 % of the factorisation.)
__device__ __forceinline__ int po_rank(const PoArgs& a, float tol, int* cont_s, int* mstar_s) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int m = 1 + wave; m < a.rank; m += kThreads / 64) {
    float lmax = -INFINITY, lnan = 0.f;
    for (int64_t b = lane; b < a.B; b += 64) {
      const float e = a.err_rec[(size_t)m * a.B + b] / a.orig[b];
      if (e != e) lnan = 1.f;
      lmax = fmaxf(lmax, e);
    }
    const float mx = wave_max(lmax);
    const float anynan = wave_sum(lnan);
    // torch.max propagates NaN and (NaN > tol) is False -> the reference stops
    if (lane == 0) cont_s[m] = ((anynan == 0.f) && (mx > tol)) ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int mstar = a.rank;
    for (int m = 1; m < a.rank; ++m)
      if (!cont_s[m]) {
        mstar = m;
        break;
      }
    *mstar_s = mstar;
  }
  __syncthreads();
  return *mstar_s;
}

__global__ __launch_bounds__(kThreads) void k_po_perm(PoArgs a, float tol, int* __restrict__ m_out,
                                                       long long* __restrict__ perm) {
  __shared__ int cont_s[P4_MAXR + 1];
  __shared__ int mstar_s;
  const int64_t b = blockIdx.x;
  const int mstar = po_rank(a, tol, cont_s, &mstar_s);
  if (b == 0 && threadIdx.x == 0) *m_out = mstar;
  long long* pb = perm + (size_t)b * a.N;
  for (int i = threadIdx.x; i < a.N; i += kThreads) pb[i] = i;
  if (mstar < a.rank) {
    float* Lb = a.L + (size_t)b * a.max_rank * a.N;
    for (size_t e = (size_t)mstar * a.N + threadIdx.x; e < (size_t)a.rank * a.N; e += kThreads) Lb[e] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int m = 0; m < mstar; ++m) {
      const int j = a.swaps[(size_t)b * a.max_rank + m];
      const long long x = pb[m];
      pb[m] = pb[j];
      pb[j] = x;
    }
  }
}

bool pc_onchip_eligible(const lo_op_desc* op, int max_rank) {
  if (resident_off() || op->kind != LO_OP_LOWRANK_DIAG) return false;
  const int64_t R = op->R;  // (any rank up to 32: zero-padded to 8 / 16 / 32 columns in the workspace)
  // (rank 17 .. 32: second generation only, one workgroup per CU -- needs a full group of the member's size)
  return R >= 1 && R <= 32 && max_rank <= P4_MAXR && op->N >= 256 &&
         op->N <= (int64_t)32 * P4_ROWS && onchip_num_workgroups() >= 64;
}

static int po_padded_rank(int64_t R) { return R <= 8 ? 8 : (R <= 16 ? 16 : 32); }

struct PoLayout {
  float* cpad;  // [B, N, RP] zero-padded copy of C when R is not 8 / 16 / 32
  float* err_rec;
  float* orig;
  int* swaps;
  unsigned long long* gbuf;
  int* err;
  int* m_out;
  long long* dbg;
};

static void po_layout(const lo_op_desc* op, int max_rank, Arena& ar, PoLayout* l) {
  const int64_t B = op->B;
  const int RP = po_padded_rank(op->R);
  l->cpad = (RP != op->R) ? ar.take<float>((size_t)B * op->N * RP) : nullptr;
  l->err = ar.take<int>(4);
  l->m_out = l->err + 1;
  l->err_rec = ar.take<float>((size_t)max_rank * B);
  l->orig = ar.take<float>(B);
  l->swaps = ar.take<int>((size_t)B * max_rank);
  l->gbuf = ar.take<unsigned long long>((size_t)64 * 2 * PO_GW * PO_SLOT);
  l->dbg = ar.take<long long>(12);
}

size_t pc_onchip_workspace_bytes(const lo_op_desc* op, int max_rank) {
  Arena ar(nullptr, 0);
  PoLayout l;
  po_layout(op, max_rank, ar, &l);
  return ar.off + 1024;
}

int pc_onchip_run(const lo_op_desc* op, int rank, int max_rank, float tol, float* L_rows, long long* perm,
                  int32_t* rank_out, void* ws, size_t ws_bytes, hipStream_t st) {
  Arena ar(ws, ws_bytes);
  PoLayout l;
  po_layout(op, max_rank, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const int RP = po_padded_rank(op->R);
  const float* Csrc = op->A0;
  if (RP != op->R) {
    const int rc = pad_rows(op->A0, (int)op->R, l.cpad, RP, op->B * op->N, st);
    if (rc) return rc;
    Csrc = l.cpad;
  }
  const int nwg = onchip_num_workgroups();
  // second generation (4 rows per thread, two workgroups per CU) when two workgroups fit on a CU
  // (smallest group that holds the member: small members no longer occupy 8 mostly idle workgroups)
  int gw2 = 32;
  for (int gq = 1; gq < 32; gq *= 2)
    if (op->N <= (int64_t)gq * P4_ROWS) {
      gw2 = gq;
      break;
    }
  if (getenv("LO_OC_GW8")) gw2 = std::max(gw2, 8);
  const bool wide = max_rank > PO_MAXR;  // rank 17 .. 32: 128 KB of L rows per workgroup, one workgroup per CU
  const int wpc = wide ? 1 : 2;          // workgroups per CU the launch relies on
  const size_t dyn_lds = wide ? sizeof(float4) * (size_t)P4_ROWS * 8 : 0;
  bool gen2 = true;
  {
    int per_cu = 0;
    hipError_t e = hipErrorUnknown;
#define LO_OCC(R_, G_)                                                                                               \
  if (wide) {                                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc_onchip4<R_, G_, 8>),                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);                               \
    if (e == hipSuccess)                                                                                             \
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pc_onchip4<R_, G_, 8>, P4_TPB, dyn_lds);           \
  } else {                                                                                                           \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pc_onchip4<R_, G_, 4>, P4_TPB, 0);                   \
  }
#define LO_OCC_R(R_)                       \
  switch (gw2) {                           \
    case 1: LO_OCC(R_, 1) break;           \
    case 2: LO_OCC(R_, 2) break;           \
    case 4: LO_OCC(R_, 4) break;           \
    case 8: LO_OCC(R_, 8) break;           \
    case 16: LO_OCC(R_, 16) break;         \
    default: LO_OCC(R_, 32) break;         \
  }
    if (RP == 32) { LO_OCC_R(32) } else if (RP == 16) { LO_OCC_R(16) } else { LO_OCC_R(8) }
#undef LO_OCC_R
#undef LO_OCC
    // every XCD must hold at least one whole group, all of its workgroups resident at once
    gen2 = (e == hipSuccess) && per_cu >= wpc && (wpc * nwg / 8) / gw2 >= 1;
  }
  if (!gen2) return LO_ERR_LAUNCH;  // two workgroups per CU do not fit this device: the caller runs the streaming engine
  const int gw = gw2;
  PoArgs a;
  a.C = Csrc;
  a.B = op->B;
  a.N = (int)op->N;
  a.RW = (int)((op->N + gw - 1) / gw);
  a.rank = rank;
  a.max_rank = max_rank;
  {
    const char* e = getenv("LO_PC_PRIO");
    a.prio = e ? atoi(e) : 1;
  }
  a.L = L_rows;
  a.err_rec = l.err_rec;
  a.orig = l.orig;
  a.swaps = l.swaps;
  a.gbuf = l.gbuf;
  a.err = l.err;
  a.allow_l2_handoff = onchip_l2_handoff_allowed();
  const bool debug = getenv("LO_OC_DEBUG") != nullptr;
  a.dbg = debug ? l.dbg : nullptr;
  if (debug) LO_HIP_CHECK(hipMemsetAsync(l.dbg, 0, 12 * sizeof(long long), st));
  LO_HIP_CHECK(hipMemsetAsync(l.err, 0, 4 * sizeof(int), st));
  if (getenv("LO_OC_TEST_FALLBACK")) LO_HIP_CHECK(hipMemsetAsync(l.err, 1, 1, st));  // as if an exchange had timed out
  LO_HIP_CHECK(hipMemsetAsync(l.gbuf, 0, sizeof(unsigned long long) * (size_t)64 * 2 * PO_GW * PO_SLOT, st));
  LO_PROF_BEGIN("pc_onchip", st);
  {
  ResidentLaunch guard(st);
  {
    dim3 grid2(wpc * nwg), block2(P4_TPB);
#define LO_GO(R_, G_)                                                                      \
  if (wide) hipLaunchKernelGGL((k_pc_onchip4<R_, G_, 8>), grid2, block2, dyn_lds, st, a);  \
  else hipLaunchKernelGGL((k_pc_onchip4<R_, G_, 4>), grid2, block2, 0, st, a)
#define LO_GO_R(R_)                      \
  switch (gw) {                          \
    case 1: LO_GO(R_, 1); break;         \
    case 2: LO_GO(R_, 2); break;         \
    case 4: LO_GO(R_, 4); break;         \
    case 8: LO_GO(R_, 8); break;         \
    case 16: LO_GO(R_, 16); break;       \
    default: LO_GO(R_, 32); break;       \
  }
    if (RP == 32) { LO_GO_R(32) } else if (RP == 16) { LO_GO_R(16) } else { LO_GO_R(8) }
#undef LO_GO_R
#undef LO_GO
  }
  }
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  LO_PROF_BEGIN("pc_onchip_perm", st);
  hipLaunchKernelGGL(k_po_perm, dim3((unsigned)op->B), dim3(kThreads), 0, st, a, tol, l.m_out, perm);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  int h[2];
  {
    void* hp = pinned_status_block();
    LO_HIP_CHECK(hipMemcpyAsync(hp ? hp : h, l.err, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LO_HIP_CHECK(hipStreamSynchronize(st));
    if (hp) memcpy(h, hp, sizeof(h));
  }
  if (debug) {
    long long ts[12];
    LO_HIP_CHECK(hipMemcpy(ts, l.dbg, sizeof(ts), hipMemcpyDeviceToHost));
    fprintf(stderr, "pc_onchip member0 (100 MHz ticks): load %lld pivots %lld store %lld | reduce+publish %lld gather %lld update %lld\n",
            ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4], ts[5], ts[6]);
    fprintf(stderr, "  update split: winner+row fetch %lld, C.C chain %lld, L.L chain %lld\n", ts[7], ts[8], ts[9]);
  }
  if (h[0]) {
    onchip_note_timeout();
    return LO_ERR_LAUNCH;
  }
  *rank_out = h[1];
  return LO_OK;
}

// ---- host side of k_pc_onchip_rows ----------------------------------------------------------------------------------
bool pc_onchip_rows_eligible(const lo_op_desc* op, int max_rank) {
  if (resident_off() || getenv("LO_PC_NO_RESIDENT_ROWS")) return false;
  if (op->kind == LO_OP_DENSE_DIAG) {
    if (!op->A0) return false;
  } else if (op->kind == LO_OP_KRON_DIAG) {
    if (!op->A0 || !op->A1 || op->R < 1 || op->n2 < 1 || op->R * op->n2 != op->N) return false;
  } else {
    return false;
  }
  return max_rank <= P4_MAXR && op->N >= 64 && op->N <= (int64_t)64 * P4_ROWS && onchip_num_workgroups() >= 64;
}

struct PrLayout {
  float* err_rec;
  float* orig;
  int* swaps;
  unsigned long long* gbuf;
  int* err;
  int* m_out;
};

static void pr_layout(const lo_op_desc* op, int max_rank, Arena& ar, PrLayout* l) {
  const int64_t B = op->B;
  l->err = ar.take<int>(4);
  l->m_out = l->err + 1;
  l->err_rec = ar.take<float>((size_t)max_rank * B);
  l->orig = ar.take<float>(B);
  l->swaps = ar.take<int>((size_t)B * max_rank);
  l->gbuf = ar.take<unsigned long long>((size_t)512 * 2 * PR_SLOT);
}

size_t pc_onchip_rows_workspace_bytes(const lo_op_desc* op, int max_rank) {
  Arena ar(nullptr, 0);
  PrLayout l;
  pr_layout(op, max_rank, ar, &l);
  return ar.off + 1024;
}

template <int SRC, int GW>
static int pr_go(const PrArgs& a, bool wide, int nwg, hipStream_t st) {
  const int wpc = wide ? 1 : 2;
  const size_t dyn_lds = wide ? sizeof(float4) * (size_t)P4_ROWS * 8 : 0;
  int per_cu = 0;
  hipError_t e;
  if (wide) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc_onchip_rows<SRC, GW, 8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
    if (e == hipSuccess) e = LO_OCCUPANCY_CACHED(per_cu, (k_pc_onchip_rows<SRC, GW, 8>), P4_TPB, dyn_lds);
  } else {
    e = LO_OCCUPANCY_CACHED(per_cu, (k_pc_onchip_rows<SRC, GW, 4>), P4_TPB, 0);
  }
  // every XCD must hold at least one whole group, all of its workgroups resident at once
  if (e != hipSuccess || per_cu < wpc || (wpc * nwg / 8) / GW < 1) return LO_ERR_LAUNCH;
  LO_PROF_BEGIN("pc_onchip_rows", st);
  {
    ResidentLaunch guard(st);
    if (wide) hipLaunchKernelGGL((k_pc_onchip_rows<SRC, GW, 8>), dim3(wpc * nwg), dim3(P4_TPB), dyn_lds, st, a);
    else hipLaunchKernelGGL((k_pc_onchip_rows<SRC, GW, 4>), dim3(wpc * nwg), dim3(P4_TPB), 0, st, a);
  }
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int SRC>
static int pr_go_gw(int gw, const PrArgs& a, bool wide, int nwg, hipStream_t st) {
  switch (gw) {
    case 1: return pr_go<SRC, 1>(a, wide, nwg, st);
    case 2: return pr_go<SRC, 2>(a, wide, nwg, st);
    case 4: return pr_go<SRC, 4>(a, wide, nwg, st);
    case 8: return pr_go<SRC, 8>(a, wide, nwg, st);
    case 16: return pr_go<SRC, 16>(a, wide, nwg, st);
    case 32: return pr_go<SRC, 32>(a, wide, nwg, st);
    default: return pr_go<SRC, 64>(a, wide, nwg, st);
  }
}

// returns LO_ERR_LAUNCH when the kernel does not fit or an exchange timed out (caller falls back to the streaming engine)
int pc_onchip_rows_run(const lo_op_desc* op, int rank, int max_rank, float tol, float* L_rows, long long* perm,
                       int32_t* rank_out, void* ws, size_t ws_bytes, hipStream_t st) {
  Arena ar(ws, ws_bytes);
  PrLayout l;
  pr_layout(op, max_rank, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const int nwg = onchip_num_workgroups();
  int gw = 64;
  for (int gq = 1; gq < 64; gq *= 2)
    if (op->N <= (int64_t)gq * P4_ROWS) {
      gw = gq;
      break;
    }
  const bool wide = max_rank > PO_MAXR;  // rank 17 .. 32: 128 KB of L rows per workgroup, one workgroup per CU
  PrArgs a;
  a.A0 = op->A0; a.A1 = op->A1;
  a.n1 = (int)op->R; a.n2 = (int)op->n2;
  a.B = op->B; a.N = (int)op->N;
  a.RW = (int)((op->N + gw - 1) / gw);
  a.rank = rank; a.max_rank = max_rank;
  a.L = L_rows; a.err_rec = l.err_rec; a.orig = l.orig; a.swaps = l.swaps; a.gbuf = l.gbuf; a.err = l.err;
  LO_HIP_CHECK(hipMemsetAsync(l.err, 0, 4 * sizeof(int), st));
  if (getenv("LO_OC_TEST_FALLBACK")) LO_HIP_CHECK(hipMemsetAsync(l.err, 1, 1, st));  // as if an exchange had timed out
  {
    const int zrc = zero_span(l.gbuf, sizeof(unsigned long long) * (size_t)512 * 2 * PR_SLOT, st);
    if (zrc) return zrc;
  }
  const int rc = (op->kind == LO_OP_DENSE_DIAG) ? pr_go_gw<PR_SRC_DENSE>(gw, a, wide, nwg, st)
                                                : pr_go_gw<PR_SRC_KRON>(gw, a, wide, nwg, st);
  if (rc) return rc;
  PoArgs pa;  // (k_po_perm reads the recorded errors / swaps and the factor)
  memset(&pa, 0, sizeof(pa));
  pa.B = op->B; pa.N = (int)op->N; pa.rank = rank; pa.max_rank = max_rank;
  pa.L = L_rows; pa.err_rec = l.err_rec; pa.orig = l.orig; pa.swaps = l.swaps;
  LO_PROF_BEGIN("pc_onchip_perm", st);
  hipLaunchKernelGGL(k_po_perm, dim3((unsigned)op->B), dim3(kThreads), 0, st, pa, tol, l.m_out, perm);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  int h[2];
  {
    void* hp = pinned_status_block();
    LO_HIP_CHECK(hipMemcpyAsync(hp ? hp : h, l.err, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    LO_HIP_CHECK(hipStreamSynchronize(st));
    if (hp) memcpy(h, hp, sizeof(h));
  }
  if (h[0]) {
    onchip_note_timeout();
    return LO_ERR_LAUNCH;
  }
  *rank_out = h[1];
  return LO_OK;
}

}  // namespace lo

// lo_pivchol_onchip.hip -- operator-resident pivoted Cholesky for the low-rank-root + diagonal operator
// (PivotedCholesky.forward, linear_operator/functions/_pivoted_cholesky.py:14-105, row source
// RootLinearOperator._get_indices root_linear_operator.py:37-50 / _diagonal :22-28).
//
// The streaming engine (lo_pivchol.hip) re-reads C (4NR bytes) and the m finished rows of L (4mN bytes) from HBM
// for every pivot.  Here one group of 8 workgroups x 1024 threads owns one batch member for ALL pivots: thread =
// one row i of the operator, its C row sits in LDS, its running diagonal, its position in the permutation and
// its L entries L[0..m-1][i] sit in registers.  HBM traffic per member: C once (4NR) + L once (4 max_rank N).
// Per pivot the group needs ONE exchange (8-byte {value, tag} granules, same mechanism as lo_cg_onchip.hip):
// every workgroup publishes its best candidate (diagonal value, position, row), that row's C row and L entries,
// and its partial of the error 1-norm; everybody then picks the same winner and has all it needs for the Schur
// update of its own rows.
//
// Every operation that feeds a pivot decision is the same individually rounded, fixed-order arithmetic as
// lo_pivchol.hip / oracle.pivoted_cholesky (file compiled with -ffp-contract=off), so L and the permutation
// are bit-identical to the streaming engine and to the CPU path.
//
// The batch-global stopping rule (:57: one shared m, stop when max_b error <= tol) cannot be evaluated inside
// a kernel whose groups work on different members at different times, so the kernel takes all `rank` pivots and
// records the error after each; k_po_rank then finds the reference's m*, and k_po_perm rebuilds the permutation
// from the recorded swaps 0..m*-1 and clears the rows of L the reference would not have written.
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang fp contract(off)

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int PO_TPB = 1024;
constexpr int PO_GW = 8;
constexpr int PO_WAVES = PO_TPB / 64;
constexpr int PO_CLD = 36;    // LDS row stride of C (floats)
constexpr int PO_SLOT = 64;   // granules per workgroup and parity
constexpr int PO_MAXR = 16;   // pivots held in registers
constexpr int PO_HDR = 4;     // value, position, (unused), error partial
constexpr unsigned PO_MAXSPIN = 1u << 22;
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
  long long* dbg;  // optional phase timers (wall_clock64 ticks) of member 0 / workgroup 0, or nullptr
};

struct PoShared {
  float wv[PO_WAVES];
  int wj[PO_WAVES];
  float we[PO_WAVES];
  unsigned part[PO_SLOT];
  unsigned gath[PO_GW][PO_SLOT];
};

__device__ __forceinline__ bool po_better(float ov, int oj, float mv, int mj) {
  // FIRST maximal position wins (torch.max on CPU, :61-63)
  return oj != PO_INVALID && (mj == PO_INVALID || ov > mv || (ov == mv && oj < mj));
}

// All workgroups of the group publish sh.part[0..cnt) and receive everybody's in sh.gath[w][0..cnt).
__device__ __forceinline__ void po_gather(PoShared& sh, int cnt, unsigned long long* gslot_base, int wig, unsigned tag,
                                          int* err, bool same_xcd) {
  const int t = threadIdx.x;
  unsigned long long* slot = gslot_base + (size_t)(tag & 1u) * PO_GW * PO_SLOT;
  if (t < cnt) {
    const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)sh.part[t];
    if (same_xcd)
      __hip_atomic_store(slot + (size_t)wig * PO_SLOT + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)wig * PO_SLOT + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int w = t >> 6, i = t & 63;  // wave w polls workgroup w's granules (PO_SLOT == 64)
  if (w < PO_GW && i < cnt) {
    const unsigned long long* src = slot + (size_t)w * PO_SLOT + i;
    unsigned long long g = 0;
    unsigned spin = 0;
    for (;;) {
      g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(g >> 32) == tag) break;
      if (++spin > PO_MAXSPIN) {
        atomicExch(err, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    sh.gath[w][i] = (unsigned)(g & 0xffffffffull);
  }
  __syncthreads();
}

// Per-row state of the factorisation (one thread = one row of the operator).
struct PoRow {
  float dg;   // running diagonal
  int pos;    // position of this row in the permutation
  int row;
  bool valid;
};

// One pivot.  M is a template parameter so that every index into the register array Lr is a compile-time
// constant (a runtime-indexed register array would be demoted to scratch memory).
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

template <int RC, int m>
__device__ __forceinline__ void po_pivot(const PoArgs& a, PoShared& sh, const float* c_s, const float* crow,
                                         float (&Lr)[PO_MAXR], PoRow& st, int64_t b, unsigned long long* gslot,
                                         int wig, unsigned& tag, bool same_xcd) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool stamp = a.dbg && b == 0 && wig == 0 && t == 0;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (stamp) c0 = wall_clock64();
  const bool valid = st.valid;
  float dg = st.dg;
  int pos = st.pos;
  // ---- workgroup candidate: argmax of the running diagonal over the positions >= m, error 1-norm partial ----
  const bool cand = valid && pos >= m;
  float bv = cand ? dg : -INFINITY;
  int bj = cand ? pos : PO_INVALID;
  po_amax_step<1>(bv, bj); po_amax_step<2>(bv, bj); po_amax_step<4>(bv, bj);
  po_amax_step<8>(bv, bj); po_amax_step<16>(bv, bj); po_amax_step<32>(bv, bj);
  const float es = wave_sum_fast(cand ? fabsf(dg) : 0.f);
  if (lane == 0) {
    sh.wv[wave] = bv;
    sh.wj[wave] = bj;
    sh.we[wave] = es;
  }
  __syncthreads();
  // every wave finishes the reduction over the 16 wave partials itself (lane l holds partial l & 15)
  float gv = sh.wv[lane & 15];
  int gj = sh.wj[lane & 15];
  float ge = lanes16_sum(sh.we[lane & 15]);
  po_amax_step<1>(gv, gj); po_amax_step<2>(gv, gj); po_amax_step<4>(gv, gj); po_amax_step<8>(gv, gj);
  gv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gv)));
  gj = __builtin_amdgcn_readfirstlane(gj);
  ge = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ge)));
  // the wave holding the candidate publishes it: header, its C row (copied from LDS by RC lanes), its L entries
  const bool mine = cand && pos == gj;
  const unsigned long long bal = __ballot(mine);
  if (bal != 0ull) {
    const int src = __ffsll((long long)bal) - 1;
    if (lane < RC) sh.part[PO_HDR + lane] = __float_as_uint(c_s[(wave * 64 + src) * PO_CLD + lane]);
    if (mine) {
      sh.part[0] = __float_as_uint(gv);
      sh.part[1] = (unsigned)gj;
#pragma unroll
      for (int j = 0; j < m; ++j) sh.part[PO_HDR + RC + j] = __float_as_uint(Lr[j]);
    }
  }
  if (t == 0) {
    if (gj == PO_INVALID) {
      sh.part[0] = __float_as_uint(-INFINITY);
      sh.part[1] = (unsigned)PO_INVALID;
    }
    sh.part[3] = __float_as_uint(ge);
  }
  __syncthreads();
  if (stamp) c1 = wall_clock64();
  po_gather(sh, PO_HDR + RC + m, gslot, wig, ++tag, a.err, same_xcd);
  if (stamp) c2 = wall_clock64();

  // ---- group winner (identical in all 8 workgroups): lane l holds candidate l & 7 ----
  float vb = __uint_as_float(sh.gath[lane & 7][0]);
  const int myj = (int)sh.gath[lane & 7][1];
  int jb = myj;
  float etot = lanes8_sum(__uint_as_float(sh.gath[lane & 7][3]));
  po_amax_step<1>(vb, jb); po_amax_step<2>(vb, jb); po_amax_step<4>(vb, jb);
  vb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vb)));
  jb = __builtin_amdgcn_readfirstlane(jb);
  etot = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(etot)));
  const unsigned long long wbal = __ballot(lane < PO_GW && myj == jb);
  const int wb = wbal ? __ffsll((long long)wbal) - 1 : 0;
  if (wig == 0 && t == 0) {
    a.err_rec[(size_t)m * a.B + b] = etot;
    if (m == 0) a.orig[b] = vb;
    a.swaps[(size_t)b * a.max_rank + m] = jb;
  }
  // permutation swap of positions m and jb (:67-70), tracked per row
  if (valid) {
    if (pos == jb) pos = m;
    else if (pos == m) pos = jb;
  }
  const float piv = sqrtf(vb);  // :73-74
  if (valid) {
    if (pos == m) {
      Lr[m] = piv;
    } else if (pos > m) {  // Schur update of row m at the not yet pivoted rows (:77-95)
      const float* g = reinterpret_cast<const float*>(sh.gath[wb]) + PO_HDR;
      float rowv = 0.f;
#pragma unroll
      for (int q = 0; q < RC / 4; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + 4 * q);
        const float4 c4 = *reinterpret_cast<const float4*>(crow + 4 * q);
        rowv = (q == 0) ? g4.x * c4.x : rowv + g4.x * c4.x;
        rowv = rowv + g4.y * c4.y;
        rowv = rowv + g4.z * c4.z;
        rowv = rowv + g4.w * c4.w;
      }
      float v = rowv;
      if constexpr (m > 0) {
        float u[PO_MAXR];
#pragma unroll
        for (int q = 0; q < (m + 3) / 4; ++q) {
          const float4 u4 = *reinterpret_cast<const float4*>(g + RC + 4 * q);
          u[4 * q] = u4.x; u[4 * q + 1] = u4.y; u[4 * q + 2] = u4.z; u[4 * q + 3] = u4.w;
        }
        float acc = u[0] * Lr[0];
#pragma unroll
        for (int j = 1; j < m; ++j) acc = acc + u[j] * Lr[j];
        v = rowv - acc;
      }
      v = v / piv;
      Lr[m] = v;
      dg = dg - v * v;
    }
  }
  // sh.gath is next written two barriers from here (candidate reduction, publication): no extra barrier
  st.dg = dg;
  st.pos = pos;
  if (stamp) {
    c3 = wall_clock64();
    a.dbg[4] += c1 - c0;
    a.dbg[5] += c2 - c1;
    a.dbg[6] += c3 - c2;
  }
}

template <int RC, int m>
__device__ __forceinline__ void po_pivots(const PoArgs& a, PoShared& sh, const float* c_s, const float* crow,
                                          float (&Lr)[PO_MAXR], PoRow& st, int64_t b, unsigned long long* gslot,
                                          int wig, unsigned& tag, bool same_xcd) {
  if constexpr (m < PO_MAXR) {
    if (m < a.rank) {
      po_pivot<RC, m>(a, sh, c_s, crow, Lr, st, b, gslot, wig, tag, same_xcd);
      po_pivots<RC, m + 1>(a, sh, c_s, crow, Lr, st, b, gslot, wig, tag, same_xcd);
    }
  }
}

template <int RC>
__global__ __launch_bounds__(PO_TPB) void k_pc_onchip(PoArgs a) {
  __shared__ PoShared sh;
  __shared__ float c_s[PO_TPB * PO_CLD];
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;  // block b runs on XCD b % 8: keep a group behind one L2 (speed only)
  const int groups_per_xcd = (gridDim.x / 8) / PO_GW;
  const int grp = xcd * groups_per_xcd + jx / PO_GW;
  const int wig = jx % PO_GW;
  const int ngroups = gridDim.x / PO_GW;
  if (jx / PO_GW >= groups_per_xcd) return;
  const int t = threadIdx.x;
  unsigned long long* gslot = a.gbuf + (size_t)grp * 2 * PO_GW * PO_SLOT;
  unsigned tag = 0;
  bool same_xcd = false;
  {  // placement check through the agent-scope path: plain-store hand-off only when all 8 share an XCD
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t == 0) sh.part[0] = xcc;
    __syncthreads();
    po_gather(sh, 1, gslot, wig, ++tag, a.err, false);
    bool same = true;
#pragma unroll
    for (int w = 1; w < PO_GW; ++w) same = same && (sh.gath[w][0] == sh.gath[0][0]);
    same_xcd = same && (a.allow_l2_handoff != 0);
    __syncthreads();
  }

  for (int64_t b = grp; b < a.B; b += ngroups) {
    const bool stamp = a.dbg && b == 0 && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    const int row = wig * a.RW + t;
    const bool valid = (t < a.RW) && (row < a.N);
    float* crow = c_s + t * PO_CLD;
    {
      const int row0 = wig * a.RW;
      const int nv = max(0, min(a.RW, a.N - row0));
      float4 cq[RC / 4];
      rows_issue<RC, PO_TPB>(a.C + ((size_t)b * a.N + row0) * RC, nv, cq);  // coalesced: 1 KiB per wave instruction
      rows_commit<RC, PO_CLD, PO_TPB>(c_s, cq);
    }
    __syncthreads();
    float dg = 0.f;
    {
      float acc = crow[0] * crow[0];  // (root ** 2).sum(-1), sequential in r
#pragma unroll
      for (int r = 1; r < RC; ++r) acc = acc + crow[r] * crow[r];
      dg = valid ? acc : 0.f;
    }
    int pos = valid ? row : PO_INVALID;
    float Lr[PO_MAXR];
#pragma unroll
    for (int m = 0; m < PO_MAXR; ++m) Lr[m] = 0.f;

    if (stamp) a.dbg[1] = wall_clock64();
    PoRow st{dg, pos, row, valid};
    po_pivots<RC, 0>(a, sh, c_s, crow, Lr, st, b, gslot, wig, tag, same_xcd);

    if (stamp) a.dbg[2] = wall_clock64();
    if (valid) {
      float* Lb = a.L + (size_t)b * a.max_rank * a.N + row;
#pragma unroll
      for (int m = 0; m < PO_MAXR; ++m)
        if (m < a.max_rank) Lb[(size_t)m * a.N] = (m < a.rank) ? Lr[m] : 0.f;
    }
    __syncthreads();  // c_s / sh reuse by the next member
    if (stamp) a.dbg[3] = wall_clock64();
  }
}

// m* = number of pivots the reference takes: pivot 0 always, pivot m >= 1 while max_b error_{m-1} > tol (:57, :99)
__global__ __launch_bounds__(kThreads) void k_po_rank(PoArgs a, float tol, int* m_out) {
  __shared__ float red[kThreads];
  int mstar = a.rank;
  for (int m = 1; m < a.rank; ++m) {
    float lmax = -INFINITY, lnan = 0.f;
    for (int64_t b = threadIdx.x; b < a.B; b += kThreads) {
      const float e = a.err_rec[(size_t)m * a.B + b] / a.orig[b];
      if (e != e) lnan = 1.f;
      lmax = fmaxf(lmax, e);
    }
    const float mx = block_max256(lmax, red);
    const float anynan = block_sum256(lnan, red);
    if (!((anynan == 0.f) && (mx > tol))) {  // torch.max propagates NaN and (NaN > tol) is False
      mstar = m;
      break;
    }
  }
  if (threadIdx.x == 0) *m_out = mstar;
}

__global__ __launch_bounds__(kThreads) void k_po_perm(PoArgs a, const int* __restrict__ m_in, long long* __restrict__ perm) {
  const int64_t b = blockIdx.x;
  const int mstar = *m_in;
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
  if (g_onchip_disabled || op->kind != LO_OP_LOWRANK_DIAG) return false;
  const int64_t R = op->R;
  return (R == 8 || R == 16 || R == 32) && max_rank <= PO_MAXR && op->N >= 1024 &&
         op->N <= (int64_t)PO_GW * PO_TPB && onchip_num_workgroups() >= 64;
}

struct PoLayout {
  float* err_rec;
  float* orig;
  int* swaps;
  unsigned long long* gbuf;
  int* err;
  int* m_out;
  long long* dbg;
};

static void po_layout(int64_t B, int max_rank, Arena& ar, PoLayout* l) {
  l->err = ar.take<int>(4);
  l->m_out = l->err + 1;
  l->err_rec = ar.take<float>((size_t)max_rank * B);
  l->orig = ar.take<float>(B);
  l->swaps = ar.take<int>((size_t)B * max_rank);
  l->gbuf = ar.take<unsigned long long>((size_t)64 * 2 * PO_GW * PO_SLOT);
  l->dbg = ar.take<long long>(8);
}

size_t pc_onchip_workspace_bytes(int64_t B, int max_rank) {
  Arena ar(nullptr, 0);
  PoLayout l;
  po_layout(B, max_rank, ar, &l);
  return ar.off + 1024;
}

int pc_onchip_run(const lo_op_desc* op, int rank, int max_rank, float tol, float* L_rows, long long* perm,
                  int32_t* rank_out, void* ws, size_t ws_bytes, hipStream_t st) {
  Arena ar(ws, ws_bytes);
  PoLayout l;
  po_layout(op->B, max_rank, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const int nwg = onchip_num_workgroups();
  const int ngroups = nwg / PO_GW;
  PoArgs a;
  a.C = op->A0;
  a.B = op->B;
  a.N = (int)op->N;
  a.RW = (int)((op->N + PO_GW - 1) / PO_GW);
  a.rank = rank;
  a.max_rank = max_rank;
  a.L = L_rows;
  a.err_rec = l.err_rec;
  a.orig = l.orig;
  a.swaps = l.swaps;
  a.gbuf = l.gbuf;
  a.err = l.err;
  a.allow_l2_handoff = getenv("LO_OC_NO_L2_HANDOFF") ? 0 : 1;
  const bool debug = getenv("LO_OC_DEBUG") != nullptr;
  a.dbg = debug ? l.dbg : nullptr;
  if (debug) LO_HIP_CHECK(hipMemsetAsync(l.dbg, 0, 8 * sizeof(long long), st));
  LO_HIP_CHECK(hipMemsetAsync(l.err, 0, 4 * sizeof(int), st));
  LO_HIP_CHECK(hipMemsetAsync(l.gbuf, 0, sizeof(unsigned long long) * (size_t)ngroups * 2 * PO_GW * PO_SLOT, st));
  dim3 grid(nwg), block(PO_TPB);
  LO_PROF_BEGIN("pc_onchip", st);
  if (op->R == 32) hipLaunchKernelGGL((k_pc_onchip<32>), grid, block, 0, st, a);
  else if (op->R == 16) hipLaunchKernelGGL((k_pc_onchip<16>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_pc_onchip<8>), grid, block, 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_po_rank, dim3(1), dim3(kThreads), 0, st, a, tol, l.m_out);
  LO_PROF_BEGIN("pc_onchip_perm", st);
  hipLaunchKernelGGL(k_po_perm, dim3((unsigned)op->B), dim3(kThreads), 0, st, a, l.m_out, perm);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  int h[2];
  LO_HIP_CHECK(hipMemcpyAsync(h, l.err, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  LO_HIP_CHECK(hipStreamSynchronize(st));
  if (debug) {
    long long ts[8];
    LO_HIP_CHECK(hipMemcpy(ts, l.dbg, sizeof(ts), hipMemcpyDeviceToHost));
    fprintf(stderr, "pc_onchip member0 (100 MHz ticks): load %lld pivots %lld store %lld | reduce+publish %lld gather %lld update %lld\n",
            ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4], ts[5], ts[6]);
  }
  if (h[0]) return LO_ERR_LAUNCH;
  *rank_out = h[1];
  return LO_OK;
}

}  // namespace lo

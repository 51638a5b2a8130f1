// lo_lowrank_mv.hip -- ONE-PASS operator-resident matvec of low-rank + diagonal members:  y = C (C^T v) + d o v
//
// Replaces RootLinearOperator._matmul (linear_operator/operators/root_linear_operator.py:68-72: root @ (root^T @ rhs))
// under AddedDiagLinearOperator._matmul (added_diag_linear_operator.py:72-76: addcmul(linear_op._matmul(rhs), diag, rhs)).
//
// The streaming form (lo_skinny.hip: k_skinny_tn, then k_skinny_nn) reads C twice -- the second pass cannot start before
// t = C^T v of the whole member is known, and 512 MiB of C neither fits the L2s nor comes back faster from the Infinity
// Cache than from HBM.  Here the rows of C are read from HBM ONCE and wait in registers for t:
//
//   * a member is a group of GW workgroups (256 threads, 1024 rows each; GW = smallest power of two that holds the member),
//     all groups co-resident (up to three workgroups per CU), members dealt round-robin to the groups;
//   * NO transposition: lane (g = l / CH, k = l % CH) of a wave keeps the 16-byte chunk k of the rows RPI i + g,
//     i = 0 .. NI-1 (CH = RC / 4 chunks per row, RPI = 64 / CH rows per wave instruction) -- every wave load is 1 KiB of
//     consecutive addresses, whole cache lines, and the 128 (RC = 32) registers it fills are used as they arrive;
//   * pass 1: t_part[4 k .. 4 k + 3] += C[row][4 k ..] v[row] (v from a per-wave LDS stage, broadcast reads), reduce-scatter
//     over the lanes that share a chunk (permlane32 / permlane16 swaps, DPP), wave partials through LDS;
//   * ONE group all-reduce of RC c values through tagged 8-byte granules (the hand-off of the resident CG kernels,
//     lo_group_reduce.h: value | tag in one never-torn store, plain same-XCD stores when the placement check passes);
//   * pass 2: row sums  C[row][4 k ..] . t[4 k ..]  from the SAME registers, reduce-scatter over the CH lanes of a row so that
//     lane (g, k) ends with the rows 64 j + RPI k + g, j = 0 .. 3 (a wave store covers 64 consecutive rows), + d o v.
//
// HBM traffic = SURVEY 8(d)'s algorithmic bytes 4 (N R + N + 2 N c) per member; no HBM round trip for t.
// Workgroups wait for each other, so every group must be resident: the grid is sized by the occupancy query, launches are
// ordered by ResidentLaunch, and a hand-off that times out (co-residency lost to a foreign kernel; never seen on a
// dedicated GPU) does NOT fail the call: the workgroup recomputes t of the whole member from HBM by itself (same result
// to summation order) and the sticky error word sends every later member of the launch down the same path.
#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_group_reduce.h"

namespace lo {

namespace {

constexpr int MV_TPB = 256;
constexpr int MV_ROWS = 1024;  // rows per workgroup (256 per wave)
constexpr int MV_ERR_BYTES = 256;

struct LrMvArgs {
  const float* C;   // [B, N, RC]
  const float* d;   // [B, N] | [B] | nullptr
  int d_mode;
  const float* v;   // [B, N, c]
  float* y;         // [B, N, c]
  int c;
  int64_t B;
  int N;
  unsigned long long* gran;  // [ngroups][2][GW][RC * CT + 2]
  unsigned* err;             // == tag_base while a hand-off of THIS launch is lost (device memory: every workgroup reads it)
  unsigned* err_host;        // the same word in pinned host memory the device maps, written ONLY by a workgroup that loses
                             // its hand-off: the host sees a loss at its next call without a copy or a synchronisation
                             // (all 512 workgroups READING a host word at kernel entry cost 70 us: 113 -> 184 us)
  const float* zero;         // a word of device memory that is never written (the "diagonal" of an operator without one)
  unsigned tag_base;         // tags of this launch: tag_base + 1 + (member of the group); larger than any earlier launch's
  int allow_l2_handoff;
  int prio_mode;             // wave priority against the oldest-first arbitration of a CU's two workgroups (LO_MV_PRIO)
  const int* stop;
  long long* dbg;  // optional phase clocks of member 0 / workgroup 0 (LO_MV_DEBUG)
};

// Global pointers with an explicit address space (an opaque copy would otherwise degrade to flat accesses) and a per-use
// opaque copy of the wave-uniform base: addresses are formed right where they are used instead of being computed at kernel
// entry for every phase and kept -- or spilled -- across the member loop (the rows of C leave 40 registers for everything else).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const f32x4 g_cf4;
typedef __attribute__((address_space(1))) const float g_cf;
typedef __attribute__((address_space(1))) float g_f;
typedef __attribute__((address_space(1))) const char g_cc;
template <class T>
__device__ __forceinline__ T* opaque_uniform(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// xor-4 exchange on the DPP path (two row shifts under complementary bank masks) instead of the LDS-crossbar ds_swizzle
__device__ __forceinline__ float xor4_dpp(float v) {
  const int x = __float_as_int(v);
  const int a = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0, 2 read lane + 4
  const int b = __builtin_amdgcn_update_dpp(a, x, 0x114, 0xf, 0xa, false);  // row_shr:4 -> banks 1, 3 read lane - 4
  return __int_as_float(b);
}
__device__ __forceinline__ float halve4(float lo, float hi, int lane) {
  const bool up = (lane & 4) != 0;
  const float keep = up ? hi : lo;
  const float send = up ? lo : hi;
  return keep + xor4_dpp(send);
}

// sum of one row's CH lane partials for CH rows at once: lane k ends with the total of p[k]
template <int CH>
__device__ __forceinline__ float rows_reduce(float (&p)[CH], int lane) {
  if constexpr (CH == 8) {
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] = halve4(p[m], p[m + 4], lane);
    p[0] = halve_pair<2>(p[0], p[2], lane);
    p[1] = halve_pair<2>(p[1], p[3], lane);
    return halve_pair<1>(p[0], p[1], lane);
  } else if constexpr (CH == 4) {
    p[0] = halve_pair<2>(p[0], p[2], lane);
    p[1] = halve_pair<2>(p[1], p[3], lane);
    return halve_pair<1>(p[0], p[1], lane);
  } else {
    return halve_pair<1>(p[0], p[1], lane);
  }
}

// the four components of a lane's chunk, summed over the lanes that hold the same chunk (lane bits log2 CH .. 5):
// returns component 2 * bit5 + bit4 of the chunk (every lane of the class holds the same bits)
template <int CH>
__device__ __forceinline__ float chunk_reduce(const float (&tv)[4], int lane) {
  const float a = halve_pair<32>(tv[0], tv[2], lane);
  const float b = halve_pair<32>(tv[1], tv[3], lane);
  float s = halve_pair<16>(a, b, lane);
  s = bfly_add<8>(s);
  if constexpr (CH <= 4) s += xor4_dpp(s);
  if constexpr (CH <= 2) s = bfly_add<2>(s);
  return s;
}

template <int RC, int GW, int CT, int WPS>
__global__ __launch_bounds__(MV_TPB, WPS) void k_lr_mv(LrMvArgs a) {
  constexpr int CH = RC / 4;     // 16-byte chunks per row
  constexpr int RPI = 64 / CH;   // rows per wave load instruction
  constexpr int NI = 256 / RPI;  // load instructions per wave = 4 CH
  constexpr int NP = RC * CT;    // payload of the all-reduce
  constexpr int NPS = NP + 2;    // + the placement check that rides on a group's first exchange (XCC id, its square)
  static_assert(NPS <= MV_TPB, "one payload entry per thread");
  __shared__ __attribute__((aligned(16))) float vst[4][256 * CT];
  __shared__ __attribute__((aligned(16))) float red[4][NPS];
  __shared__ __attribute__((aligned(16))) float res[NPS];
  __shared__ float pol[GW > 1 ? GW * NPS : 1];
  __shared__ int lost_s;
  // (the sticky stop word of a CG solve and the launch's error word are REQUESTED here and looked at behind the first
  //  member's loads: two dependent round trips less in front of the first byte from HBM)
  const int stop0 = a.stop ? *a.stop : 0;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;  // (workgroups go round-robin to the XCDs: checked below, not assumed)
  const int per_xcd = (int)gridDim.x / 8;
  const int groups_per_xcd = per_xcd / GW;
  const int gix = jx / GW, wig = jx % GW;
  if (gix >= groups_per_xcd) return;
  const int grp = xcd * groups_per_xcd + gix;
  const int ngroups = groups_per_xcd * 8;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k = lane & (CH - 1), g = lane / CH;
  unsigned long long* const gbase = a.gran + (size_t)grp * 2 * GW * NPS;
  // The granules live in a buffer the LIBRARY owns and are never cleared between launches: a launch's tags are larger than
  // every tag an earlier launch left behind (lowrank_mv_run hands out tag ranges), so a stale granule never matches.
  unsigned tag = a.tag_base;
  bool same_xcd = false;
  unsigned err0 = 0;
  if (t == 0) err0 = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // Sum over the group of red[0..3][0 .. cnt) -> res[0 .. cnt); false when the hand-off was lost.  The first cnt threads
  // publish the workgroup's partials; ALL threads poll (granule idx = workgroup w * cnt + entry e, one or a few per thread:
  // two registers instead of 2 GW for the poll), park the values in LDS, and the first cnt threads add them in the fixed
  // order w = 0 .. GW-1 -- the same bits in every workgroup of the group.
  auto group_sum = [&](const int cnt) -> bool {
    ++tag;
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));  // (LDS / granule addresses are formed here, not at kernel entry)
    __syncthreads();
    if (lost_s != 0) return false;  // (another launch-wide loss, or the test switch: decided behind a barrier, uniformly)
    if constexpr (GW == 1) {
      if (t < cnt) res[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    } else {
      unsigned long long* slot = gbase + (size_t)(tag & 1u) * GW * NPS;
      if (t < cnt) {
        const float s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(s);
        if (same_xcd) __hip_atomic_store(slot + (size_t)wig * NPS + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(slot + (size_t)wig * NPS + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned spin = 0;
      bool lost = false;
#pragma unroll 1
      for (int idx = t; idx < GW * cnt && !lost; idx += MV_TPB) {
        const int w = idx / cnt, e = idx - w * cnt;
        unsigned long long x;
        for (;;) {
          x = __hip_atomic_load(slot + (size_t)w * NPS + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(x >> 32) == tag) break;
          if (++spin > R4_MAXSPIN ||
              ((spin & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.tag_base)) {
            if (atomicExch(a.err, a.tag_base) != a.tag_base)
              __hip_atomic_store(a.err_host, a.tag_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        pol[idx] = __uint_as_float((unsigned)(x & 0xffffffffull));
      }
      if (lost) lost_s = 1;
      __syncthreads();
      if (t < cnt) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < GW; ++w) tot += pol[w * cnt + t];
        res[t] = tot;
      }
    }
    __syncthreads();
    return lost_s == 0;
  };
  // placement check (see k_cg_onchip4): the group's FIRST exchange goes through agent-scope stores and carries the XCC id;
  // plain same-XCD stores (hand-off at L2 latency) only from the second member on, and only if the whole group shares an XCD
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
  bool first = true, lost = false;
  if (a.dbg && wig == 0 && t == 0) a.dbg[8 + 2 * grp] = wall_clock64();  // (LO_MV_DEBUG: when each group starts / ends)

  // A wave owns the rows [row0, row0 + 256) of the member; it LOADS the 256 rows from row0c = min(row0, N - 256) on, so that
  // no address has to be clamped per row (one uniform base per load instruction + one 32-bit lane offset: the per-row
  // clamp cost 64 address registers) -- rows below row0 belong to the previous wave and enter with v = 0, as do rows >= N.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int row0 = wig * MV_ROWS + 256 * wave_u;
  const int row0c = max(0, min(row0, a.N - 256));
  const unsigned loff = (unsigned)((g * RC + 4 * k) * sizeof(float));
  float* const vw = vst[wave_u];
  const int c = a.c;
  // Of a CU's two workgroups the one dispatched first is served first (oldest-wave arbitration) and ran 18 % faster:
  // first-slot groups finished at 88 - 97 us, second-slot groups at 107 - 112 (LO_MV_DEBUG; with the priority of the
  // second slot raised for good it is the other way round).  The members of a launch are dealt out statically, so the
  // priority ALTERNATES from member to member: all groups finish at 92 - 104 us.  (Groups mixed from both slots -- of the
  // same CUs, or of different ones -- were measured at 121 - 123 us against 115: a group runs at the pace of its slowest
  // workgroup.)  Performance only: jx >= per_xcd / 2 is "second slot" when the dispatcher fills the CUs in order.
  const int slot2 = jx >= per_xcd / 2 ? 1 : 0;
  int mseq = 0;
  for (int64_t b = grp; b < a.B; b += ngroups, ++mseq) {
    if (a.prio_mode == 2) {
      if ((slot2 ^ (mseq & 1)) != 0) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else if (a.prio_mode == 1) {
      if (slot2) __builtin_amdgcn_s_setprio(1);
    }
    const bool stamp = a.dbg && grp == 0 && wig == 0 && t == 0;
    long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (stamp) s0 = wall_clock64();
    // ---- every request of the member in flight before anything waits ----
    f32x4 Cr[NI];
    {
      g_cc* Cw = (g_cc*)(a.C + ((size_t)b * a.N + row0c) * RC);
#pragma unroll
      for (int q = 0; q < NI / 8; ++q) {  // (a wave instruction covers 1 KiB: eight of them around one base, offsets -4096 .. 3072)
        g_cc* bq = opaque_uniform(Cw + (8 * q + 4) * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) Cr[8 * q + i] = *(g_cf4*)(bq + loff + (i - 4) * 1024);
      }
    }
    float vq[4 * CT];
    {
      g_cf* vb = opaque_uniform((g_cf*)(a.v + ((size_t)b * a.N + row0c) * c));
      const int lo_e = (row0 - row0c) * c;                      // floats of rows the previous wave owns
      const int hi_e = max(0, min(256, a.N - row0c)) * c;       // floats of rows below N
#pragma unroll
      for (int m = 0; m < 4 * CT; ++m) {
        const int e = lane + 64 * m;
        vq[m] = (e >= lo_e && e < hi_e) ? vb[e] : 0.f;
      }
    }
    float dq[4];
    {
      // (branch-free: stride 1 for a full diagonal, 0 for a constant one, and a zeroed word of the workspace for none)
      const int dstride = a.d_mode == LO_DIAG_FULL ? 1 : 0;
      g_cf* db = opaque_uniform((g_cf*)(a.d_mode == LO_DIAG_FULL ? a.d + (size_t)b * a.N + row0c
                                        : a.d_mode == LO_DIAG_CONST ? a.d + b : a.zero));
#pragma unroll
      for (int j = 0; j < 4; ++j) dq[j] = db[(64 * j + RPI * k + g) * dstride];  // (row0c + 255 < N)
    }
    // v -> the wave's own LDS stage, row stride CT (columns beyond c stay zero)
    if (CT == 1 || c == CT) {
#pragma unroll
      for (int m = 0; m < 4 * CT; ++m) vw[lane + 64 * m] = vq[m];
    } else {
#pragma unroll
      for (int m = 0; m < 4 * CT; ++m) vw[lane + 64 * m] = 0.f;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < 4 * CT; ++m) {
        const int e = lane + 64 * m;
        if (e < 256 * c) vw[(e / c) * CT + (e % c)] = vq[m];
      }
    }
    if (mseq == 0) {  // (behind the loads: the words requested at kernel entry)
      if (stop0) return;
      if (t == 0) lost_s = (err0 == a.tag_base) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    if (stamp) s1 = wall_clock64();

    if (!lost) {
      // ---- pass 1: partials of t = C^T v for this lane's chunk ----
      int l1 = lane;
      asm volatile("" : "+v"(l1));
      const int comp = ((l1 >> 5) << 1) | ((l1 >> 4) & 1);
      const bool writer = (l1 & 15 & ~(CH - 1)) == 0;
      // (two columns at a time, eight rows at a time: the scheduler otherwise hoists every stage read of the pass over the
      //  loop -- 53 spilled registers at four columns beside the 128 of C)
      constexpr int CB = CT < 2 ? CT : 2;
#pragma unroll
      for (int c0 = 0; c0 < CT; c0 += CB) {
        float tacc[CB][4];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) tacc[cc][0] = tacc[cc][1] = tacc[cc][2] = tacc[cc][3] = 0.f;
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += 8) {
#pragma unroll
          for (int i = i0; i < i0 + 8; ++i) {
            const int r = RPI * i + g;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
              const float vv = vw[r * CT + c0 + cc];
              tacc[cc][0] = fmaf(Cr[i].x, vv, tacc[cc][0]);
              tacc[cc][1] = fmaf(Cr[i].y, vv, tacc[cc][1]);
              tacc[cc][2] = fmaf(Cr[i].z, vv, tacc[cc][2]);
              tacc[cc][3] = fmaf(Cr[i].w, vv, tacc[cc][3]);
            }
          }
          if constexpr (CT > 2) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
          const float s = chunk_reduce<CH>(tacc[cc], l1);
          if (writer) red[wave_u][(c0 + cc) * RC + 4 * (l1 & (CH - 1)) + comp] = s;
        }
        if constexpr (CT > 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    bool have_t = !lost;
    if (stamp) s2 = wall_clock64();
    if (have_t) {
      if (GW > 1 && first) {
        if (t < 8) red[t >> 1][NP + (t & 1)] = t == 0 ? (float)xcc : (t == 1 ? (float)(xcc * xcc) : 0.f);
        have_t = group_sum(NPS);
        const float fx = (float)xcc;
        same_xcd = have_t && (res[NP] == GW * fx) && (res[NP + 1] == GW * fx * fx) && (a.allow_l2_handoff != 0);
        first = false;
      } else {
        have_t = group_sum(NP);
      }
    }
    if (stamp) s3 = wall_clock64();
    lost = !have_t;  // (uniform: read behind a barrier; once lost, always lost for this launch)
    if (!have_t) {
      // ---- the hand-off is lost: t of the WHOLE member from HBM, by this workgroup alone (every workgroup of the group
      // computes the same bits) ----
      float ts[CT][4];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) ts[cc][0] = ts[cc][1] = ts[cc][2] = ts[cc][3] = 0.f;
      int ls = (int)threadIdx.x;
      asm volatile("" : "+v"(ls));
      const int lane = ls & 63, k = lane & (CH - 1), g = lane / CH, t = ls;
      g_cf* Cm = opaque_uniform((g_cf*)(a.C + (size_t)b * a.N * RC));
      g_cf* vm = opaque_uniform((g_cf*)(a.v + (size_t)b * a.N * c));
#pragma unroll 1
      for (int rb = 0; rb < GW; ++rb) {
#pragma unroll 1
        for (int i = 0; i < NI; ++i) {
          const int row = rb * MV_ROWS + 256 * wave_u + RPI * i + g;
          if (row < a.N) {
            const f32x4 c4 = *(g_cf4*)(Cm + (size_t)row * RC + 4 * k);
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) {
              const float vv = (cc < c) ? vm[(size_t)row * c + cc] : 0.f;
              ts[cc][0] = fmaf(c4.x, vv, ts[cc][0]);
              ts[cc][1] = fmaf(c4.y, vv, ts[cc][1]);
              ts[cc][2] = fmaf(c4.z, vv, ts[cc][2]);
              ts[cc][3] = fmaf(c4.w, vv, ts[cc][3]);
            }
          }
        }
      }
      const int comp = ((lane >> 5) << 1) | ((lane >> 4) & 1);
      const bool writer = (lane & 15 & ~(CH - 1)) == 0;
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        const float s = chunk_reduce<CH>(ts[cc], lane);
        if (writer) red[wave_u][cc * RC + 4 * k + comp] = s;
      }
      __syncthreads();
      if (t < NP) res[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      __syncthreads();
    }

    // ---- pass 2: y = C t + d o v from the same registers ----
    g_f* const yb = opaque_uniform((g_f*)(a.y + ((size_t)b * a.N + row0c) * c));
    int l2 = lane;
    asm volatile("" : "+v"(l2));  // (the store offsets are formed here, not at kernel entry)
    const int k2 = l2 & (CH - 1), g2 = l2 / CH;
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      const float4 t4 = *reinterpret_cast<const float4*>(&res[cc * RC + 4 * k2]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p[CH];
#pragma unroll
        for (int m = 0; m < CH; ++m) {
          const f32x4 c4 = Cr[CH * j + m];
          p[m] = fmaf(c4.w, t4.w, fmaf(c4.z, t4.z, fmaf(c4.y, t4.y, c4.x * t4.x)));
        }
        const float yv = rows_reduce<CH>(p, lane);
        const int rw = 64 * j + RPI * k2 + g2;  // row inside the wave's block
        const int row = row0c + rw;
        const float out = fmaf(dq[j], vw[rw * CT + cc], yv);
        if (row >= row0 && row < a.N && cc < c) yb[rw * c + cc] = out;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next member's stage writes follow this member's reads (in-order LDS per wave)
    if (stamp) {  // accumulated over the members of group 0 (100 MHz ticks): load | pass 1 | exchange | pass 2 + store
      a.dbg[0] += s1 - s0;
      a.dbg[1] += s2 - s1;
      a.dbg[2] += s3 - s2;
      a.dbg[3] += wall_clock64() - s3;
      a.dbg[4] += 1;
    }
  }
  if (a.dbg && wig == 0 && t == 0) a.dbg[9 + 2 * grp] = wall_clock64();
}

template <int RC, int GW, int CT>
int lr_mv_go(const LrMvArgs& a, int ncu, hipStream_t st) {
  constexpr int WPS = 2;  // (three workgroups per CU fit at one column -- 164 VGPRs -- and measure the same: 116 vs 114 us)
  int per_cu = 0;
  if (LO_OCCUPANCY_CACHED(per_cu, (k_lr_mv<RC, GW, CT, WPS>), MV_TPB, 0) != hipSuccess || per_cu < 1)
    return LO_ERR_UNSUPPORTED;
  per_cu = std::min(per_cu, WPS);
  if (const char* e = getenv("LO_MV_WGS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));
  const int grid = per_cu * ncu;
  if ((grid / 8) / GW < 1) return LO_ERR_UNSUPPORTED;
  LrMvArgs a2 = a;
  a2.prio_mode = (per_cu == 2) ? 2 : 0;
  if (const char* e = getenv("LO_MV_PRIO")) a2.prio_mode = (per_cu == 2) ? atoi(e) : 0;
  LO_PROF_BEGIN("lr_mv", st);
  hipLaunchKernelGGL((k_lr_mv<RC, GW, CT, WPS>), dim3(grid), dim3(MV_TPB), 0, st, a2);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int RC, int CT>
int lr_mv_gw(int GW, const LrMvArgs& a, int ncu, hipStream_t st) {
  switch (GW) {
    case 1: return lr_mv_go<RC, 1, CT>(a, ncu, st);
    case 2: return lr_mv_go<RC, 2, CT>(a, ncu, st);
    case 4: return lr_mv_go<RC, 4, CT>(a, ncu, st);
    case 8: return lr_mv_go<RC, 8, CT>(a, ncu, st);
    case 16: return lr_mv_go<RC, 16, CT>(a, ncu, st);
    case 32: return lr_mv_go<RC, 32, CT>(a, ncu, st);
  }
  return LO_ERR_UNSUPPORTED;
}

template <int RC>
int lr_mv_ct(int CT, int GW, const LrMvArgs& a, int ncu, hipStream_t st) {
  if (CT == 1) return lr_mv_gw<RC, 1>(GW, a, ncu, st);
  if (CT == 2) return lr_mv_gw<RC, 2>(GW, a, ncu, st);
  return lr_mv_gw<RC, 4>(GW, a, ncu, st);
}

int group_size(int64_t N) {
  int gw = 1;
  while ((int64_t)gw * MV_ROWS < N) gw *= 2;
  return gw;
}

// The hand-off granules of this kernel live in ONE buffer per device that the library allocates on first use and never
// clears again (the only device memory liblo_amd owns besides the profiling aids): clearing a caller's workspace cost a
// 5.5 us launch in front of every 110 us product.  Tags grow monotonically over the launches of a process (resident
// launches of a process never overlap: ResidentLaunch), the buffer is re-zeroed when the 31-bit tag space runs out.
struct MvCtl {
  char* buf = nullptr;
  size_t bytes = 0;
  unsigned next_tag = 1;
  bool failed = false;
  unsigned* err_host = nullptr;  // the launch error word: pinned, mapped
  unsigned* err_dev = nullptr;   // ... as the device addresses it
  unsigned last_tag = 0;         // tag range of the previous launch (0: none / already looked at)
};
MvCtl g_mv_ctl[16];
constexpr size_t MV_MAX_WGS = 3 * 320;  // up to three workgroups per CU, up to 320 CUs

MvCtl* mv_ctl() {  // (called with the ResidentLaunch lock held)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  MvCtl& m = g_mv_ctl[dev];
  if (!m.buf && !m.failed) {
    m.bytes = MV_ERR_BYTES + MV_MAX_WGS * 2 * (size_t)(32 * 4 + 2) * sizeof(unsigned long long);
    void* eh = nullptr;
    void* ed = nullptr;
    if (hipMalloc(&m.buf, m.bytes) != hipSuccess || hipMemset(m.buf, 0, m.bytes) != hipSuccess ||
        hipHostMalloc(&eh, 256, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&ed, eh, 0) != hipSuccess) {
      (void)hipGetLastError();
      m.buf = nullptr;
      m.failed = true;
    } else {
      memset(eh, 0, 256);
      m.err_host = static_cast<unsigned*>(eh);
      m.err_dev = static_cast<unsigned*>(ed);
    }
  }
  return m.buf ? &m : nullptr;
}

}  // namespace

bool lowrank_mv_eligible(int R4, int64_t N, int64_t c) {
  if (getenv("LO_NO_RESIDENT_MV")) return false;
  const int ncu = onchip_num_workgroups();
  return (R4 == 8 || R4 == 16 || R4 == 32) && c >= 1 && c <= 4 && N >= 256 &&
         N <= (int64_t)32 * MV_ROWS && ncu >= 64 && (size_t)3 * ncu <= MV_MAX_WGS;
}

int lowrank_mv_run(const float* C, int R4, const float* d, int d_mode, const float* v, float* y, int64_t B, int64_t N,
                   int64_t c, const int* stop, hipStream_t st) {
  if (!lowrank_mv_eligible(R4, N, c) || resident_off() || tls_graph_capture) return LO_ERR_UNSUPPORTED;
  const int ncu = onchip_num_workgroups();
  const int ct = c == 1 ? 1 : (c == 2 ? 2 : 4);
  ResidentLaunch guard(st);
  MvCtl* m = mv_ctl();
  if (!m) return LO_ERR_UNSUPPORTED;
  // a launch of this process that lost its co-residency (another process' resident kernel held part of the CUs) repaired
  // itself inside the kernel, at the price of a 0.5 s spin: it left its tag in the pinned error word, and the gate of
  // the resident kernels starts its cool-down here -- the next calls run the two streaming passes (lo_resident_status)
  if (m->last_tag && *static_cast<volatile unsigned*>(m->err_host) == m->last_tag) {
    m->last_tag = 0;
    onchip_note_timeout();
    return LO_ERR_UNSUPPORTED;
  }
  // tags of this launch: tag_base + 1 .. tag_base + (members per group) <= tag_base + B / 8 + 1
  const unsigned need = (unsigned)std::min<int64_t>(B / 8 + 2, 1 << 28);
  if (m->next_tag + need >= 0x7ff00000u) {
    LO_HIP_CHECK(hipMemsetAsync(m->buf, 0, m->bytes, st));  // (ordered behind every earlier resident launch by the guard)
    m->next_tag = 1;
  }
  LrMvArgs a;
  a.C = C; a.d = d; a.d_mode = d ? d_mode : LO_DIAG_NONE; a.v = v; a.y = y; a.c = (int)c; a.B = B; a.N = (int)N;
  a.err = reinterpret_cast<unsigned*>(m->buf + 128);
  a.err_host = m->err_dev;
  a.zero = reinterpret_cast<const float*>(m->buf);  // (the first 128 bytes of the buffer stay zero for good)
  a.gran = reinterpret_cast<unsigned long long*>(m->buf + MV_ERR_BYTES);
  a.tag_base = m->next_tag;
  m->next_tag += need;
  m->last_tag = a.tag_base;
  a.allow_l2_handoff = onchip_l2_handoff_allowed();
  a.stop = stop;
  a.dbg = nullptr;
  if (getenv("LO_MV_TEST_FALLBACK")) {  // every workgroup starts "lost" (tests of the in-kernel repair; not a real loss)
    LO_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)a.err, (int)a.tag_base, 1, st));
    m->last_tag = 0;
  } else if (resident_take_injection()) {  // lo_resident_inject_timeouts: the same THROUGH the gate's bookkeeping
    LO_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)a.err, (int)a.tag_base, 1, st));
    *static_cast<volatile unsigned*>(m->err_host) = a.tag_base;  // (what a losing workgroup would have stored)
  }
  static long long* dbg_buf = nullptr;
  const bool dbg = getenv("LO_MV_DEBUG") != nullptr;
  if (dbg) {
    if (!dbg_buf && hipMalloc(&dbg_buf, (8 + 2 * MV_MAX_WGS) * sizeof(long long)) != hipSuccess) dbg_buf = nullptr;
    if (dbg_buf) LO_HIP_CHECK(hipMemsetAsync(dbg_buf, 0, (8 + 2 * MV_MAX_WGS) * sizeof(long long), st));
    a.dbg = dbg_buf;
  }
  const int GW = group_size(N);
  int rc;
  if (R4 == 32) rc = lr_mv_ct<32>(ct, GW, a, ncu, st);
  else if (R4 == 16) rc = lr_mv_ct<16>(ct, GW, a, ncu, st);
  else rc = lr_mv_ct<8>(ct, GW, a, ncu, st);
  if (rc == LO_OK && dbg && dbg_buf) {
    static long long h[8 + 2 * MV_MAX_WGS];
    if (hipMemcpyAsync(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess &&
        hipStreamSynchronize(st) == hipSuccess && h[4] > 0) {
      fprintf(stderr, "lr_mv group 0, %lld members, us per member: load %.2f  pass 1 %.2f  exchange %.2f  pass 2 + store %.2f\n",
              h[4], 0.01 * h[0] / h[4], 0.01 * h[1] / h[4], 0.01 * h[2] / h[4], 0.01 * h[3] / h[4]);
      long long t0 = h[8];
      int ng = 0;
      for (size_t gi = 0; gi < MV_MAX_WGS && h[8 + 2 * gi]; ++gi, ++ng) t0 = std::min(t0, h[8 + 2 * gi]);
      fprintf(stderr, "lr_mv groups (start, end in us after the first start):");
      for (int gi = 0; gi < ng; ++gi)
        fprintf(stderr, "%s%.0f-%.0f", gi % 12 ? " " : "\n  ", 0.01 * (h[8 + 2 * gi] - t0), 0.01 * (h[9 + 2 * gi] - t0));
      fprintf(stderr, "\n");
    }
  }
  return rc;
}

}  // namespace lo

// lo_rspace3.hip -- k_cg_rspace<.., true> (lo_rspace.hip: the preconditioned CG of a low-rank + diagonal member as the CG of a
// DIAGONAL matrix on its R + 1 Krylov coordinates, linear_operator/utils/linear_cg.py:245-332) in the register layout of the
// one-pass matvec (lo_lowrank_mv.hip).  Same algebra, same iteration chain (copied operation for operation), same outputs;
// what changed is everything around the chain, which is where a member spent 14 of its 20 us:
//
//   * NO transposition window: lane (g = l / CH, k = l % CH) keeps the 16-byte chunk k of the rows RPI i + g of its wave's
//     256 rows -- every wave load is 1 KiB of consecutive addresses and the registers are used as they arrive.  The old
//     layout (a row per lane) moved every row set through a 32 KB LDS window: 4 x (8 ds_write_b128 + 8 ds_read_b128) per lane
//     and two wave barriers per row set on the critical path of the member load (6.4 - 10 us; here 3 - 4.5 us);
//   * the reduction over the rows (w0 = C^T D^-1 b, u0 = C^T b, fp64) accumulates the lane's chunk over its 32 rows (8
//     accumulators instead of 2 x 8 per 8-column block), the right-hand side and 1/d come from a per-wave LDS stage as
//     broadcast reads; ONE reduce-scatter of 8 values over the lanes that share a chunk leaves the 64 totals of the wave
//     in its 64 lanes (the old layout: four 16-value reduce-scatters);
//   * three of the form's matrices in LDS (TinT | TuT | G2: 26 KB instead of four in 35 KB), no 32 KB window: 45 KB per
//     workgroup; the group all-reduce polls with every thread (granule pairs spread over the 256 threads, all loads of a
//     thread in flight together) instead of 2 GW loads in each of 68 threads; the placement check rides on the group's
//     first exchange;
//   * x = D^-1 (xi b + C (nrm y)) in fp64 from the same registers: row sums reduce-scattered over the CH lanes of a row, a
//     wave store covers 64 consecutive rows;
//   * (second pass) a member starts with all of its rows already requested: a quarter during the previous member's
//     iterations, the rest inside its x pass (168 -> 152 us per launch on one box, 142.5 us in the committed profile).
//
// Numerics: fp64 sums in another order (the old kernel's results to ~1e-15 relative before the final rounding to fp32).
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"
#include "lo_group_reduce.h"
#include "lo_cg_close.h"
#include "lo_f64_lanes.h"

namespace lo {

namespace {

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

constexpr int R3_TPB = 256;
constexpr int R3_ROWS = 1024;

// Gather by peer writes (prototype, SURVEY 8(e): "one RCCL all-gather over xGMI at the end"): the x pass can store every
// solution value into up to seven more buffers next to the caller's -- on a node these would be the IPC-mapped gather
// buffers of the seven peers (each rank writes its slice straight over its own xGMI link while the solve runs; no RCCL
// kernel competes for CUs with the spin-waiting groups).  lo_peer_gather_set installs the pointers for the calling
// process; tools/mb_peer_gather.py emulates the peers with local buffers and times the solve with and without them.
struct PeerOut {
  float* buf[7];
  int n;
  long long member_off;  // this rank's first member inside a peer's gather buffer
};
PeerOut g_peer_out = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, 0, 0};

__host__ __device__ constexpr int r3_np(int RC) { return 2 * RC + 6; }  // w0 | u0 | s | a0 | next member | sum dinv^2 | xcc | xcc^2

// xor-4 exchange of a double on the DPP path (two row shifts under complementary bank masks per word) instead of the
// LDS-crossbar ds_swizzle with its own wait
__device__ __forceinline__ unsigned xor4_dpp_u(unsigned x) {
  const int a = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0, 2 read lane + 4
  return (unsigned)__builtin_amdgcn_update_dpp(a, (int)x, 0x114, 0xf, 0xa, false);  // row_shr:4 -> banks 1, 3 read lane - 4
}
__device__ __forceinline__ double halve4_d(double lo, double hi, int lane) {
  const bool up = (lane & 4) != 0;
  const double keep = up ? hi : lo;
  const double send = up ? lo : hi;
  return keep + mk_d(xor4_dpp_u(lo_w(send)), xor4_dpp_u(hi_w(send)));
}

// sum of one row's CH lane partials for CH rows at once (fp64): lane k ends with the total of p[k]
template <int CH>
__device__ __forceinline__ double rows_reduce_d(double (&p)[CH], int lane) {
  if constexpr (CH == 8) {
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] = halve4_d(p[m], p[m + 4], lane);
    p[0] = halve_pair_d<2>(p[0], p[2], lane);
    p[1] = halve_pair_d<2>(p[1], p[3], lane);
    return halve_pair_d<1>(p[0], p[1], lane);
  } else if constexpr (CH == 4) {
    p[0] = halve_pair_d<2>(p[0], p[2], lane);
    p[1] = halve_pair_d<2>(p[1], p[3], lane);
    return halve_pair_d<1>(p[0], p[1], lane);
  } else {
    return halve_pair_d<1>(p[0], p[1], lane);
  }
}

template <int RC, int GW>
__global__ __launch_bounds__(R3_TPB, 2) void k_cg_rspace3(OnchipArgs a, PeerOut po) {
  constexpr int CH = RC / 4;     // 16-byte chunks per row
  constexpr int RPI = 64 / CH;   // rows per wave load instruction
  constexpr int NI = 256 / RPI;  // load instructions per wave
  constexpr int NP = r3_np(RC);
  constexpr int MLD = RC + 2;    // row stride of the fp64 matrices in LDS (16-byte aligned, conflict-free b128 reads)
  constexpr int NH = RC / 2;     // columns per half-wave in the R x R products
  constexpr int NG = 2 * GW * NP;  // granules of one exchange (two per double)
  __shared__ __attribute__((aligned(16))) float bst[4][256];   // right-hand side of the wave's rows
  __shared__ __attribute__((aligned(16))) float dst[4][256];   // 1 / d of the wave's rows
  __shared__ __attribute__((aligned(16))) double mat_s[3 * RC * MLD];  // TinT | TuT | G2
  __shared__ __attribute__((aligned(16))) double red[4][NP];
  __shared__ __attribute__((aligned(16))) double res[NP];
  __shared__ __attribute__((aligned(16))) double gv[4][32];
  __shared__ unsigned polw[GW > 1 ? NG : 2];
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = ((int)gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned long long* const gbase = a.gbuf + (size_t)grp * 2 * NG;  // [parity][GW][NP][2]
  unsigned tag = 0;
  bool same_xcd = false;
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
  bool first = true;
  // rows: a wave owns [row0, row0 + 256); it loads from row0c = min(row0, N - 256) on (no per-row clamp: lo_lowrank_mv.hip),
  // rows below row0 belong to the previous wave and enter with b = 0 and 1/d = 0, as do rows >= N
  const int row0 = wig * R3_ROWS + 256 * wave_u;
  const int row0c = max(0, min(row0, a.N - 256));
  const bool di_full = a.dinv_mode == LO_DIAG_FULL;
  float* const bw = bst[wave_u];
  float* const dw = dst[wave_u];
  // The first NPF load instructions of a member's rows (a QUARTER of them at 32 columns: half spills) are
  // requested while the PREVIOUS member's iterations run: the registers the chain leaves free hold them (C of the current
  // member stays for its x pass) and the requests travel while this workgroup would otherwise ask HBM for nothing.  The
  // other instructions (nxt[]) are requested inside the previous member's x pass, group by group as it frees their
  // registers.  Both arrays are carried around the member loop; the first member's rows are requested here.
  constexpr int NPF = 8;  // (a quarter of the rows at 32 columns, half at 16, all at 8)
  f32x4 pre[NPF];
  constexpr int NXT = NI - NPF;          // the other load instructions: requested while the x pass frees their registers
  f32x4 nxt[NXT > 0 ? NXT : 1];
  auto request = [&](int64_t bm, int tl_, int i_lo, auto cnt, f32x4* dst) {  // load instructions i_lo .. i_lo + cnt - 1 of member bm
    const int ln_ = tl_ & 63, k_ = ln_ & (CH - 1), g_ = ln_ / CH;
    const unsigned loff = (unsigned)((g_ * RC + 4 * k_) * sizeof(float));
    g_cc* Cw = (g_cc*)(a.C + ((size_t)bm * a.N + row0c) * RC);
#pragma unroll
    for (int q = 0; q < decltype(cnt)::value / 8; ++q) {
      g_cc* bq = opaque_uniform(Cw + (i_lo + 8 * q + 4) * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[8 * q + i] = *(g_cf4*)(bq + loff + (i - 4) * 1024);
    }
  };
  int64_t b = grp;
  {
    int tl0 = t;
    asm volatile("" : "+v"(tl0));
    if (b < a.B) {
      request(b, tl0, 0, std::integral_constant<int, NPF>{}, pre);
      request(b, tl0, NPF, std::integral_constant<int, NXT>{}, nxt);
    } else {
#pragma unroll
      for (int i = 0; i < NPF; ++i) pre[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  while (b < a.B) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    int drawn = 0;
    if (wig == 0 && t == 0) drawn = atomicAdd(a.next_member, 1);  // (its round trip hides behind the member load)
    int tl = t;
    asm volatile("" : "+v"(tl));
    const int ln = tl & 63, k = ln & (CH - 1), g = ln / CH;
    // ---- every request of the member in flight before anything waits ----
    f32x4 Cr[NI];
#pragma unroll
    for (int i = 0; i < NPF; ++i) Cr[i] = pre[i];   // (requested during the previous member's iterations)
#pragma unroll
    for (int i = 0; i < NXT; ++i) Cr[NPF + i] = nxt[i];
    float bq4[4], dq4[4];
    {
      g_cf* bb_ = opaque_uniform((g_cf*)(a.rhs + (size_t)b * a.N + row0c));
      g_cf* dd_ = opaque_uniform((g_cf*)(di_full ? a.dinv + (size_t)b * a.N + row0c : a.dinv + b));
      const int lo_e = row0 - row0c, hi_e = max(0, min(256, a.N - row0c));
      const int dstride = di_full ? 1 : 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int e = ln + 64 * m;
        const bool own = e >= lo_e && e < hi_e;
        bq4[m] = own ? bb_[e] : 0.f;
        dq4[m] = own ? dd_[e * dstride] : 0.f;
      }
    }
    // TinT | TuT | G2 of the diagonal form (matrices 0, 2, 3 of lo_precond_desc.RSD), 16-byte pieces
    constexpr int NM = (3 * RC * RC / 2 + R3_TPB - 1) / R3_TPB;
    f32x4 mv[NM];
    {
      typedef __attribute__((address_space(1))) const f32x4 g_m4;
      const g_m4* src = (const g_m4*)(a.RSD + (size_t)b * 6 * RC * RC);
#pragma unroll
      for (int u = 0; u < NM; ++u) {
        const int e = min(tl + R3_TPB * u, 3 * RC * RC / 2 - 1);  // piece e of the three matrices in LDS order
        const int ms = e / (RC * RC / 2);                        // 0, 1, 2 -> matrix 0, 2, 3
        const int mg = ms == 0 ? 0 : ms + 1;
        mv[u] = src[(size_t)mg * (RC * RC / 2) + (e - ms * (RC * RC / 2))];
      }
    }
    const double lam_mine = a.RSD[((size_t)b * 6 + 5) * RC * RC + min(ln & 31, RC - 1)];  // this lane's eigenvalue
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      bw[ln + 64 * m] = bq4[m];
      dw[ln + 64 * m] = dq4[m];
    }
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      const int e = tl + R3_TPB * u;
      if (e < 3 * RC * RC / 2) {
        const int mi = e / (RC / 2), jj = e % (RC / 2);  // (matrix slot * RC + row, column pair)
        *reinterpret_cast<f32x4*>(&mat_s[mi * MLD + 2 * jj]) = mv[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (stamp) a.dbg[1] = wall_clock64();

    // ---- the member's one reduction over the rows, fp64: w0 = C^T (dinv o b), u0 = C^T b (this lane's chunk), s, a0 ----
    {
      double aw[4] = {0.0, 0.0, 0.0, 0.0}, au[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += 4) {
#pragma unroll
        for (int i = i0; i < i0 + 4; ++i) {
          const int r = RPI * i + g;
          const double bb = (double)bw[r];
          const double bd = bb * (double)dw[r];
          const double c0 = (double)Cr[i].x, c1 = (double)Cr[i].y, c2 = (double)Cr[i].z, c3 = (double)Cr[i].w;
          aw[0] = fma(c0, bd, aw[0]); au[0] = fma(c0, bb, au[0]);
          aw[1] = fma(c1, bd, aw[1]); au[1] = fma(c1, bb, au[1]);
          aw[2] = fma(c2, bd, aw[2]); au[2] = fma(c2, bb, au[2]);
          aw[3] = fma(c3, bd, aw[3]); au[3] = fma(c3, bb, au[3]);
        }
        // (four rows at a time: the scheduler otherwise hoists all 64 stage reads and their conversions over the loop and
        //  spills 36 registers of C around the reduction)
        __builtin_amdgcn_sched_barrier(0);
      }
      // scalars over the lane's own four rows 64 j + RPI k + g (every row of the wave exactly once)
      double s_acc = 0.0, a_acc = 0.0, d2_acc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rw = 64 * j + RPI * k + g;
        const double bb = (double)bw[rw], di = (double)dw[rw];
        s_acc = fma(bb, bb * di, s_acc);
        a_acc = fma(bb, bb, a_acc);
        d2_acc = fma(di, di, d2_acc);
      }
      // reduce-scatter over the lanes that share a chunk: bit 5 -> w0 | u0, bits 4, 3 -> component of the chunk
      double h0 = halve_pair_d<32>(aw[0], au[0], ln), h1 = halve_pair_d<32>(aw[1], au[1], ln);
      double h2 = halve_pair_d<32>(aw[2], au[2], ln), h3 = halve_pair_d<32>(aw[3], au[3], ln);
      h0 = halve_pair_d<16>(h0, h2, ln);
      h1 = halve_pair_d<16>(h1, h3, ln);
      double mine = halve_pair_d<8>(h0, h1, ln);
      if constexpr (CH <= 4) mine = bfly_add_d<4>(mine);
      if constexpr (CH <= 2) mine = bfly_add_d<2>(mine);
      const int comp = (((ln >> 4) & 1) << 1) | ((ln >> 3) & 1);
      const bool writer = (ln & 7 & ~(CH - 1)) == 0;
      if (writer) red[wave_u][((ln >> 5) ? RC : 0) + 4 * k + comp] = mine;
      const double ssum = wave_sum_fast_d(s_acc), asum = wave_sum_fast_d(a_acc), d2sum = wave_sum_fast_d(d2_acc);
      if (ln == 0) {
        red[wave_u][2 * RC] = ssum;
        red[wave_u][2 * RC + 1] = asum;
        red[wave_u][2 * RC + 2] = (wig == 0 && wave_u == 0) ? (double)(ngroups + drawn) : 0.0;
        red[wave_u][2 * RC + 3] = d2sum;
        // (loop invariants that are cheaper to form than to keep: opaque copies, or they are spilled around the loop and
        //  their reload waits behind every request in flight)
        unsigned xc = xcc;
        asm volatile("" : "+s"(xc));
        red[wave_u][2 * RC + 4] = (first && wave_u == 0) ? (double)xc : 0.0;
        red[wave_u][2 * RC + 5] = (first && wave_u == 0) ? (double)(xc * xc) : 0.0;
      }
    }
    // (opaque to value numbering: otherwise the fp64 conversions of the rows are kept -- and spilled -- for the last pass)
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(Cr[i]));
    __builtin_amdgcn_sched_barrier(0);
    // ---- group all-reduce of NP doubles (two tagged granules each): the first NP threads publish, ALL threads poll ----
    if (a.prefetch & 4) __builtin_amdgcn_s_setprio(3);
    {
      ++tag;
      int tt = (int)threadIdx.x;
      asm volatile("" : "+v"(tt));
      __syncthreads();
      if constexpr (GW == 1) {
        if (tt < NP) res[tt] = (red[0][tt] + red[1][tt]) + (red[2][tt] + red[3][tt]);
      } else {
        unsigned long long* slot = gbase + (size_t)(tag & 1u) * NG;
        if (tt < NP) {
          const double v = (red[0][tt] + red[1][tt]) + (red[2][tt] + red[3][tt]);
          unsigned long long* mine = slot + ((size_t)wig * NP + tt) * 2;
          const unsigned long long g0 = ((unsigned long long)tag << 32) | (unsigned long long)lo_w(v);
          const unsigned long long g1 = ((unsigned long long)tag << 32) | (unsigned long long)hi_w(v);
          if (same_xcd) {
            __hip_atomic_store(mine, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(mine + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            __hip_atomic_store(mine, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        constexpr int PER = (NG + R3_TPB - 1) / R3_TPB;  // granules per thread
        constexpr int CKQ = PER < 8 ? PER : 8;            // ... polled together
        unsigned spin = 0;
        bool lost = false;
#pragma unroll 1
        for (int q0 = 0; q0 < PER && !lost; q0 += CKQ) {
          unsigned long long x[CKQ];
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < CKQ; ++q) {
              const int idx = tt + R3_TPB * (q0 + q);
              x[q] = __hip_atomic_load(slot + min(idx, NG - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && ((unsigned)(x[q] >> 32) == tag);
            }
            if (ok) break;
            if (++spin > R4_MAXSPIN ||
                ((spin & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              atomicExch(a.err, 1);  // timed out, or another workgroup already did: give up at once
              lost = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
#pragma unroll
          for (int q = 0; q < CKQ; ++q) {
            const int idx = tt + R3_TPB * (q0 + q);
            if (idx < NG) polw[idx] = (unsigned)(x[q] & 0xffffffffull);
          }
        }
        __syncthreads();
        if (tt < NP) {
          double tot = 0.0;
#pragma unroll
          for (int w = 0; w < GW; ++w) tot += mk_d(polw[(w * NP + tt) * 2], polw[(w * NP + tt) * 2 + 1]);  // fixed order
          res[tt] = tot;
        }
      }
      __syncthreads();
    }
    if (GW > 1 && first) {  // placement check (see k_cg_onchip4): plain-store hand-off only when the group shares an XCD
      unsigned xc = xcc;
      asm volatile("" : "+s"(xc));
      const double fx = (double)xc;
      same_xcd = (res[2 * RC + 4] == GW * fx) && (res[2 * RC + 5] == GW * fx * fx) && (a.allow_l2_handoff != 0);
    }
    first = false;
    if (stamp) a.dbg[2] = wall_clock64();
    const int64_t b_next = (int64_t)res[2 * RC + 2];
    {
      int tp = t;
      asm volatile("" : "+v"(tp));
      if (b_next < a.B) {
        request(b_next, tp, 0, std::integral_constant<int, NPF>{}, pre);
      } else {  // (defined on both paths: no value of the finished members stays live)
#pragma unroll
        for (int i = 0; i < NPF; ++i) pre[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (a.prefetch & 1) __builtin_amdgcn_s_setprio(3);

    // ---- the iterations: the chain of k_cg_rspace<.., true> (lo_rspace.hip), operation for operation ----
    const int j = lane & 31, hf = lane >> 5;
    const bool live = j < RC;
    const int jr = live ? j : RC - 1;
    double yj = 0.0, xi = 0.0;
    float nrm;
    {
      double a0 = res[2 * RC + 1], s = res[2 * RC];
      const double d2 = res[2 * RC + 3];
      nrm = sqrtf((float)a0);                             // rhs.norm(2, dim=-2)          :177
      const bool rhs_zero = nrm < a.eps;                  // :178
      if (rhs_zero) nrm = 1.0f;                           // :179
      const double inv = 1.0 / (double)nrm;
      const double w0 = live ? res[jr] * inv : 0.0, u0 = live ? res[RC + jr] * inv : 0.0;
      s *= inv * inv;
      a0 *= inv * inv;
      // r^T r >= r.z / max(dinv) >= r.z / sqrt(sum dinv^2): above this r.z the has_converged mask (:300, 1e-10) cannot hold
      float sa = a.stop_after;
      asm volatile("" : "+s"(sa));
      const double sure_rz = 2.0 * (double)sa * (double)sa * sqrt(d2);
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
      double* myv = gv[wave];
      if (hf == 0) myv[j] = w0;
      __builtin_amdgcn_wave_barrier();
      const double* row0m = mat_s + (size_t)jr * MLD + hf * NH;                    // row j of TinT (this half)
      // c0 = TinT w0 = W^T beta0, g0 = TuT w0 = W^-1 beta0 (beta0 = U^T b^): |beta0|^2 = c0 . g0 and V S^-1 beta0 = Tin g0
      double c0 = own_row_dot(row0m, myv + hf * NH);                               // TinT w0
      double e0 = own_row_dot(row0m + (size_t)1 * RC * MLD, myv + hf * NH);        // g0 = TuT w0
      const double* nrow = row0m + (size_t)2 * RC * MLD;                           // row j of G2 = C^T C
      const double* tucol = mat_s + (size_t)(1 * RC + hf * NH) * MLD + jr;         // column j of TuT (this half of its rows)
      c0 = live ? c0 : 0.0;
      e0 = live ? e0 : 0.0;
      double tau2 = s - lanes32_sum_d(c0 * e0);            // |b_perp|^2 = s - |beta0|^2
      tau2 = tau2 > 0.0 ? tau2 : 0.0;
      // a right-hand side (almost) inside span(C) asks for the dense form (CgCtrl::rs_redo; lo_rspace.hip)
      const bool delicate = !rhs_zero && tau2 < 1e-2 * s;
      const double lam = live ? lam_mine : 1.0;
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
      for (int kk = -1; kk < a.iters; ++kk) {
        if (kk >= 0) {  // x += alpha p (:31);  r -= alpha A p (:264)
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
        const auto pick = [&](double v, int ln_) {
          return mk_d((unsigned)__builtin_amdgcn_readlane((int)lo_w(v), ln_),
                      (unsigned)__builtin_amdgcn_readlane((int)hi_w(v), ln_));
        };
        const double cc = pick(red2, 0), clc = pick(red2, 16), clq = pick(red2, 32);
        const double rzn = fma(cp * cp, tau2, cc);                     // residual_inner_prod :215 / :35-36
        const bool sure = rzn > sure_rz;
        const bool rec_first = kk == 0 && a.close_gran == nullptr && wave == 0;
        const bool rec_last = kk == a.iters - 1 && wave == last_owner;
        const bool own = rec_first || rec_last || !sure;
        float rnn = 0.f;
        if (kk < 0) {
          const float s1f = (float)a0;
          rnn = __builtin_amdgcn_sqrtf(s1f < 0.f ? 0.f : s1f);
        } else if (own) {
          // in the coordinates of C, as the dense form: r = c' r0 + C g with g = Tu (c - c' c0), r^T r = c'^2 a0 + 2 c' g.u0 + g^T G2 g
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
        if (kk >= 0) {                                               // closes iteration kk: beta, residual norm, records
          beta = ((float)rz < a.eps) ? 0.0 : (double)((float)rzn * inv_rz);  // :39-42
          if (rhs_zero) rnn = 0.f;                                   // :299
          rn = rnn;
          if (rec && (rec_first || rec_last)) a.resid_rec[(size_t)kk * a.B + bc] = rn;
        } else {
          beta = 0.0;
          rn = rnn;
          if (wig == 0 && t == 0) a.init_conv[bc] = (rn < a.stop_after) ? 1 : 0;  // :204-205
          close_flags = ((rn < a.stop_after) ? 1u : 0u) | (delicate ? 4u : 0u);
          if (delicate && rec && wave == 0) atomicOr(a.err + 4, 1);  // (CgCtrl::rs_redo, for the host-launched control step)
        }
        // (NaN after the first product, linear_cg.py:199-200: the residual norm is NaN exactly when the coordinates are)
        if (kk == 0 && (rnn != rnn || rzn != rzn)) close_flags |= 2u;
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
      // y = Tin (eta - xi g0): lane i walks column i of TinT (its half of the eigen-indices)
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
      // every wave writes nrm * y for its own x pass
      __builtin_amdgcn_wave_barrier();
      if (hf == 0) myv[j] = yj * (double)nrm;
      __builtin_amdgcn_wave_barrier();
    }
    if (!(a.prefetch & 2)) __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[3] = wall_clock64();
    // ---- x = nrm D^-1 (xi r0 + C y) = D^-1 (xi b + C (nrm y)), fp64 (for small diagonals the two terms cancel) ----
    {
      int l2 = lane;
      asm volatile("" : "+v"(l2));
      const int k2 = l2 & (CH - 1), g2 = l2 / CH;
      const double* myv = gv[wave];
      const double2 ya = *reinterpret_cast<const double2*>(&myv[4 * k2]);
      const double2 yb = *reinterpret_cast<const double2*>(&myv[4 * k2 + 2]);
      g_f* const xb = opaque_uniform((g_f*)(a.xout + (size_t)b * a.N + row0c));
      const bool more = b_next < a.B;
      g_cc* const Cn = (g_cc*)(a.C + ((size_t)(more ? b_next : b) * a.N + row0c) * RC);
      const unsigned loffn = (unsigned)((g2 * RC + 4 * k2) * sizeof(float));
#pragma unroll
      for (int js = 0; js < 4; ++js) {
        constexpr int J0 = NPF / CH;        // first group behind the prefetched instructions
        const int jj = (js + J0) & 3;       // (compile-time after unrolling)
        double p[CH];
#pragma unroll
        for (int m = 0; m < CH; ++m) {
          const f32x4 c4 = Cr[CH * jj + m];
          p[m] = fma((double)c4.w, yb.y, fma((double)c4.z, yb.x, fma((double)c4.y, ya.y, (double)c4.x * ya.x)));
        }
        const double xv = rows_reduce_d<CH>(p, l2);
        const int rw = 64 * jj + RPI * k2 + g2;
        const int row = row0c + rw;
        const double acc = fma(xi, (double)bw[rw], xv);
        if (row >= row0 && row < a.N) {
          const float xval = (float)(acc * (double)dw[rw]);  // :335
          xb[rw] = xval;
          for (int pp = 0; pp < po.n; ++pp)  // (prototype: the gather as peer writes)
            ((g_f*)po.buf[pp])[((size_t)(b + po.member_off)) * a.N + row] = xval;
        }
        if (CH * jj >= NPF) {  // the registers of this group are free: the next member's rows of the same instructions
          g_cc* bqn = opaque_uniform(Cn + (CH * jj + 4) * 1024);
#pragma unroll
          for (int m = 0; m < CH; ++m)
            nxt[CH * jj - NPF + m] = more ? *(g_cf4*)(bqn + loffn + (m - 4) * 1024) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[4] = wall_clock64();
    __syncthreads();  // the next member's matrices / stages may replace these from here on
    b = b_next;
  }
  if (a.close_gran && wig == 0) cg_close_solve(a, ngroups, t);
}

template <int RC, int GW>
int rspace3_go(const OnchipArgs& a, int nwg, hipStream_t st) {
  int per_cu = 0;
  if (LO_OCCUPANCY_CACHED(per_cu, (k_cg_rspace3<RC, GW>), R3_TPB, 0) != hipSuccess || per_cu < 2) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_onchip", st);  // (one scope for the resident single-column kernels: lo_cg_last_executed tells them apart)
  ResidentLaunch guard(st);
  OnchipArgs a2 = a;
  {
    const char* e = getenv("LO_RS_PRIO");
    a2.prefetch = e ? atoi(e) : 3;  // iterations AND the x pass at raised priority (0.1935 -> 0.1894 ms per headline solve)
  }
  hipLaunchKernelGGL((k_cg_rspace3<RC, GW>), dim3(2 * nwg), dim3(R3_TPB), 0, st, a2, g_peer_out);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int RC>
int rspace3_gw(const OnchipArgs& a, int nwg, hipStream_t st) {
  switch (a.GW) {
    case 1: return rspace3_go<RC, 1>(a, nwg, st);
    case 2: return rspace3_go<RC, 2>(a, nwg, st);
    case 4: return rspace3_go<RC, 4>(a, nwg, st);
    case 8: return rspace3_go<RC, 8>(a, nwg, st);
    case 16: return rspace3_go<RC, 16>(a, nwg, st);
    case 32: return rspace3_go<RC, 32>(a, nwg, st);
  }
  return LO_ERR_UNSUPPORTED;
}

}  // namespace

// The diagonal form in the chunk-per-lane layout: groups of up to 32 workgroups (N <= 32768), members of at least 256 rows,
// rows per workgroup = 1024 (a.RW).  LO_ERR_UNSUPPORTED: rspace_launch runs k_cg_rspace<.., true>.
int rspace3_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st) {
  if (!a.RSD || a.x || a.c != 1 || a.ab_rec || !a.xout || a.GW > 32 || a.N < 256 || (int64_t)a.GW * R3_ROWS < a.N)
    return LO_ERR_UNSUPPORTED;
  // granules of a group: 2 x GW x NP x 2 -- inside what rspace_gbuf_bytes reserves per workgroup (432)
  if (RC == 32) return rspace3_gw<32>(a, nwg, st);
  if (RC == 16) return rspace3_gw<16>(a, nwg, st);
  if (RC == 8) return rspace3_gw<8>(a, nwg, st);
  return LO_ERR_UNSUPPORTED;
}

}  // namespace lo

extern "C" int lo_peer_gather_set(float* const* bufs, int n, long long member_offset) {
  if (n < 0 || n > 7 || (n > 0 && !bufs)) return LO_ERR_BADARG;
  for (int i = 0; i < 7; ++i) lo::g_peer_out.buf[i] = i < n ? bufs[i] : nullptr;
  lo::g_peer_out.n = n;
  lo::g_peer_out.member_off = member_offset;
  return LO_OK;
}

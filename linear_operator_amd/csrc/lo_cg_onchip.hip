// lo_cg_onchip.hip -- "operator-resident" preconditioned CG for the headline case:
//   A = C C^T + diag(d)  (C [N, R<=32]),  Woodbury/QR preconditioner Q [N, k<=16],  one right-hand side.
// The multi-kernel engine (lo_cg.hip) streams C and Q from HBM twice per iteration (2.9 MiB per member and
// iteration) -- HBM-bound at ~390 us per iteration for 512 members.  Here a member's whole operator lives
// ON CHIP for all iterations the reference is guaranteed to run (the 11-iteration floor, linear_cg.py:303):
// 8 workgroups (one per CU, 1024 threads) form a group that owns one member; thread t of workgroup w keeps
// row (w * RW + t) of C and of Q -- 48 floats -- plus its elements of x, r, p, z, d, 1/d in VGPRs.  C and Q
// are read from HBM ONCE per solve instead of 22 times.
//
// Every inner product of CG (C^T p, p.Ap, Q^T r with ||r||^2, r.z -- four per iteration) becomes
//   wave:       recursive-halving butterfly (a lane ends with the wave sum of ONE component)
//   workgroup:  16 wave partials through LDS, summed in fixed order
//   group:      each workgroup publishes its partial as 8-byte {value, tag} granules with agent-scope relaxed
//               atomic stores (sc1), every workgroup polls the 8 x n granules with agent-scope loads until the
//               tag equals the phase counter, then sums them in fixed order -> bitwise identical in all 8.
//   A granule is one naturally aligned 8-byte store, so value and tag can never be seen torn and no fence or
//   separate flag is needed (MI355X_MICROARCH.md "hand-off" rows); two slot sets alternate by phase parity so a
//   fast workgroup cannot overwrite a granule a slow one still has to read.  Polls are bounded: on timeout an
//   error word is set and the host falls back to the streaming engine.
//
// The arithmetic is the reference's (same masked alpha/beta, same update order, linear_cg.py:245-300); only the
// summation order of the inner products differs.  After the last guaranteed iteration the state (x, r, p, z and
// the per-member scalars) is written back in the streaming engine's layout; the batch-global stop rule is then
// evaluated by lo_cg.hip's control kernel, which continues with the streaming loop in the (rare) case that the
// tolerance is not yet met.
#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include "lo_device.h"
#include <string.h>

#include "lo_internal.h"
#include "lo_cg_onchip.h"

namespace lo {

constexpr int OC_TPB = 1024;
constexpr int OC_GW = 8;          // workgroups per member
constexpr int OC_WAVES = OC_TPB / 64;
constexpr unsigned OC_MAXSPIN = 1u << 20;  // ~0.5 s of polling: co-residency was lost (never seen on a dedicated GPU)


// Wave reduce-scatter of the products a[j] * s with the operand row in LDS (read as float4): on exit lane l holds
// the sum over the 64 lanes of component (l >> (6 - log2 n)).  The first halving step forms the products on the
// fly, so only n/2 temporaries are live.  Used for the C row, which lives in LDS so that its 32 floats do not
// occupy a quarter of the 128-VGPR budget.
template <int n>
__device__ __forceinline__ float wave_reduce_scatter_prod_lds(const float* __restrict__ a, float s) {
  const int lane = threadIdx.x & 63;
  constexpr int h0 = n / 2;
  float v[h0];
#pragma unroll
  for (int j = 0; j < h0; j += 4) {
    const float4 lo4 = *reinterpret_cast<const float4*>(a + j);
    const float4 hi4 = *reinterpret_cast<const float4*>(a + j + h0);
    v[j + 0] = halve_pair<32>(lo4.x * s, hi4.x * s, lane);
    v[j + 1] = halve_pair<32>(lo4.y * s, hi4.y * s, lane);
    v[j + 2] = halve_pair<32>(lo4.z * s, hi4.z * s, lane);
    v[j + 3] = halve_pair<32>(lo4.w * s, hi4.w * s, lane);
  }
  halving_steps<h0, 16, h0>(v, lane);
  return v[0];
}

// The same with the operand row in registers (Q row).
template <int n>
__device__ __forceinline__ float wave_reduce_scatter_prod(const float (&a)[n], float s) {
  const int lane = threadIdx.x & 63;
  constexpr int h0 = n / 2;
  float v[h0];
#pragma unroll
  for (int j = 0; j < h0; ++j) v[j] = halve_pair<32>(a[j] * s, a[j + h0] * s, lane);
  halving_steps<h0, 16, h0>(v, lane);
  return v[0];
}

constexpr int PF_STRIDE = 64;  // bytes between L2 prefetch touches
constexpr int OC_CLD = 36;  // padded LDS row stride of C (floats): 144-B rows -> conflict-free ds_read_b128

struct OcShared {
  float red[OC_WAVES][40];
  float part[40];
  float gath[OC_GW][40];
  float res[40];
};

// Group-wide sum of `cnt` (<= 40) workgroup partials sitting in sh.part[0..cnt); result in sh.res[0..cnt),
// identical in all workgroups of the group.  Must be called by all 1024 threads.
// `same_xcd`: all 8 workgroups were VERIFIED (XCC_ID exchanged through the agent-scope path first) to run on one
// XCD, i.e. behind one L2.  Then the granule may be published with a plain store (write-through L1 -> stays in
// that L2) and is picked up by the pollers' L1-bypassing loads as an L2 hit instead of a fabric round trip.
// Without the verification the agent-scope (sc1, write-through to the fabric) store is used: correct anywhere.
__device__ __forceinline__ void group_exchange(OcShared& sh, int cnt, unsigned long long* gslot_base, int wig,
                                               unsigned tag, int* err, bool same_xcd = false) {
  const int t = threadIdx.x;
  unsigned long long* slot = gslot_base + (size_t)(tag & 1u) * OC_GW * 40;
  if (t < cnt) {
    const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(sh.part[t]);
    if (same_xcd)
      __hip_atomic_store(slot + (size_t)wig * 40 + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)wig * 40 + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t < OC_GW * cnt) {
    const int w = t / cnt, i = t % cnt;
    const unsigned long long* src = slot + (size_t)w * 40 + i;
    unsigned long long g = 0;
    unsigned spin = 0;
    for (;;) {
      g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(g >> 32) == tag) break;
      if (++spin > OC_MAXSPIN ||
          ((spin & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        atomicExch(err, 1);  // timed out, or another workgroup already did: give up at once
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    sh.gath[w][i] = __uint_as_float((unsigned)(g & 0xffffffffull));
  }
  __syncthreads();
  if (t < cnt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < OC_GW; ++w) s += sh.gath[w][t];
    sh.res[t] = s;
  }
  __syncthreads();
}

// workgroup partial of n per-thread components + `ns` scalars -> sh.part[0 .. n+ns)
template <int n>
__device__ __forceinline__ void wg_partial(OcShared& sh, const float (&a)[n], float mult, const float* scal, int ns) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int sh_bits = (n == 32) ? 1 : (n == 16) ? 2 : (n == 8) ? 3 : (n == 4) ? 4 : 5;
  const float mine = wave_reduce_scatter_prod<n>(a, mult);
  if ((lane & ((1 << sh_bits) - 1)) == 0) sh.red[wave][lane >> sh_bits] = mine;
  for (int j = 0; j < ns; ++j) {
    const float sv = wave_sum_fast(scal[j]);  // DPP/permlane butterfly: no LDS-crossbar latency chain
    if (lane == 0) sh.red[wave][n + j] = sv;
  }
  __syncthreads();
  if (t < n + ns) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < OC_WAVES; ++w) s += sh.red[w][t];
    sh.part[t] = s;
  }
  __syncthreads();
}

template <int n>
__device__ __forceinline__ void wg_partial_lds(OcShared& sh, const float* a, float mult, const float* scal, int ns) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int sh_bits = (n == 32) ? 1 : (n == 16) ? 2 : (n == 8) ? 3 : (n == 4) ? 4 : 5;
  const float mine = wave_reduce_scatter_prod_lds<n>(a, mult);
  if ((lane & ((1 << sh_bits) - 1)) == 0) sh.red[wave][lane >> sh_bits] = mine;
  for (int j = 0; j < ns; ++j) {
    const float sv = wave_sum_fast(scal[j]);  // DPP/permlane butterfly: no LDS-crossbar latency chain
    if (lane == 0) sh.red[wave][n + j] = sv;
  }
  __syncthreads();
  if (t < n + ns) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < OC_WAVES; ++w) s += sh.red[w][t];
    sh.part[t] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ void wg_partial_scalars(OcShared& sh, const float* scal, int ns) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int j = 0; j < ns; ++j) {
    const float sv = wave_sum_fast(scal[j]);  // DPP/permlane butterfly: no LDS-crossbar latency chain
    if (lane == 0) sh.red[wave][j] = sv;
  }
  __syncthreads();
  if (t < ns) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < OC_WAVES; ++w) s += sh.red[w][t];
    sh.part[t] = s;
  }
  __syncthreads();
}

template <int RC, int RK>
__global__ __launch_bounds__(OC_TPB) void k_cg_onchip(OnchipArgs a) {
  __shared__ OcShared sh;
  __shared__ float pf_sink[64];           // landing pad of the L2 prefetch loads (never read)
  __shared__ float c_s[OC_TPB * OC_CLD];  // this workgroup's rows of C (144 KiB of the CU's 160 KiB LDS); Q rows in VGPRs
  const int wg = blockIdx.x;
  // keep the 8 workgroups of a group on one XCD (block b runs on XCD b % 8; speed only)
  const int xcd = wg % 8, j = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / OC_GW;
  const int grp = xcd * groups_per_xcd + j / OC_GW;
  const int wig = j % OC_GW;
  const int ngroups = groups_per_xcd * 8;
  if (j / OC_GW >= groups_per_xcd) return;  // grid not a multiple of 64: spare workgroups idle
  const int t = threadIdx.x;
  unsigned long long* gslot = a.gbuf + (size_t)grp * 2 * OC_GW * 40;
  unsigned tag = 0;
  // placement check (speed only, never assumed): do the 8 workgroups of this group share an XCD / L2?
  bool same_xcd = false;
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t == 0) sh.part[0] = (float)xcc;
    __syncthreads();
    // each workgroup publishes its id; gather WITHOUT summing: compare the 8 raw values
    const unsigned tg = ++tag;
    unsigned long long* slot = gslot + (size_t)(tg & 1u) * OC_GW * 40;
    if (t == 0) {
      const unsigned long long g = ((unsigned long long)tg << 32) | (unsigned long long)xcc;
      __hip_atomic_store(slot + (size_t)wig * 40, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < OC_GW) {
      unsigned long long g = 0;
      unsigned spin = 0;
      for (;;) {
        g = __hip_atomic_load(slot + (size_t)t * 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g >> 32) == tg) break;
        if (++spin > OC_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err, 1);  // timed out, or another workgroup already did: give up at once
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      sh.gath[t][0] = (float)(unsigned)(g & 0xffffffffull);
    }
    __syncthreads();
    bool same = true;
#pragma unroll
    for (int w = 1; w < OC_GW; ++w) same = same && (sh.gath[w][0] == sh.gath[0][0]);
    same_xcd = same && (a.allow_l2_handoff != 0);
    __syncthreads();
  }

  for (int64_t b = grp; b < a.B; b += ngroups) {
    const bool stamp = a.dbg && b == a.dbg_member && wig == 0 && t == 0;
    if (stamp) a.dbg[0] = wall_clock64();
    const int row = wig * a.RW + t;
    const bool valid = (t < a.RW) && (row < a.N);
    float Qr[RK];
    float* crow = c_s + t * OC_CLD;
    float dv = 0.f, dinvv = 0.f, rhsv = 0.f;
    {
      // coalesced: both row blocks are in flight together; Q passes through the (still empty) C area of LDS
      // on its way to the owning thread's registers
      const int row0 = wig * a.RW;
      const int nv = max(0, min(a.RW, a.N - row0));
      float4 cq[RC / 4], qq[RK / 4];
      rows_issue<RK, OC_TPB>(a.Q + ((size_t)b * a.N + row0) * RK, nv, qq);
      rows_issue<RC, OC_TPB>(a.C + ((size_t)b * a.N + row0) * RC, nv, cq);
      if (valid) {
        dv = (a.d_mode == LO_DIAG_FULL) ? a.d[(size_t)b * a.N + row] : (a.d_mode == LO_DIAG_CONST ? a.d[b] : 0.f);
        dinvv = (a.dinv_mode == LO_DIAG_FULL) ? a.dinv[(size_t)b * a.N + row] : a.dinv[b];
        rhsv = a.rhs[(size_t)b * a.N + row];
      }
      rows_commit<RK, RK + 4, OC_TPB>(c_s, qq);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < RK / 4; ++i) {
        const float4 q4 = *reinterpret_cast<const float4*>(c_s + t * (RK + 4) + 4 * i);
        Qr[4 * i] = q4.x; Qr[4 * i + 1] = q4.y; Qr[4 * i + 2] = q4.z; Qr[4 * i + 3] = q4.w;
      }
      __syncthreads();
      rows_commit<RC, OC_CLD, OC_TPB>(c_s, cq);
    }

    __syncthreads();
    if (stamp) a.dbg[1] = wall_clock64();
    // ---- initialisation (linear_cg.py:177-215) ----
    float sc[2];
    sc[0] = rhsv * rhsv;
    wg_partial_scalars(sh, sc, 1);
    group_exchange(sh, 1, gslot, wig, ++tag, a.err, same_xcd);
    float nrm = sqrtf(sh.res[0]);                           // rhs.norm(2, dim=-2)          :177
    const bool rhs_zero = nrm < a.eps;                      // :178
    if (rhs_zero) nrm = 1.0f;                               // :179
    float r = rhsv / nrm;                                   // :182 (x0 = 0 -> residual = rhs)
    float x = 0.f;
    sc[0] = r * r;
    sc[1] = dinvv * r * r;
    wg_partial<RK>(sh, Qr, r, sc, 2);
    group_exchange(sh, RK + 2, gslot, wig, ++tag, a.err, same_xcd);
    float rr = sh.res[RK];
    bool conv = sqrtf(rr) < a.stop_after;                   // :204-205
    if (wig == 0 && t == 0) a.init_conv[b] = conv ? 1 : 0;
    float z = dinvv * r;                                    // precondition_closure :135-140
    float uu = 0.f;
#pragma unroll
    for (int i = 0; i < RK; ++i) {
      z = fmaf(-Qr[i], sh.res[i], z);
      uu = fmaf(sh.res[i], sh.res[i], uu);
    }
    // r.z = sum r o (r/d - Q u) = sum r^2/d - ||Q^T r||^2 : both terms come out of the exchange above, which
    // saves a group-wide reduction per iteration (same value up to rounding; residual_inner_prod :215 / :35-36)
    float rz = sh.res[RK + 1] - uu;
    float p = z, beta = 0.f, alpha = 0.f, rn = sqrtf(rr);

    // L2 prefetch plan for the next member of this group (see the iteration loop)
    int pf_lines = 0, pf_chunk = 0, pf_lc = 0, pf_lq = 0, pf_lv = 0;
    const char *pf_c = nullptr, *pf_q = nullptr, *pf_v0 = nullptr, *pf_v1 = nullptr, *pf_v2 = nullptr;
    if (a.prefetch && b + ngroups < a.B && a.iters > 0) {
      const int64_t nb = b + ngroups;
      const int row0 = wig * a.RW;
      const int nv = max(0, min(a.RW, a.N - row0));
      pf_lc = (nv * RC * 4) / PF_STRIDE;
      pf_lq = (nv * RK * 4) / PF_STRIDE;
      pf_lv = (nv * 4) / PF_STRIDE;
      pf_c = reinterpret_cast<const char*>(a.C + ((size_t)nb * a.N + row0) * RC);
      pf_q = reinterpret_cast<const char*>(a.Q + ((size_t)nb * a.N + row0) * RK);
      pf_v0 = reinterpret_cast<const char*>(a.rhs + (size_t)nb * a.N + row0);
      int nvec = 1;
      if (a.d_mode == LO_DIAG_FULL) {
        pf_v1 = reinterpret_cast<const char*>(a.d + (size_t)nb * a.N + row0);
        nvec = 2;
        if (a.dinv_mode == LO_DIAG_FULL) {
          pf_v2 = reinterpret_cast<const char*>(a.dinv + (size_t)nb * a.N + row0);
          nvec = 3;
        }
      } else if (a.dinv_mode == LO_DIAG_FULL) {
        pf_v1 = reinterpret_cast<const char*>(a.dinv + (size_t)nb * a.N + row0);
        nvec = 2;
      }
      pf_lines = pf_lc + pf_lq + nvec * pf_lv;
      pf_chunk = min(OC_TPB, (pf_lines + a.iters - 1) / a.iters);
    }
    if (stamp) a.dbg[2] = wall_clock64();
    for (int k = 0; k < a.iters; ++k) {
      if (k > 0) p = fmaf(p, beta, z);                      // p.mul_(beta).add_(z)  :46
      sc[0] = dv * p * p;
      wg_partial_lds<RC>(sh, crow, p, sc, 1);
      group_exchange(sh, RC + 1, gslot, wig, ++tag, a.err, same_xcd);  // t = C^T p  and  sum d p^2
      if (pf_lines > 0) {
        // L2 prefetch of this workgroup's rows of the NEXT member, one slice per iteration: one 4-byte
        // global_load_lds per 128-byte line (no VGPR result, nothing waits on it); issued here so that the loads
        // have landed before the next vector-memory wait (the poll of the second exchange)
        const int line = k * pf_chunk + t;
        if (t < pf_chunk && line < pf_lines) {
          const char* src;
          if (line < pf_lc) src = pf_c + (size_t)line * PF_STRIDE;
          else if (line < pf_lc + pf_lq) src = pf_q + (size_t)(line - pf_lc) * PF_STRIDE;
          else {
            const int v = (line - pf_lc - pf_lq) / pf_lv, o = (line - pf_lc - pf_lq) % pf_lv;
            src = (v == 0 ? pf_v0 : v == 1 ? pf_v1 : pf_v2) + (size_t)o * PF_STRIDE;
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)pf_sink, 4, 0, 0);
        }
      }
      float y = dv * p;                                     // A p = C t + d o p     added_diag...py:72-76
      float ct = 0.f, tt = 0.f;
#pragma unroll
      for (int i = 0; i < RC; i += 4) {
        const float4 c4 = *reinterpret_cast<const float4*>(crow + i);
        const float4 t4 = *reinterpret_cast<const float4*>(&sh.res[i]);
        ct = fmaf(c4.x, t4.x, ct);
        ct = fmaf(c4.y, t4.y, ct);
        ct = fmaf(c4.z, t4.z, ct);
        ct = fmaf(c4.w, t4.w, ct);
        tt = fmaf(t4.x, t4.x, tt);
        tt = fmaf(t4.y, t4.y, tt);
        tt = fmaf(t4.z, t4.z, tt);
        tt = fmaf(t4.w, t4.w, tt);
      }
      y += ct;
      // p.Ap = p^T C C^T p + p^T D p = ||C^T p||^2 + sum d p^2   (:250-251; no extra group reduction)
      const float pAp = tt + sh.res[RC];
      alpha = (pAp < a.eps) ? 0.f : rz / pAp;               // :254-257
      if (conv) alpha = 0.f;                                // :260
      r = fmaf(-alpha, y, r);                               // :264
      x = fmaf(alpha, p, x);                                // :31
      sc[0] = r * r;
      sc[1] = dinvv * r * r;
      wg_partial<RK>(sh, Qr, r, sc, 2);
      group_exchange(sh, RK + 2, gslot, wig, ++tag, a.err, same_xcd);  // Q^T r, ||r||^2, sum r^2/d
      rr = sh.res[RK];
      z = dinvv * r;
      float uu2 = 0.f;
#pragma unroll
      for (int i = 0; i < RK; ++i) {
        z = fmaf(-Qr[i], sh.res[i], z);
        uu2 = fmaf(sh.res[i], sh.res[i], uu2);
      }
      const float rzn = sh.res[RK + 1] - uu2;               // r.z    :35-36
      beta = (rz < a.eps) ? 0.f : rzn / rz;                 // :39-42
      rz = rzn;
      rn = sqrtf(rr);                                       // :298
      if (rhs_zero) rn = 0.f;                               // :299
      conv = rn < a.stop_after;                             // :300
      if (wig == 0 && t == 0) a.resid_rec[(size_t)k * a.B + b] = rn;
    }

    if (stamp) a.dbg[3] = wall_clock64();
    // ---- write the state back in the streaming engine's layout ----
    if (valid) {
      const size_t o = (size_t)b * a.N + row;
      a.x[o] = x;
      a.r[o] = r;
      a.p[o] = p;
      a.z[o] = z;
    }
    if (wig == 0 && t == 0) {
      a.rhs_norm[b] = nrm;
      a.rhs_is_zero[b] = rhs_zero ? 1 : 0;
      a.rz[b] = rz;
      a.alpha[b] = alpha;
      a.beta[b] = beta;
      a.resid_norm[b] = rn;
      a.has_conv[b] = conv ? 1 : 0;
    }
    if (stamp) a.dbg[4] = wall_clock64();
  }
}

size_t onchip_gbuf_bytes(int ngroups) { return (size_t)ngroups * 2 * OC_GW * 40 * sizeof(unsigned long long); }

bool onchip_eligible(int RC, int RK, int64_t N, int64_t c) {
  const bool rc_ok = (RC == 8 || RC == 16 || RC == 32);
  const bool rk_ok = (RK == 4 || RK == 8 || RK == 16);
  return rc_ok && rk_ok && c == 1 && N <= (int64_t)OC_GW * OC_TPB && N >= 1024;
}

namespace {
std::mutex g_res_mu;
hipStream_t g_res_last = nullptr;
int g_res_last_dev = -1;
bool g_res_have = false;
hipEvent_t g_res_ev[16] = {};
}  // namespace

thread_local bool tls_graph_capture = false;

ResidentLaunch::ResidentLaunch(hipStream_t st) {
  g_res_mu.lock();
  if (tls_graph_capture) return;  // (the captured launch does not execute; the replay goes through the caller's guard)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
  if (g_res_have && g_res_last_dev == dev && g_res_last != st && !getenv("LO_NO_RESIDENT_ORDER")) {
    if (!g_res_ev[dev] && hipEventCreateWithFlags(&g_res_ev[dev], hipEventDisableTiming) != hipSuccess) g_res_ev[dev] = nullptr;
    // (a stream the caller has destroyed in the meantime makes the record fail: nothing left to wait for)
    if (g_res_ev[dev] && hipEventRecord(g_res_ev[dev], g_res_last) == hipSuccess)
      (void)hipStreamWaitEvent(st, g_res_ev[dev], 0);
    else
      (void)hipGetLastError();
  }
  g_res_last = st;
  g_res_last_dev = dev;
  g_res_have = true;
}

ResidentLaunch::~ResidentLaunch() { g_res_mu.unlock(); }

// The same-XCD hand-off publishes granules with WORKGROUP-scope stores that other workgroups of the XCD read through the
// shared L2: outside the HIP memory model, correct only where a CU's vector L1 is write-through into the XCD's L2.  That
// holds on gfx950 (verified on MI355X: tools/fuzz_resident.py, the GPU test suite) -- so the path is OPT-IN per
// verified architecture string; anywhere else (and with LO_OC_NO_L2_HANDOFF) every granule is an agent-scope store.
// Cost of the agent-scope stores on MI355X: profiles/r04/l2_handoff_cost.txt.
int onchip_l2_handoff_allowed() {
  if (getenv("LO_OC_NO_L2_HANDOFF")) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  static int verified[64] = {0};  // 0 unknown, 1 yes, 2 no
  if (verified[dev] == 0) {
    hipDeviceProp_t prop;
    verified[dev] = 2;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) verified[dev] = 1;
  }
  return verified[dev] == 1 ? 1 : 0;
}

int onchip_num_workgroups() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  // (hipGetDeviceProperties fills a ~1.5 KB structure and costs tens of microseconds: asked once per device)
  static int cu_count[64] = {0};
  if (dev < 0 || dev >= 64) return 0;
  if (cu_count[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cu_count[dev] = n;
  }
  int cus = cu_count[dev];
  // LO_OC_RESERVE_CUS=<n>: leave n CUs' worth of workgroup slots unused so that a concurrently running collective
  // (RCCL's all-gather kernel on its own stream) finds room WITHOUT displacing workgroups of a resident group -- a
  // displaced workgroup stalls its whole group until the other kernel ends.  The spare slots end up scattered over the
  // CUs (the dispatcher balances), each big enough for a 256-thread workgroup of <= 256 VGPRs.
  if (const char* e = getenv("LO_OC_RESERVE_CUS")) cus = std::max(64, cus - std::max(0, atoi(e)));
  return (cus / 32) * 32;  // 2 workgroups per CU: a multiple of 8 XCDs x 8 workgroups per group
}

int onchip_launch(int RC, int RK, const OnchipArgs& a, int nwg, hipStream_t st) {
  dim3 grid(nwg), block(OC_TPB);
  ResidentLaunch guard(st);
  LO_PROF_BEGIN("cg_onchip", st);
#define LO_OC(C_, K_) hipLaunchKernelGGL((k_cg_onchip<C_, K_>), grid, block, 0, st, a)
  if (RC == 32 && RK == 16) LO_OC(32, 16);
  else if (RC == 32 && RK == 8) LO_OC(32, 8);
  else if (RC == 32 && RK == 4) LO_OC(32, 4);
  else if (RC == 16 && RK == 16) LO_OC(16, 16);
  else if (RC == 16 && RK == 8) LO_OC(16, 8);
  else if (RC == 16 && RK == 4) LO_OC(16, 4);
  else if (RC == 8 && RK == 16) LO_OC(8, 16);
  else if (RC == 8 && RK == 8) LO_OC(8, 8);
  else if (RC == 8 && RK == 4) LO_OC(8, 4);
  else return LO_ERR_UNSUPPORTED;
#undef LO_OC
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // namespace lo

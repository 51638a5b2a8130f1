// lo_group_reduce.h -- building blocks of the operator-resident kernels that own FOUR rows per thread (256-thread
// workgroups, groups of GW workgroups per batch member): the wave reduce-scatter and the group all-reduce through
// tagged 8-byte granules (see lo_cg_onchip.hip / lo_cg_onchip4.hip for the design notes).
#pragma once
#include "lo_device.h"

namespace lo {

constexpr int R4_TPB = 256;
constexpr int R4_NR = 4;                 // rows per thread
constexpr int R4_WAVES = R4_TPB / 64;    // 4
constexpr int R4_ROWS = R4_TPB * R4_NR;  // rows per workgroup
constexpr int R4_MAXGW = 32;  // (the root-form CG kernel also takes groups of 64: lane-parallel all-reduce below)
constexpr int R4_SLOT = 40;
constexpr unsigned R4_MAXSPIN = 1u << 20;  // ~0.5 s of polling: co-residency was lost (never seen on a dedicated GPU)

struct alignas(16) R4Shared {
  float red[R4_WAVES][R4_SLOT];
  float res[R4_SLOT];
};

// physical 16-byte slot of logical slot q of Q row r (NQ slots per row)
template <int NQ>
__device__ __forceinline__ int q_slot(int r, int q) {
  if constexpr (NQ == 4) return r * 4 + (q ^ ((r >> 2) & 3));
  else if constexpr (NQ == 2) return r * 2 + (q ^ ((r >> 3) & 1));
  else return r;
}

// wave reduce-scatter of n register values (destroyed): lane l ends with the wave sum of component l >> (6 - log2 n)
// The components come from a generator so that only n/2 temporaries are ever live (component j and j + n/2 are
// formed right before their first halving step).
template <int n, class Gen>
__device__ __forceinline__ float r4_wave_rs_t(Gen gen, const int lane) {
  constexpr int h0 = n / 2;
  float w[h0];
#pragma unroll
  for (int j = 0; j < h0; ++j) w[j] = halve_pair<32>(gen(j), gen(j + h0), lane);
  halving_steps<h0, 16, h0>(w, lane);
  return w[0];
}
template <int n, class Gen>
__device__ __forceinline__ float r4_wave_rs(Gen gen) {
  return r4_wave_rs_t<n>(gen, (int)(threadIdx.x & 63));
}

struct R4Group {
  unsigned long long* gslot;  // [2][GW][R4_SLOT] granules of this group
  int wig;
  long long* dbg;  // phase timers of the stamped member (or nullptr)
  unsigned tag;
  int* err;
  bool same_xcd;
};

// Group all-reduce for large groups (up to 64 workgroups) and a small payload: reduce-scatter + all-gather through tagged
// granules, in two halves so that independent work can be placed between the publication and the wait.  Workgroup e owns
// payload entry e: the 64 lanes of its first wave fetch that entry of all GW workgroups (one granule per lane -- the
// fan-in costs lanes, not registers), sum them with the fixed-order wave butterfly and publish the total; everybody then
// reads the cnt totals.  sh.red[w][0..cnt) hold the wave partials; result in sh.res[0..cnt).
struct PfSlots {
  unsigned long long* slot;
  unsigned long long* tot;
  unsigned tag;
};

__device__ __forceinline__ void pf_store(const R4Group& g, unsigned tag, unsigned long long* dst, float v) {
  const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (g.same_xcd) __hip_atomic_store(dst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(dst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every lane of the wave takes part; lanes with active == false only vote
__device__ __forceinline__ float pf_wait(const R4Group& g, unsigned tag, const unsigned long long* src, bool active) {
  unsigned long long x = 0;
  unsigned spin = 0;
  for (;;) {
    bool ok = true;
    if (active) {
      x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = (unsigned)(x >> 32) == tag;
    }
    if (__all(ok)) break;
    if (++spin > R4_MAXSPIN ||
        ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(g.err, 1);  // timed out, or another workgroup already did: give up at once
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return active ? __uint_as_float((unsigned)(x & 0xffffffffull)) : 0.f;
}

template <int GW>
__device__ __forceinline__ PfSlots pf_publish(R4Shared& sh, int cnt, R4Group& g) {
  const int t = threadIdx.x;
  PfSlots ps;
  ps.tag = ++g.tag;
  __syncthreads();
  ps.slot = g.gslot + (size_t)(ps.tag & 1u) * (GW + 1) * R4_SLOT;  // GW partial arrays | totals
  ps.tot = ps.slot + (size_t)GW * R4_SLOT;
  if (t < cnt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < R4_WAVES; ++w) s += sh.red[w][t];
    pf_store(g, ps.tag, ps.slot + (size_t)g.wig * R4_SLOT + t, s);
  }
  return ps;
}

template <int GW>
__device__ __forceinline__ void pf_collect(R4Shared& sh, int cnt, R4Group& g, const PfSlots& ps) {
  const int t = threadIdx.x;
  if (t < 64) {
    for (int e = g.wig; e < cnt; e += GW) {  // entries this workgroup owns
      const float v = pf_wait(g, ps.tag, ps.slot + (size_t)t * R4_SLOT + e, t < GW);
      const float total = wave_sum_fast(v);
      if (t == 0) pf_store(g, ps.tag, ps.tot + e, total);
    }
    const float r = pf_wait(g, ps.tag, ps.tot + t, t < cnt);
    if (t < cnt) sh.res[t] = r;
  }
  __syncthreads();
}

// Second half of an all-reduce: sh.red[w][0..cnt) hold the wave partials.  Thread t < cnt sums them (fixed
// order), publishes the granule, polls the same component of every workgroup of the group and sums those in fixed
// order -> sh.res[t], bitwise identical in all workgroups.  Ends with a barrier.
// (the _t variants take the thread index from the caller: a kernel with several phases passes a per-phase opaque copy
// so that the addresses derived from it are not hoisted to the kernel entry and kept live -- or spilled -- throughout)
template <int GW>
__device__ __forceinline__ void r4_group_sum_t(R4Shared& sh, int cnt, R4Group& g, const int t) {
  if constexpr (GW == 64) {  // (cnt <= 64: one owner workgroup per entry; granule layout [2][GW + 1][R4_SLOT])
    const PfSlots ps = pf_publish<64>(sh, cnt, g);
    pf_collect<64>(sh, cnt, g, ps);
    return;
  }
  const unsigned tag = ++g.tag;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const bool stamp = g.dbg && t == 0;
  if (stamp) c0 = wall_clock64();
  __syncthreads();
  if (stamp) c1 = wall_clock64();
  if (t < cnt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < R4_WAVES; ++w) s += sh.red[w][t];
    unsigned long long* slot = g.gslot + (size_t)(tag & 1u) * GW * R4_SLOT;
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(s);
    if (g.same_xcd)
      __hip_atomic_store(slot + (size_t)g.wig * R4_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)g.wig * R4_SLOT + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp) c2 = wall_clock64();
    float tot = 0.f;
    unsigned spin = 0;
    if constexpr (GW == 1) {
      tot = s;  // a group of one: nothing to wait for
    } else if constexpr (GW <= 16) {
      float vals[GW];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
          const unsigned long long x =
              __hip_atomic_load(slot + (size_t)w * R4_SLOT + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(x >> 32) == tag);
          vals[w] = __uint_as_float((unsigned)(x & 0xffffffffull));
        }
        if (ok) break;
        if (++spin > R4_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(g.err, 1);  // timed out, or another workgroup already did: give up at once
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int w = 0; w < GW; ++w) tot += vals[w];
    } else {
      // large groups: wait for all tags first, then read the (now final: a granule of this parity is not rewritten
      // before every workgroup has finished this all-reduce) values again -- 32 values need not stay in registers
      // (32-bit halves of the granules: tag = upper word, value = lower word; the writer stores both with one 64-bit
      // store, so a matching tag means the value word next to it is the one of this all-reduce)
      const unsigned* words = reinterpret_cast<const unsigned*>(slot);
      for (;;) {
        unsigned bad = 0;  // (no short-circuit: all tag loads must be in flight together)
#pragma unroll
        for (int w = 0; w < GW; ++w)
          bad |= __hip_atomic_load(words + 2 * ((size_t)w * R4_SLOT + t) + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) ^ tag;
        if (bad == 0) break;
        if (++spin > R4_MAXSPIN ||
            ((spin & 1023u) == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(g.err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int h = 0; h < GW; h += 16) {  // 16 loads in flight, summed in the fixed order w = 0 .. GW-1
        float v[16];
#pragma unroll
        for (int w = 0; w < 16; ++w)
          v[w] = __uint_as_float(__hip_atomic_load(words + 2 * ((size_t)(h + w) * R4_SLOT + t), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += v[w];
      }
    }
    sh.res[t] = tot;
    if (stamp) {
      c3 = wall_clock64();
      g.dbg[5] += c1 - c0;  // waiting for the slowest wave of this workgroup
      g.dbg[6] += c2 - c1;  // wave-partial sum + publish
      g.dbg[7] += c3 - c2;  // poll + sum
    }
  }
  __syncthreads();
}

template <int GW>
__device__ __forceinline__ void r4_group_sum(R4Shared& sh, int cnt, R4Group& g) {
  r4_group_sum_t<GW>(sh, cnt, g, (int)threadIdx.x);
}

// all-reduce of n generated components + ns scalars over the whole group -> sh.res[0 .. n + ns)
template <int GW, int n, class Gen>
__device__ __forceinline__ void r4_allreduce_t(R4Shared& sh, Gen gen, const float* scal, int ns, R4Group& g,
                                               const int t) {
  const int lane = t & 63, wave = t >> 6;
  constexpr int sh_bits = (n == 32) ? 1 : (n == 16) ? 2 : (n == 8) ? 3 : 4;
  const float mine = r4_wave_rs_t<n>(gen, lane);
  if ((lane & ((1 << sh_bits) - 1)) == 0) sh.red[wave][lane >> sh_bits] = mine;
  for (int j = 0; j < ns; ++j) {
    const float sv = wave_sum_fast(scal[j]);
    if (lane == 0) sh.red[wave][n + j] = sv;
  }
  r4_group_sum_t<GW>(sh, n + ns, g, t);
}
template <int GW, int n, class Gen>
__device__ __forceinline__ void r4_allreduce(R4Shared& sh, Gen gen, const float* scal, int ns, R4Group& g) {
  r4_allreduce_t<GW, n>(sh, gen, scal, ns, g, (int)threadIdx.x);
}

// All-reduce of NS <= 4 scalars over a group of GW <= 16 workgroups with ONE barrier and no LDS round trip for the
// result: every wave sums the four wave partials itself (LDS broadcast reads), the first wave publishes the workgroup's
// granules, and EVERY wave polls the group's GW x NS granules with one granule per lane (lane = 4 w + j) and sums over
// the workgroups with the fixed xor butterflies over the lane bits 2.. -- the same tree in every wave of every
// workgroup, so the totals are bitwise identical everywhere.  The totals come back in registers (wave-uniform), the
// second barrier and the sh.res write / read of r4_group_sum are gone.  The wave partials are double-buffered by the
// phase parity (sh.red[wave][4 parity + j]): a wave can only write parity p again after the barrier of the phase in
// between, which every wave passes after it has read its values of parity p.
template <int GW, int NS>
__device__ __forceinline__ void r4_allreduce_small_t(R4Shared& sh, const float* scal, float* out, R4Group& g,
                                                     const int t) {
  static_assert(GW <= 16 && NS <= 4, "one granule per lane");
  const int lane = t & 63, wave = t >> 6;
  const unsigned tag = ++g.tag;
  const int par = (int)(tag & 1u);
  float sv[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) sv[j] = wave_sum_fast(scal[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NS; ++j) sh.red[wave][4 * par + j] = sv[j];
  }
  __syncthreads();
  const int j = lane & 3, w = lane >> 2;
  float s = 0.f;
  if (j < NS) {
#pragma unroll
    for (int q = 0; q < R4_WAVES; ++q) s += sh.red[q][4 * par + j];  // fixed order, the same bits in every wave
  }
  float v = 0.f;
  if constexpr (GW == 1) {
    v = (w == 0) ? s : 0.f;  // a group of one: nothing to publish or to wait for
  } else {
    unsigned long long* slot = g.gslot + (size_t)par * GW * R4_SLOT;
    if (wave == 0 && lane < NS) pf_store(g, tag, slot + (size_t)g.wig * R4_SLOT + lane, s);
    v = pf_wait(g, tag, slot + (size_t)w * R4_SLOT + j, w < GW && j < NS);
  }
  if constexpr (GW > 1) v = bfly_add<4>(v);
  if constexpr (GW > 2) v = bfly_add<8>(v);
  if constexpr (GW > 4) v = bfly_add<16>(v);
  if constexpr (GW > 8) v = bfly_add<32>(v);
#pragma unroll
  for (int q = 0; q < NS; ++q) out[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), q));
}

template <int GW, int NS>
__device__ __forceinline__ void r4_allreduce_small(R4Shared& sh, const float* scal, float* out, R4Group& g) {
  r4_allreduce_small_t<GW, NS>(sh, scal, out, g, (int)threadIdx.x);
}

// the same with a compile-time count (the butterflies of the NS scalars interleave)
template <int GW, int NS>
__device__ __forceinline__ void r4_allreduce_scalars4(R4Shared& sh, const float* scal, R4Group& g) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float sv[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) sv[j] = wave_sum_fast(scal[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NS; ++j) sh.red[wave][j] = sv[j];
  }
  r4_group_sum<GW>(sh, NS, g);
}

template <int GW>
__device__ __forceinline__ void r4_allreduce_scalars(R4Shared& sh, const float* scal, int ns, R4Group& g) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int j = 0; j < ns; ++j) {
    const float sv = wave_sum_fast(scal[j]);
    if (lane == 0) sh.red[wave][j] = sv;
  }
  r4_group_sum<GW>(sh, ns, g);
}

}  // namespace lo

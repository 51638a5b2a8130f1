// lo_cg_step_cols.hip -- everything of a streaming CG iteration BEHIND the operator product, for up to 32 columns, in
// ONE launch (round 4).  Reference: linear_cg.py:245-332 (alpha, the residual / solution update, the preconditioner
// call precondition_closure of added_diag_linear_operator.py:135-140, beta, the search direction, the stop rule and the
// tridiagonal recurrence).
//
// The streaming engine of lo_cg.hip spent four launches per iteration on a multi-column solve of a dense operator
// (BASELINE cfg5: 16 probes + the right-hand side): the product, Q^T r with the r / x update (skinny_tn), z = r/d - Q u
// with the r.z partials (skinny_nn) and the one-workgroup control step.  The three small ones move 7 vectors of
// [N, c] and Q once -- 36 MB for four members of 16384 rows -- but took 145 us next to a 760 us product because each is a
// chain of dependent latencies (row pairs per wave, S partial buffers summed by every consumer, one workgroup for the
// control).  Here a member is a GROUP of GW workgroups of 256 rows (64 per wave) that exchange their 16 x c + 2 c partial
// sums ONCE through the tagged granules of lo_group_reduce.h (reduce-scatter + all-gather, all 256 threads fetch):
//
//   alpha_j = r.z_j / p.Ap_j (masked)                 from the product's partials, every workgroup for itself
//   r -= alpha Ap,  x += alpha p                      vectors in the ACCUMULATOR layout of v_mfma_f32_16x16x4_f32:
//   u = Q^T r      (16 x c, matrix cores)             lane (n = l & 15, kk = l >> 4) owns rows 16 rb + 4 kk + i of column n,
//   sum r^2, sum r^2 / d                              which is also the B-operand layout of the reduce product when its
//   ---- one group all-reduce (18 c + 1 values) ----  k-step i covers the rows 4 kk + i: no transposition anywhere
//   r.z = sum r^2/d - |u|^2,  beta = r.z / r.z_old
//   p = (r/d - Q u) + beta p   (matrix cores)         the next iteration's search direction; z itself is never stored
//   control step of the member by the group's first workgroup (beta, residual norm, has_converged, the tridiagonal
//   recurrence); the group that finishes the LAST member takes the batch-global decisions (stop rule :302-308, tridiag
//   freeze :326-327) from three tagged granules per member -- no fence, no second launch.
//
// Without a preconditioner (N below settings.min_preconditioning_size) the same kernel runs with z = r (PRE = false:
// c + 1 values cross the group).  Members of more than 16384 rows (up to 65536) give a workgroup 2 or 4 row blocks: it
// walks over them twice and re-reads r and p behind the exchange.  Groups are persistent and take members dynamically (the next member rides on the
// all-reduce).  Bitwise reproducible: every sum has a fixed order.  A timed-out hand-off sets the error word; the host
// redoes the solve on the multi-launch path.
//   bound: HBM / L2 -- 4 c N 7 bytes of vectors + 4 N 16 of Q (read twice, the second time from cache) per member.
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_group_reduce.h"

namespace lo {

constexpr int SC_TPB = 256;
constexpr int SC_ROWS = 256;              // rows of a member per workgroup: 4 waves x 4 blocks of 16
constexpr int SC_MAXC = 32;               // two column tiles of 16
constexpr int SC_LD = 18 * SC_MAXC + 16;  // granule / LDS row stride (18 c + 1 payload entries at most)

typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

struct ScArgs {
  const float* Q;  // [B, N, ldq] orthonormal factor of the Woodbury preconditioner (ldq = 4, 8 or 16), PRE only
  int ldq;
  const float* dinv;  // [B, N] or [B]
  int dinv_mode;
  float* r;         // [B, N, c] in / out
  const float* Ap;  // [B, N, c]
  float* p;         // in / out: the NEXT iteration's search direction
  float* x;         // in / out
  int c;
  const float* pAp_part;  // [B, S_dot, c]
  int S_dot;
  float* rz;      // [B, c] r.z of the previous iteration (rewritten by the folded control step)
  int* has_conv;  // [B, c]
  float eps;
  float* alpha_out;  // [B, c]
  float* rr_part;    // control step NOT folded: totals into slot 0 of [B, S, c], zeros elsewhere
  float* rz_part;
  int S;
  int64_t B;
  int N;
  unsigned long long* gbuf;
  int* err;
  int* next_member;  // this launch's hand-out counter (zeroed once per solve)
  const int* stop;
  unsigned tag_base;
  int launch;  // iteration index k
  ScCtrl cf;
  long long* dbg;  // LO_SC_DEBUG: phase clocks (100 MHz) of member 0's first workgroup, accumulated over the launches
};

struct alignas(16) ScShared {
  float red[4][SC_LD];
  float res[SC_LD];
  float pa[8][SC_MAXC];
  float alpha[SC_MAXC], beta[SC_MAXC], rzo[SC_MAXC];
};

template <int M, int GW>
__device__ __forceinline__ float sc_seg_step(float v) {
  if constexpr (M < GW) return sc_seg_step<2 * M, GW>(bfly_add<M>(v));
  else return v;
}

// Group all-reduce of sh.red[w][0 .. cnt) -> sh.res[0 .. cnt), cnt <= SC_LD.  Workgroup (e mod GW) owns entry e; all 256
// threads of the owner fetch (256 / GW entries x GW sources per round), the GW lanes of an entry sum with the fixed xor
// butterflies, everybody reads the totals.  Granules: [2 parities][GW + 1][SC_LD] per group.  Ends with a barrier.
template <int GW>
__device__ __forceinline__ void sc_allreduce(ScShared& sh, int cnt, R4Group& g) {
  const int t = threadIdx.x;
  const unsigned tag = ++g.tag;
  __syncthreads();
  if constexpr (GW == 1) {
    for (int e = t; e < cnt; e += SC_TPB) sh.res[e] = ((sh.red[0][e] + sh.red[1][e]) + sh.red[2][e]) + sh.red[3][e];
    __syncthreads();
    return;
  } else {
    unsigned long long* slot = g.gslot + (size_t)(tag & 1u) * (GW + 1) * SC_LD;
    unsigned long long* tot = slot + (size_t)GW * SC_LD;
    for (int e = t; e < cnt; e += SC_TPB)
      pf_store(g, tag, slot + (size_t)g.wig * SC_LD + e, ((sh.red[0][e] + sh.red[1][e]) + sh.red[2][e]) + sh.red[3][e]);
    constexpr int J = SC_TPB / GW;  // entries per round
    const int w = t % GW, jj = t / GW;
    const int nown = (cnt > g.wig) ? (cnt - g.wig + GW - 1) / GW : 0;
    for (int j0 = 0; j0 < nown; j0 += J) {
      const int j = j0 + jj;
      const bool act = j < nown;
      const int e = g.wig + GW * j;
      float v = pf_wait(g, tag, slot + (size_t)w * SC_LD + (act ? e : 0), act);
      v = sc_seg_step<1, GW>(v);
      if (act && w == 0) pf_store(g, tag, tot + e, v);
    }
    for (int e0 = 0; e0 < cnt; e0 += SC_TPB) {
      const int e = e0 + t;
      const float v = pf_wait(g, tag, tot + (e < cnt ? e : 0), e < cnt);
      if (e < cnt) sh.res[e] = v;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float sc_ld(__amdgpu_buffer_rsrc_t rs, int voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
}
__device__ __forceinline__ void sc_st(__amdgpu_buffer_rsrc_t rs, int voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, 0, 0);
}
constexpr int SC_OOB = 0x40000000;  // byte offset beyond every descriptor's range: the load returns 0, the store is dropped

// Addressing: one buffer descriptor per member and array (wave-uniform base, the member's byte count as the range), the
// lane's byte offset in a VGPR, the 16 row steps of a lane as uniform addends.  Rows beyond N and columns beyond c fall
// outside the descriptor's range -- no per-element predicates, no 64-bit address registers.
// RB = row blocks of 256 per workgroup: 1 keeps r, p, 1/d of the workgroup's rows in registers across the exchange; 2 / 4
// (members of up to 65536 rows on groups of at most 64) walk over their blocks twice and re-read r and p behind the exchange
// (cache hits: the first walk has just written them).
template <int GW, int CT, bool PRE, int RB>
__global__ __launch_bounds__(SC_TPB, 2) void k_cg_step_cols(ScArgs a) {
  if (a.stop && *a.stop) return;
  __shared__ ScShared sh;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, kk = lane >> 4;
  R4Group g;
  g.gslot = a.gbuf + (size_t)grp * 2 * (GW + 1) * SC_LD;
  g.wig = wig;
  g.dbg = nullptr;
  g.tag = a.tag_base;
  g.err = a.err;
  g.same_xcd = false;  // (agent-scope granules: this kernel exchanges once per member)
  const int c = a.c;
  const int cnt = PRE ? 18 * c + 1 : c + 1;
  const int rbase = wig * SC_ROWS * RB + 64 * wave;  // first row of the wave (block 0)
  const int k_it = a.launch;
  const int vec_bytes = a.N * c * 4;
  // lane offsets (bytes): vectors at (row rbase + 4 kk, column 16 ct + n); Q as the A operand of the two products
  int voff[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) voff[ct] = (16 * ct + n < c) ? ((rbase + 4 * kk) * c + 16 * ct + n) * 4 : SC_OOB;
  const int voff_d = (rbase + 4 * kk) * 4;
  const int voff_qa = (n < a.ldq) ? ((rbase + 4 * kk) * a.ldq + n) * 4 : SC_OOB;
  const int voff_qb = ((rbase + n) * a.ldq + kk) * 4;

  int b = grp;
  while (b < (int)a.B) {
    b = __builtin_amdgcn_readfirstlane(b);
    const size_t mrow = (size_t)b * a.N;
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(a.r + mrow * c, 0, vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ap =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Ap) + mrow * c, 0, vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.p + mrow * c, 0, vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.x + mrow * c, 0, vec_bytes, 0x00020000);
    const bool stamp = a.dbg && b == 0 && wig == 0 && t == 0;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (stamp) c0 = wall_clock64();
    float rv[CT][16], pv[CT][16], apv[CT][16], xv[CT][16];
    float dv[16], qa[16];
    if constexpr (RB == 1) {
      // ---- loads of the first column tile, the diagonal and Q ----
  #pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int so = (16 * (q >> 2) + (q & 3)) * c * 4;
        rv[0][q] = sc_ld(rs_r, voff[0] + so);
        apv[0][q] = sc_ld(rs_ap, voff[0] + so);
        pv[0][q] = sc_ld(rs_p, voff[0] + so);
        xv[0][q] = sc_ld(rs_x, voff[0] + so);
      }
      if constexpr (PRE) {
        const __amdgpu_buffer_rsrc_t rs_q =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q) + mrow * a.ldq, 0, a.N * a.ldq * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_d =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dinv) + mrow, 0, a.N * 4, 0x00020000);
        const float dconst = (a.dinv_mode == LO_DIAG_FULL) ? 0.f : a.dinv[b];
  #pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ro = 16 * (q >> 2) + (q & 3);
          dv[q] = (a.dinv_mode == LO_DIAG_FULL) ? sc_ld(rs_d, voff_d + ro * 4) : dconst;
          qa[q] = sc_ld(rs_q, voff_qa + ro * a.ldq * 4);
        }
      } else {
  #pragma unroll
        for (int q = 0; q < 16; ++q) dv[q] = 1.f, qa[q] = 0.f;
      }
    }
    // ---- alpha of every column from the product's partials (linear_cg.py:250-260), while the loads are in flight ----
    float rzo_t = 0.f;
    int conv_t = 0;
    if (t < c) {
      rzo_t = a.rz[(size_t)b * c + t];
      conv_t = a.has_conv[(size_t)b * c + t];
    }
    {
      // thread (column t & 31, part t >> 5) sums the partials part, part + 8, ...: eight loads in flight per round trip
      const int col = t & 31, part = t >> 5;
      const float* pp = a.pAp_part + (size_t)b * a.S_dot * c + col;
      float s = 0.f;
      for (int s0 = part; s0 < a.S_dot; s0 += 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ss = s0 + 8 * j;
          v[j] = (col < c && ss < a.S_dot) ? pp[(size_t)ss * c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
      }
      sh.pa[part][col] = s;
    }
    __syncthreads();
    if (t < SC_MAXC) {
      float al = 0.f;
      if (t < c) {
        const float pAp = ((sh.pa[0][t] + sh.pa[1][t]) + (sh.pa[2][t] + sh.pa[3][t])) +
                          ((sh.pa[4][t] + sh.pa[5][t]) + (sh.pa[6][t] + sh.pa[7][t]));
        al = (pAp < a.eps) ? 0.f : rzo_t / pAp;
        if (conv_t) al = 0.f;
        if (wig == 0) a.alpha_out[(size_t)b * c + t] = al;
      }
      sh.alpha[t] = al;
      sh.rzo[t] = rzo_t;
    }
    __syncthreads();
    if (stamp) c1 = wall_clock64();
    if constexpr (RB == 1) {
      if constexpr (CT == 2) {  // the second tile's r / Ap travel while the first tile is updated
  #pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int so = (16 * (q >> 2) + (q & 3)) * c * 4;
          rv[1][q] = sc_ld(rs_r, voff[1] + so);
          apv[1][q] = sc_ld(rs_ap, voff[1] + so);
        }
      }
      // ---- r -= alpha Ap, x += alpha p (:264, :31); the partials of Q^T r, sum r^2, sum r^2 / d ----
  #pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = 16 * ct + n;
        const float al = sh.alpha[col];
        if (ct == 1) {
  #pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int so = (16 * (q >> 2) + (q & 3)) * c * 4;
            pv[ct][q] = sc_ld(rs_p, voff[ct] + so);
            xv[ct][q] = sc_ld(rs_x, voff[ct] + so);
          }
        }
        float s0 = 0.f, s1 = 0.f;
        sc_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int so = (16 * (q >> 2) + (q & 3)) * c * 4;
          const float rn = fmaf(-al, apv[ct][q], rv[ct][q]);
          rv[ct][q] = rn;
          s0 = fmaf(rn, rn, s0);
          if constexpr (PRE) {
            s1 = fmaf(rn * dv[q], rn, s1);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q], rn, acc, 0, 0, 0);
          }
          sc_st(rs_r, voff[ct] + so, rn);
          sc_st(rs_x, voff[ct] + so, fmaf(al, pv[ct][q], xv[ct][q]));
        }
        s0 = bfly_add<32>(bfly_add<16>(s0));
        if constexpr (PRE) {
          s1 = bfly_add<32>(bfly_add<16>(s1));
          if (col < c) {
  #pragma unroll
            for (int e = 0; e < 4; ++e) sh.red[wave][(4 * kk + e) * c + col] = acc[e];
            if (kk == 0) {
              sh.red[wave][16 * c + col] = s0;
              sh.red[wave][17 * c + col] = s1;
            }
          }
        } else {
          if (col < c && kk == 0) sh.red[wave][col] = s0;
        }
      }
    } else {
      // ---- members of more than 256 GW rows: block after block; the partial sums ride in registers across the blocks ----
      sc_f32x4 accb[CT];
      float s0b[CT], s1b[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        accb[ct] = sc_f32x4{0.f, 0.f, 0.f, 0.f};
        s0b[ct] = 0.f;
        s1b[ct] = 0.f;
      }
      for (int blk = 0; blk < RB; ++blk) {
        if constexpr (PRE) {
          const __amdgpu_buffer_rsrc_t rs_q =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q) + mrow * a.ldq, 0, a.N * a.ldq * 4, 0x00020000);
          const __amdgpu_buffer_rsrc_t rs_d =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dinv) + mrow, 0, a.N * 4, 0x00020000);
          const float dconst = (a.dinv_mode == LO_DIAG_FULL) ? 0.f : a.dinv[b];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int ro = SC_ROWS * blk + 16 * (q >> 2) + (q & 3);
            dv[q] = (a.dinv_mode == LO_DIAG_FULL) ? sc_ld(rs_d, voff_d + ro * 4) : dconst;
            qa[q] = sc_ld(rs_q, voff_qa + ro * a.ldq * 4);
          }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float al = sh.alpha[16 * ct + n];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int so = (SC_ROWS * blk + 16 * (q >> 2) + (q & 3)) * c * 4;
            rv[0][q] = sc_ld(rs_r, voff[ct] + so);
            apv[0][q] = sc_ld(rs_ap, voff[ct] + so);
            pv[0][q] = sc_ld(rs_p, voff[ct] + so);
            xv[0][q] = sc_ld(rs_x, voff[ct] + so);
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int so = (SC_ROWS * blk + 16 * (q >> 2) + (q & 3)) * c * 4;
            const float rn = fmaf(-al, apv[0][q], rv[0][q]);
            s0b[ct] = fmaf(rn, rn, s0b[ct]);
            if constexpr (PRE) {
              s1b[ct] = fmaf(rn * dv[q], rn, s1b[ct]);
              accb[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q], rn, accb[ct], 0, 0, 0);
            }
            sc_st(rs_r, voff[ct] + so, rn);
            sc_st(rs_x, voff[ct] + so, fmaf(al, pv[0][q], xv[0][q]));
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = 16 * ct + n;
        const float s0 = bfly_add<32>(bfly_add<16>(s0b[ct]));
        if constexpr (PRE) {
          const float s1 = bfly_add<32>(bfly_add<16>(s1b[ct]));
          if (col < c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sh.red[wave][(4 * kk + e) * c + col] = accb[ct][e];
            if (kk == 0) {
              sh.red[wave][16 * c + col] = s0;
              sh.red[wave][17 * c + col] = s1;
            }
          }
        } else {
          if (col < c && kk == 0) sh.red[wave][col] = s0;
        }
      }
    }
    if (lane == 0)  // the next member rides on the same all-reduce: drawn by the group's first workgroup (exact < 2^24)
      sh.red[wave][cnt - 1] = (wig == 0 && wave == 0) ? (float)(ngroups + atomicAdd(a.next_member, 1)) : 0.f;
    // rows of Q as the A operand of the expanding product (the same lines again: cache hits), issued before the wait
    float qb[16];
    if constexpr (PRE && RB == 1) {
      const __amdgpu_buffer_rsrc_t rs_q =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q) + mrow * a.ldq, 0, a.N * a.ldq * 4, 0x00020000);
#pragma unroll
      for (int q = 0; q < 16; ++q)
        qb[q] = (4 * (q & 3) < a.ldq) ? sc_ld(rs_q, voff_qb + (16 * (q >> 2) * a.ldq + 4 * (q & 3)) * 4) : 0.f;
    }
    if (stamp) c2 = wall_clock64();
    sc_allreduce<GW>(sh, cnt, g);
    if (stamp) c3 = wall_clock64();
    // ---- beta per column; the member's control step ----
    if (t < SC_MAXC) {
      float be = 0.f;
      if (t < c) {
        float srr, srz;
        if constexpr (PRE) {
          float uu = 0.f;
#pragma unroll
          for (int m = 0; m < 16; ++m) uu = fmaf(sh.res[m * c + t], sh.res[m * c + t], uu);
          srr = sh.res[16 * c + t];
          srz = sh.res[17 * c + t] - uu;  // r.z = sum r^2/d - |Q^T r|^2
        } else {
          srr = sh.res[t];
          srz = srr;  // z = r
        }
        const float rzo = sh.rzo[t];
        be = (rzo < a.eps) ? 0.f : srz / rzo;  // :39-42
        if (wig == 0) {
          if (a.cf.on) {
            // cg_scal_body of lo_cg.hip for this member; rz / has_conv of the member were read by every workgroup of
            // the group before the exchange completed
            const size_t i = (size_t)b * c + t;
            float rn = sqrtf(srr);  // :298
            if (a.cf.rhs_is_zero[i]) rn = 0.f;
            a.rz[i] = srz;
            a.cf.beta[i] = be;
            a.cf.resid_norm[i] = rn;
            a.has_conv[i] = rn < a.cf.stop_after;  // :300
            float lmax = -INFINITY;
            const bool tri = a.cf.n_tridiag && k_it < a.cf.n_tridiag_iter && !a.cf.ctrl->tri_disabled;
            if (tri && t < a.cf.n_tridiag) {
              const size_t it = (size_t)b * a.cf.n_tridiag + t;
              const float al = sh.alpha[t];
              const float ar = 1.0f / ((al == 0.f) ? 1.0f : al);  // :314-317
              const int T = a.cf.T;
              float* tm = a.cf.t_mat + ((size_t)t * a.B + b) * T * T;
              if (k_it == 0) {
                tm[0] = ar;  // :320
              } else {
                const float pb = a.cf.prev_beta[it], par = a.cf.prev_ar[it];
                tm[k_it * T + k_it] = fmaf(pb, par, ar);  // :322
                const float off = sqrtf(pb) * par;        // :323
                tm[k_it * T + k_it - 1] = off;
                tm[(k_it - 1) * T + k_it] = off;  // :324
                lmax = off;
              }
              a.cf.prev_ar[it] = ar;  // :331-332
              a.cf.prev_beta[it] = be;
            }
            sh.pa[0][t] = rn;
            sh.pa[1][t] = (srr != srr || srz != srz) ? 1.f : 0.f;
            sh.pa[2][t] = lmax;
          } else {
            for (int s = 0; s < a.S; ++s) {
              a.rr_part[((size_t)b * a.S + s) * c + t] = (s == 0) ? srr : 0.f;
              a.rz_part[((size_t)b * a.S + s) * c + t] = (s == 0) ? srz : 0.f;
            }
          }
        }
      }
      sh.beta[t] = be;
    }
    const int bnext = (int)sh.res[cnt - 1];
    __syncthreads();
    int cf_old = -1;
    if (a.cf.on && wig == 0 && t == 0) {
      float ls = 0.f, ln = 0.f, lm = -INFINITY;
      for (int j = 0; j < c; ++j) {  // fixed order
        ls += sh.pa[0][j];
        ln += sh.pa[1][j];
        lm = fmaxf(lm, sh.pa[2][j]);
      }
      const unsigned long long tg = (unsigned long long)(unsigned)(k_it + 1) << 32;
      __hip_atomic_store(a.cf.gran + b, tg | __float_as_uint(ls), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.cf.gran + a.B + b, tg | __float_as_uint(ln), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.cf.gran + 2 * a.B + b, tg | __float_as_uint(lm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cf_old = atomicAdd(a.cf.done, 1);
    }
    if (stamp) c4 = wall_clock64();
    // ---- p <- z + beta p with z = r/d - Q u (:140, :46) ----
    if constexpr (RB == 1) {
  #pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = 16 * ct + n;
        const float be = sh.beta[col];
        float ub[4];
        if constexpr (PRE) {
  #pragma unroll
          for (int s = 0; s < 4; ++s) ub[s] = (col < c) ? sh.res[(4 * s + kk) * c + col] : 0.f;
        }
  #pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          sc_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRE) {
  #pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[4 * rb + s], ub[s], acc, 0, 0, 0);
          }
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * rb + e;
            const float z = PRE ? (rv[ct][q] * dv[q] - acc[e]) : rv[ct][q];
            sc_st(rs_p, voff[ct] + (16 * rb + e) * c * 4, fmaf(pv[ct][q], be, z));
          }
        }
      }
    } else {
      for (int blk = 0; blk < RB; ++blk) {
        if constexpr (PRE) {
          const __amdgpu_buffer_rsrc_t rs_q =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Q) + mrow * a.ldq, 0, a.N * a.ldq * 4, 0x00020000);
          const __amdgpu_buffer_rsrc_t rs_d =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dinv) + mrow, 0, a.N * 4, 0x00020000);
          const float dconst = (a.dinv_mode == LO_DIAG_FULL) ? 0.f : a.dinv[b];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            dv[q] = (a.dinv_mode == LO_DIAG_FULL) ? sc_ld(rs_d, voff_d + (SC_ROWS * blk + 16 * (q >> 2) + (q & 3)) * 4) : dconst;
            qb[q] = (4 * (q & 3) < a.ldq)
                        ? sc_ld(rs_q, voff_qb + ((SC_ROWS * blk + 16 * (q >> 2)) * a.ldq + 4 * (q & 3)) * 4) : 0.f;
          }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int col = 16 * ct + n;
          const float be = sh.beta[col];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int so = (SC_ROWS * blk + 16 * (q >> 2) + (q & 3)) * c * 4;
            rv[0][q] = sc_ld(rs_r, voff[ct] + so);
            pv[0][q] = sc_ld(rs_p, voff[ct] + so);
          }
          float ub[4];
          if constexpr (PRE) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ub[s] = (col < c) ? sh.res[(4 * s + kk) * c + col] : 0.f;
          }
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            sc_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PRE) {
#pragma unroll
              for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[4 * rb + s], ub[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 4 * rb + e;
              const float z = PRE ? (rv[0][q] * dv[q] - acc[e]) : rv[0][q];
              sc_st(rs_p, voff[ct] + (SC_ROWS * blk + 16 * rb + e) * c * 4, fmaf(pv[0][q], be, z));
            }
          }
        }
      }
    }
    if (a.cf.on && wig == 0 && t < 64) {
      const int last = __builtin_amdgcn_readfirstlane((cf_old == (int)a.B - 1) ? 1 : 0);
      if (last) {  // every member of the batch is recorded: the batch-global decisions (cg_ctrl_body), fixed order
        float ls = 0.f, ln = 0.f, lm = -INFINITY;
        const unsigned want = (unsigned)(k_it + 1);
        for (int64_t i = t; i < 3 * a.B; i += 64) {
          unsigned long long gq;
          unsigned spin = 0;
          do {  // (the counter said every member was issued; its granules may still be on their way)
            gq = __hip_atomic_load(a.cf.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned)(gq >> 32) != want && ++spin < R4_MAXSPIN);
          if ((unsigned)(gq >> 32) != want) atomicExch(a.err, 1);
        }
        for (int64_t i = t; i < a.B; i += 64) {
          ls += __uint_as_float((unsigned)(__hip_atomic_load(a.cf.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull));
          ln += __uint_as_float((unsigned)(__hip_atomic_load(a.cf.gran + a.B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull));
          lm = fmaxf(lm, __uint_as_float((unsigned)(__hip_atomic_load(a.cf.gran + 2 * a.B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull)));
        }
        const float mean = wave_sum_fast(ls) / (float)(a.B * c);
        const float anynan = wave_sum_fast(ln);
        const float mx = wave_max(lm);
        if (t == 0) {
          const int k = k_it;
          CgCtrl* ctrl = a.cf.ctrl;
          const int kfloor = min(10, a.cf.max_iter - 1);
          const bool stopnow = (k >= kfloor) && (mean < a.cf.tol) &&
                               !(a.cf.n_tridiag && k < min(a.cf.n_tridiag_iter, a.cf.max_iter - 1));  // :302-306
          ctrl->iterations = k + 1;
          ctrl->mean_resid = mean;
          if (k == 0 && a.cf.check_nan_first && anynan > 0.f) {  // NaN matvec detected on the first product
            ctrl->nan_detected = 1;
            ctrl->stop = 1;
          }
          if (stopnow) {
            ctrl->tol_reached = 1;  // :307
            ctrl->stop = 1;
          } else if (a.cf.n_tridiag && k < a.cf.n_tridiag_iter && !ctrl->tri_disabled) {
            if (k > 0 && mx < 1e-6f) ctrl->tri_disabled = 1;  // :326-327
            ctrl->last_tridiag_iter = k;                       // :329
          }
        }
      }
    }
    if (stamp) {
      c5 = wall_clock64();
      a.dbg[0] += c1 - c0; a.dbg[1] += c2 - c1; a.dbg[2] += c3 - c2; a.dbg[3] += c4 - c3; a.dbg[4] += c5 - c4; a.dbg[5] += 1;
    }
    __syncthreads();  // (sh.res / sh.alpha / sh.beta / sh.pa are reused by the next member)
    b = bnext;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int sc_row_blocks(int64_t N) { return N <= 64 * (int64_t)SC_ROWS ? 1 : (N <= 128 * (int64_t)SC_ROWS ? 2 : 4); }
int cg_step_cols_group(int64_t N) {  // workgroups per member: the smallest power of two that holds its rows
  const int64_t rows = (int64_t)SC_ROWS * sc_row_blocks(N);
  int gw = 1;
  while ((int64_t)gw * rows < N) gw <<= 1;
  return gw;
}

bool cg_step_cols_eligible(int64_t B, int64_t N, int64_t c, int ldq) {
  return c >= 1 && c <= SC_MAXC && N >= 1 && N <= 256 * (int64_t)SC_ROWS && (ldq == 0 || ldq == 4 || ldq == 8 || ldq == 16) &&
         B >= 1 && B < (1 << 24) - 4096 && !getenv("LO_NO_STEP_COLS");
}

// Where the one launch pays (tools/check_step_cols.py): a member's step is a chain of latencies (loads, exchange, stores:
// 25 - 35 us) that a group runs for one member after the other, while the multi-launch kernels are bandwidth-bound.  So
// the batch may take at most four rounds of the resident groups; thousands of tiny members (N < 512: 1000 x 300 rows
// 4.7 vs 4.2 ms) and few-column unpreconditioned solves (16 lanes per load instruction carry c < 4 values) stay on the
// multi-launch path unless every member has a group of its own.
bool cg_step_cols_worthwhile(int64_t B, int64_t N, int64_t c, bool has_pre) {
  const int64_t slots = B * cg_step_cols_group(N);
  if (slots <= 512) return true;
  return slots <= 4 * 512 && N >= 512 && (has_pre || c >= 4);
}

size_t cg_step_cols_gbuf_bytes() {  // 512 workgroup slots + up to 512 total rows, two parities
  return (size_t)2 * (512 + 512) * SC_LD * sizeof(unsigned long long) + 256;
}

template <int GW, int CT, bool PRE, int RB>
static int sc_go(ScArgs& a, int ncu, hipStream_t st) {
  int per_cu = 0;
  const hipError_t oe = LO_OCCUPANCY_CACHED(per_cu, (k_cg_step_cols<GW, CT, PRE, RB>), SC_TPB, 0);
  if (oe != hipSuccess || per_cu < 1) return LO_ERR_UNSUPPORTED;
  per_cu = std::min(per_cu, 2);
  const int nwg = std::min(512, per_cu * ncu);
  if ((nwg / 8) < GW) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("cg_step_cols", st);
  ResidentLaunch guard(st);
  hipLaunchKernelGGL((k_cg_step_cols<GW, CT, PRE, RB>), dim3(nwg), dim3(SC_TPB), 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int CT, bool PRE>
static int sc_go_gw(int GW, int RB, ScArgs& a, int ncu, hipStream_t st) {
  if (RB == 2) return sc_go<64, CT, PRE, 2>(a, ncu, st);  // (more than 16384 rows: always groups of 64)
  if (RB == 4) return sc_go<64, CT, PRE, 4>(a, ncu, st);
  switch (GW) {
    case 1: return sc_go<1, CT, PRE, 1>(a, ncu, st);
    case 2: return sc_go<2, CT, PRE, 1>(a, ncu, st);
    case 4: return sc_go<4, CT, PRE, 1>(a, ncu, st);
    case 8: return sc_go<8, CT, PRE, 1>(a, ncu, st);
    case 16: return sc_go<16, CT, PRE, 1>(a, ncu, st);
    case 32: return sc_go<32, CT, PRE, 1>(a, ncu, st);
    default: return sc_go<64, CT, PRE, 1>(a, ncu, st);
  }
}

// gbuf: cg_step_cols_gbuf_bytes() zeroed once per solve; next_member: base of one zeroed int per launch; `launch` = the
// iteration index (tags start at launch * (B + 2)).  Q == nullptr: no preconditioner (z = r).
int cg_step_cols(const float* Q, int ldq, const float* dinv, int dinv_mode, float* r, const float* Ap, float* p, float* x,
                 int64_t c, const float* pAp_part, int S_dot, float* rz, int* has_conv, float eps, float* alpha_out,
                 float* rr_part, float* rz_part, int S, int64_t B, int64_t N, unsigned long long* gbuf, int* err,
                 int* next_member, int launch, int max_launch, const int* stop, int ncu, const ScCtrl* cf,
                 long long* dbg, hipStream_t st) {
  if (!cg_step_cols_eligible(B, N, c, Q ? ldq : 0)) return LO_ERR_UNSUPPORTED;
  if ((unsigned long long)(max_launch + 1) * (unsigned long long)(B + 2) >= 0xffff0000ull) return LO_ERR_UNSUPPORTED;
  ScArgs a;
  a.Q = Q; a.ldq = ldq; a.dinv = dinv; a.dinv_mode = dinv_mode; a.r = r; a.Ap = Ap; a.p = p; a.x = x; a.c = (int)c;
  a.pAp_part = pAp_part; a.S_dot = S_dot; a.rz = rz; a.has_conv = has_conv; a.eps = eps; a.alpha_out = alpha_out;
  a.rr_part = rr_part; a.rz_part = rz_part; a.S = S; a.B = B; a.N = (int)N; a.gbuf = gbuf; a.err = err;
  a.next_member = next_member + launch; a.stop = stop;
  a.tag_base = (unsigned)((unsigned long long)launch * (unsigned long long)(B + 2));
  a.launch = launch;
  a.dbg = dbg;
  if (cf) {
    a.cf = *cf;
    if (a.cf.on) a.cf.done += launch;
  } else {
    a.cf.on = 0;
  }
  const int GW = cg_step_cols_group(N), RB = sc_row_blocks(N);
  const bool two = c > 16;
  if (Q) return two ? sc_go_gw<2, true>(GW, RB, a, ncu, st) : sc_go_gw<1, true>(GW, RB, a, ncu, st);
  return two ? sc_go_gw<2, false>(GW, RB, a, ncu, st) : sc_go_gw<1, false>(GW, RB, a, ncu, st);
}

}  // namespace lo

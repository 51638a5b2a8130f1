// lo_precond_fused.hip -- one pass over the Woodbury preconditioner's Q per CG iteration (single right-hand-side column,
// large N): the residual / solution update of linear_cg.py:264,:31 and z = r o dinv - Q (Q^T r) (precondition_closure,
// added_diag_linear_operator.py:135-140) in ONE kernel.
//
// The streaming engine applies the preconditioner in two launches -- Q^T r over the rows (skinny_tn, with the vector
// updates fused), then z = r/d - Q u (skinny_nn) -- and streams Q twice.  For BASELINE cfg4 (Kronecker operator,
// N = 65536, rank-15 preconditioner) Q is 537 MB for a 128-member shard and these two passes are 70 % of a CG iteration.
// Here a member is a group of GW workgroups (256 threads x 4 rows); a thread keeps its four rows of Q (64 values) in
// registers between the two halves and the group exchanges the 16 + 2 partial sums through the tagged granules of the
// operator-resident kernels (lo_group_reduce.h): Q is read from HBM ONCE per iteration.  Groups are persistent, members
// are handed out dynamically, four workgroups share a CU so that one group's hand-off wait is filled by another's
// loads, and the r / x stores sit between the publication of the partial sums and the wait for the totals.  Results
// are bitwise reproducible (fixed summation order); a timed-out hand-off sets the error word and the host redoes the
// solve with the two-launch path.
// The kernel also knows the new r.z after the exchange, hence beta: it writes the next search direction
// p <- z + beta p instead of z, and the separate p-update pass of the streaming loop disappears.
//   bound: HBM -- 4 N (16 + 8) bytes per member and iteration (Q once; r, Ap, p, x, dinv in; r, x, p out).
//
// k_precond_fused_kron (round 3): the same iteration step for the KRONECKER ROOT FORM (lo_precond_desc.kron_*).  The
// preconditioner of K1 (x) K2 + sigma I is P = KP (KP[pivots, :])^-1 KP^T + sigma I, where column m of KP is the
// Kronecker product of pivot row a_m of K1 and pivot row b_m of K2, so
//   z = r / sigma - KP F KP^T r / sigma^2,   F = (KP[pivots, :] + KP^T KP / sigma)^-1   (16 x 16),
//   (KP^T v)_m = sum_i1 a_m[i1] sum_i2 b_m[i2] v[i1, i2],   (KP u)[i1, i2] = sum_m a_m[i1] (b_m[i2] u_m):
// a thread owns rows with ONE second-factor index i2 (its 16 values b_.[i2] sit in registers) and takes the 16 values
// a_.[i1] of each of its four rows from a 64-byte line most lanes of the wave share.  Nothing tall is read besides the
// CG vectors: 4 N 7 bytes per member and iteration instead of 4 N 23.
#include <algorithm>
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_group_reduce.h"

namespace lo {

struct PfArgs {
  const float* Q;         // [B, N, 16]
  const float* dinv;      // [B, N] or [B]
  int dinv_mode;
  float* r;               // [B, N] in / out
  const float* Ap;
  float* p;               // in / out: the kernel also takes the NEXT iteration's p = z + beta p (linear_cg.py:34-46)
  float* x;               // in / out
  float* z;               // not written: z only feeds r.z and the p update, both done here
  const float* pAp_part;  // [B, S_dot]
  int S_dot;
  const float* rz;        // [B]
  const int* has_conv;    // [B]
  float eps;
  float* alpha_out;       // [B]
  float* rr_part;         // [B, S]: slot 0 <- sum r^2, the others <- 0
  float* rz_part;         // [B, S]: slot 0 <- r.z
  int S;
  int64_t B;
  int N, RW;
  unsigned long long* gbuf;
  int* err;
  int* next_member;
  const int* stop;
  int allow_l2_handoff;
  unsigned tag_base;  // tags of this launch start above it: the granules of earlier launches of the solve never match
  const int* iter_ptr;  // replayed as a graph node: the launch index is read from the control block (else nullptr)
  // the control step of the iteration (k_cg_scal + k_cg_ctrl of lo_cg.hip) folded into this launch: the group that
  // finishes a member records its scalars, the group that finishes the LAST member takes the batch-global decision
  PfCtrl cf;
  int launch;  // iteration index k of this launch (iter_ptr: read on the device)
  // Kronecker root form (k_precond_fused_kron): kron_a [B, n1, 16], kron_b [B, n2, 16], kron_F [B, 16, 16]; Q unused
  const float* ka;
  const float* kb;
  const float* kF;
  int n1, n2;
  float inv_n2;
};

// (pf_publish / pf_collect: the lane-parallel reduce-scatter all-reduce for groups of up to 64 workgroups, lo_group_reduce.h)
// value of quad lane K in all four lanes of the quad (DPP quad_perm)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf,
                                                    false));
}
__device__ __forceinline__ float quad_pick(float v, int k) {
  return k == 0 ? quad_bcast<0>(v) : (k == 1 ? quad_bcast<1>(v) : (k == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
}

template <int GW, int OCC>
__global__ __launch_bounds__(R4_TPB, OCC) void k_precond_fused(PfArgs a) {
  if (a.stop && *a.stop) return;
  __shared__ R4Shared sh;
  __shared__ float alpha_s, rzo_s;
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x;
  R4Group g;
  g.gslot = a.gbuf + (size_t)grp * 2 * (GW + 1) * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  if (a.iter_ptr) {  // (graph replay: same arguments every time, the iteration index lives on the device)
    const int launch = *a.iter_ptr;
    a.tag_base = (unsigned)launch * (unsigned)(a.B + 2);
    a.next_member += launch;
    a.launch = launch;
    if (a.cf.on) a.cf.done += launch;
  }
  g.tag = a.tag_base;
  g.err = a.err;
  g.same_xcd = false;
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t < 64) {
      sh.red[0][0] = (float)xcc;
      sh.red[0][1] = (float)(xcc * xcc);
    }
    if (t < 2 * (R4_WAVES - 1)) sh.red[1 + t / 2][t % 2] = 0.f;
    const PfSlots ps = pf_publish<GW>(sh, 2, g);
    pf_collect<GW>(sh, 2, g, ps);
    const float fx = (float)xcc;
    g.same_xcd = (sh.res[0] == GW * fx) && (sh.res[1] == GW * fx * fx) && (a.allow_l2_handoff != 0);
    __syncthreads();
  }
  const int row0 = wig * a.RW;
  const int nv = max(0, min(a.RW, a.N - row0));
  // Ownership of Q: thread t holds the 16-byte piece (row = 64 i + (t >> 2), columns 4 (t & 3) .. + 3) for i = 0 .. 15:
  // consecutive lanes read consecutive pieces (fully coalesced 1 KiB per wave instruction), the four lanes of a quad
  // share a row.  Ownership of the vectors: of the 16 rows of a quad, lane qd has the four with i % 4 == qd; the others
  // get r through a DPP quad broadcast.
  constexpr int NP = R4_ROWS * 4 / R4_TPB;  // 16 pieces per thread
  constexpr int NV = NP / 4;                // 4 vector rows per thread
  const int qd = t & 3, rsub = t >> 2;
  int64_t b = grp;
  while (b < a.B) {
    float4 Qp[NP];
    float rv[NV], apv[NV], dv[NV];
    const size_t mb = (size_t)b * a.N + row0;
    const float4* qsrc = reinterpret_cast<const float4*>(a.Q + mb * 16);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int lr = 64 * i + rsub;
      Qp[i] = (lr < nv) ? qsrc[i * R4_TPB + t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int lr = 64 * (4 * j + qd) + rsub;
      const bool ok = lr < nv;
      rv[j] = ok ? a.r[mb + lr] : 0.f;
      apv[j] = ok ? a.Ap[mb + lr] : 0.f;
      dv[j] = ok ? ((a.dinv_mode == LO_DIAG_FULL) ? a.dinv[mb + lr] : a.dinv[b]) : 0.f;
    }
    if (t < 64) {  // masked alpha from the matvec's p.Ap partials (linear_cg.py:250-260), while the loads are in flight
      float pAp = 0.f;
      for (int s = t; s < a.S_dot; s += 64) pAp += a.pAp_part[(size_t)b * a.S_dot + s];
      pAp = wave_sum_fast(pAp);
      const float rzo = a.rz[b];
      float al = (pAp < a.eps) ? 0.f : rzo / pAp;
      if (a.has_conv[b]) al = 0.f;
      if (t == 0) {
        rzo_s = rzo;
        alpha_s = al;
        if (wig == 0) a.alpha_out[b] = al;
      }
    }
    __syncthreads();
    const float al = alpha_s;
    float sc0 = 0.f, sc1 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      rv[j] = fmaf(-al, apv[j], rv[j]);  // r -= alpha Ap     :264
      dv[j] *= rv[j];                    // r / d
      sc0 = fmaf(rv[j], rv[j], sc0);
      sc1 = fmaf(dv[j], rv[j], sc1);
    }
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);  // partial of (Q^T r)[4 qd .. 4 qd + 3]
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float ri = quad_pick(rv[i >> 2], i & 3);
      u.x = fmaf(Qp[i].x, ri, u.x);
      u.y = fmaf(Qp[i].y, ri, u.y);
      u.z = fmaf(Qp[i].z, ri, u.z);
      u.w = fmaf(Qp[i].w, ri, u.w);
    }
    // wave: sum over the lanes of equal quad index (lane bits 2..5), then the two scalars over all lanes
    {
      float v[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xx = v[e];
        xx = bfly_add<4>(xx); xx = bfly_add<8>(xx); xx = bfly_add<16>(xx); xx = bfly_add<32>(xx);
        v[e] = xx;
      }
      const int lane = t & 63, wave = t >> 6;
      if (lane < 4) {
        sh.red[wave][4 * lane] = v[0]; sh.red[wave][4 * lane + 1] = v[1];
        sh.red[wave][4 * lane + 2] = v[2]; sh.red[wave][4 * lane + 3] = v[3];
      }
      const float s0 = wave_sum_fast(sc0), s1 = wave_sum_fast(sc1);
      if (lane == 0) {
        sh.red[wave][16] = s0;
        sh.red[wave][17] = s1;
        // the next member rides on the same all-reduce: drawn by the group's first workgroup (exact below 2^24)
        sh.red[wave][18] = (wig == 0 && wave == 0) ? (float)(ngroups + atomicAdd(a.next_member, 1)) : 0.f;
      }
    }
    const PfSlots ps = pf_publish<GW>(sh, 19, g);
    // while the partial sums travel: store r, update x
    float pv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int lr = 64 * (4 * j + qd) + rsub;
      pv[j] = 0.f;
      if (lr < nv) {
        pv[j] = a.p[mb + lr];
        a.r[mb + lr] = rv[j];
        a.x[mb + lr] = fmaf(al, pv[j], a.x[mb + lr]);  // x += alpha p      :31
      }
    }
    pf_collect<GW>(sh, 19, g, ps);
    const float4 ut = *reinterpret_cast<const float4*>(&sh.res[4 * qd]);
    float uu = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) uu = fmaf(sh.res[j], sh.res[j], uu);
    const float srr = sh.res[16], srz = sh.res[17] - uu;  // r.z = sum r^2/d - |Q^T r|^2
    const float rzo = rzo_s;
    const float beta = (rzo < a.eps) ? 0.f : srz / rzo;  // the control step's rule (:39-42) on the same numbers
    const int64_t bnext = (int64_t)sh.res[18];
    int cf_old = -1;
    if (a.cf.on && wig == 0 && t == 0) {
      // control step of this member (cg_scal_body of lo_cg.hip for one column, no tridiagonal): beta (:39-42), residual
      // norm (:298-299), has_converged (:300).  rz / has_conv of the member were read by every workgroup of the group
      // before the exchange above completed.  Issued HERE, before the p stores: the fence only waits for the r / x
      // stores (sent before the exchange), and the counter's answer is not needed until the p stores are out.
      float rn = sqrtf(srr);
      if (a.cf.rhs_is_zero[b]) rn = 0.f;
      a.cf.rz[b] = srz;
      a.cf.beta[b] = beta;
      a.cf.resid_norm[b] = rn;
      a.cf.has_conv[b] = rn < a.cf.stop_after;
      // No fence (a device-scope release writes the XCD's L2 back: +13 us per launch): the norm travels to the group
      // that closes the launch inside a tagged 8-byte granule, like the hand-offs; the plain stores above are for the
      // NEXT launch.
      __hip_atomic_store(a.cf.gran + b, ((unsigned long long)(unsigned)(a.launch + 1) << 32) | __float_as_uint(rn),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cf_old = atomicAdd(a.cf.done, 1);
    }
    float zv[NV];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float part = Qp[i].x * ut.x;
      part = fmaf(Qp[i].y, ut.y, part);
      part = fmaf(Qp[i].z, ut.z, part);
      part = fmaf(Qp[i].w, ut.w, part);
      part = bfly_add<1>(part);
      part = bfly_add<2>(part);  // (Q u)[row] in all four lanes of the row
      if ((i & 3) == 0) zv[i >> 2] = part;
      else if (qd == (i & 3)) zv[i >> 2] = part;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int lr = 64 * (4 * j + qd) + rsub;
      // z = r/d - Q (Q^T r) (:140) goes straight into p <- z + beta p (:46)
      if (lr < nv) a.p[mb + lr] = fmaf(pv[j], beta, dv[j] - zv[j]);
    }
    if (wig == 0 && t < a.S) {  // the control step sums S partials per member: the totals go to slot 0
      a.rr_part[(size_t)b * a.S + t] = (t == 0) ? srr : 0.f;
      a.rz_part[(size_t)b * a.S + t] = (t == 0) ? srz : 0.f;
    }
    if (a.cf.on && wig == 0 && t < 64) {
      const int last = __builtin_amdgcn_readfirstlane((cf_old == (int)a.B - 1) ? 1 : 0);
      if (last) {  // every member of the batch is recorded: the batch-global decisions (cg_ctrl_body), fixed order
        float ls = 0.f;
        const unsigned want = (unsigned)(a.launch + 1);
        for (int64_t i = t; i < a.B; i += 64) {
          unsigned long long gq;
          unsigned spin = 0;
          do {  // (the counter said every member was issued; its granule may still be on its way)
            gq = __hip_atomic_load(a.cf.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned)(gq >> 32) != want && ++spin < R4_MAXSPIN);
          if ((unsigned)(gq >> 32) != want) atomicExch(a.err, 1);
          ls += __uint_as_float((unsigned)(gq & 0xffffffffull));
        }
        const float mean = wave_sum_fast(ls) / (float)a.B;
        if (t == 0) {
          const int k = a.launch;
          a.cf.ctrl->iterations = k + 1;
          a.cf.ctrl->mean_resid = mean;
          if (k >= a.cf.kfloor && mean < a.cf.tol) {  // :302-306 (no tridiagonal columns on this path)
            a.cf.ctrl->tol_reached = 1;
            a.cf.ctrl->stop = 1;
          }
        }
      }
    }
    __syncthreads();  // (sh.res / alpha_s / rzo_s are reused by the next member)
    b = bnext;
  }
}

// read-only data behind a wave-uniform address: the constant address space lets the compiler use the scalar cache
// (a plain global pointer may alias the kernel's stores, which rules scalar loads out)
typedef const float __attribute__((address_space(4)))* cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_const_space(const float* p) { return (cfloat_ptr)(uintptr_t)p; }

// Kronecker root form (see the header of this file).  Same group structure, member hand-out, control step and tags as
// k_precond_fused.  Thread t owns the rows row0 + t + 256 q (q < NRK) of its workgroup: n2 divides 256, so they share
// the second-factor index i2 (16 values b_.[i2] in registers), and n2 >= 64 with row0 a multiple of 256 makes the
// first-factor index i1 uniform over a wave: the 16 values a_.[i1] of a row come through the scalar cache into SGPRs.
// NRK = 4 or 8 rows per thread (8: a member of 65536 rows is a group of 32 workgroups instead of 64 -- half the
// hand-offs for the same bytes).
template <int GW, int OCC, int NRK>
__global__ __launch_bounds__(R4_TPB, OCC) void k_precond_fused_kron(PfArgs a) {
  if (a.stop && *a.stop) return;
  __shared__ R4Shared sh;
  __shared__ float alpha_s, rzo_s;
  __shared__ __attribute__((aligned(16))) float kf_s[256];
  const int wg = blockIdx.x;
  const int xcd = wg % 8, jx = wg / 8;
  const int groups_per_xcd = (gridDim.x / 8) / GW;
  const int grp = xcd * groups_per_xcd + jx / GW;
  const int wig = jx % GW;
  const int ngroups = groups_per_xcd * 8;
  if (jx / GW >= groups_per_xcd) return;
  const int t = threadIdx.x;
  R4Group g;
  g.gslot = a.gbuf + (size_t)grp * 2 * (GW + 1) * R4_SLOT;
  g.wig = wig;
  g.dbg = nullptr;
  if (a.iter_ptr) {
    const int launch = *a.iter_ptr;
    a.tag_base = (unsigned)launch * (unsigned)(a.B + 2);
    a.next_member += launch;
    a.launch = launch;
    if (a.cf.on) a.cf.done += launch;
  }
  g.tag = a.tag_base;
  g.err = a.err;
  g.same_xcd = false;
  {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (t < 64) {
      sh.red[0][0] = (float)xcc;
      sh.red[0][1] = (float)(xcc * xcc);
    }
    if (t < 2 * (R4_WAVES - 1)) sh.red[1 + t / 2][t % 2] = 0.f;
    const PfSlots ps = pf_publish<GW>(sh, 2, g);
    pf_collect<GW>(sh, 2, g, ps);
    const float fx = (float)xcc;
    g.same_xcd = (sh.res[0] == GW * fx) && (sh.res[1] == GW * fx * fx) && (a.allow_l2_handoff != 0);
    __syncthreads();
  }
  const int row0 = wig * a.RW;  // multiple of 256 (precond_fused_rupdate)
  const int nv = max(0, min(a.RW, a.N - row0));
  const int lane = t & 63, wave = t >> 6;
  // first-factor index of the wave's rows at q = 0 and its step per q; the second-factor index of the thread
  const int nw0 = row0 + 64 * wave;  // (n2 >= 64 divides 256: the wave's 64 rows share i1)
  const int i1w = __builtin_amdgcn_readfirstlane(min(nw0, a.N - 1) / a.n2);
  const int i1step = R4_TPB / a.n2;
  const int i2 = (nw0 + lane) % a.n2;
  const int i1max = a.n1 - 1;
  int64_t b = grp;
  while (b < a.B) {
    float rv[NRK], apv[NRK];
    const size_t mb = (size_t)b * a.N + row0;
    const float* kab = a.ka + (size_t)b * a.n1 * 16;
    const float4* kb4 = reinterpret_cast<const float4*>(a.kb) + ((size_t)b * a.n2 + i2) * 4;
    float bm[16];
    {
      const float4 b0 = kb4[0], b1 = kb4[1], b2 = kb4[2], b3 = kb4[3];
      bm[0] = b0.x; bm[1] = b0.y; bm[2] = b0.z; bm[3] = b0.w; bm[4] = b1.x; bm[5] = b1.y; bm[6] = b1.z; bm[7] = b1.w;
      bm[8] = b2.x; bm[9] = b2.y; bm[10] = b2.z; bm[11] = b2.w; bm[12] = b3.x; bm[13] = b3.y; bm[14] = b3.z; bm[15] = b3.w;
    }
    kf_s[t] = a.kF[(size_t)b * 256 + t];  // (R4_TPB == 256 entries; the barrier behind the alpha block covers it)
    const float dc = a.dinv[b];
#pragma unroll
    for (int q = 0; q < NRK; ++q) {
      const int lr = t + R4_TPB * q;
      const bool ok = lr < nv;
      rv[q] = ok ? a.r[mb + lr] : 0.f;
      apv[q] = ok ? a.Ap[mb + lr] : 0.f;
    }
    if (t < 64) {  // masked alpha from the matvec's p.Ap partials (linear_cg.py:250-260), while the loads are in flight
      float pAp = 0.f;
      for (int s = t; s < a.S_dot; s += 64) pAp += a.pAp_part[(size_t)b * a.S_dot + s];
      pAp = wave_sum_fast(pAp);
      const float rzo = a.rz[b];
      float al = (pAp < a.eps) ? 0.f : rzo / pAp;
      if (a.has_conv[b]) al = 0.f;
      if (t == 0) {
        rzo_s = rzo;
        alpha_s = al;
        if (wig == 0) a.alpha_out[b] = al;
      }
    }
    __syncthreads();
    const float al = alpha_s;
    float sc0 = 0.f;
    float wm[16];  // sum over the thread's rows of a_m[i1] r[i1, i2]; times b_m[i2] / sigma below
#pragma unroll
    for (int m = 0; m < 16; ++m) wm[m] = 0.f;
#pragma unroll
    for (int q = 0; q < NRK; ++q) {
      rv[q] = fmaf(-al, apv[q], rv[q]);  // r -= alpha Ap     :264
      sc0 = fmaf(rv[q], rv[q], sc0);
      const cfloat_ptr ar = as_const_space(kab + (size_t)min(i1w + q * i1step, i1max) * 16);  // wave-uniform: s_load
#pragma unroll
      for (int m = 0; m < 16; ++m) wm[m] = fmaf(ar[m], rv[q], wm[m]);
    }
    {
      // lane l: wave sum of component l >> 2 of w = KP^T (r / sigma)
      const float mine = r4_wave_rs_t<16>([&](int c) { return wm[c] * (bm[c] * dc); }, lane);
      if ((lane & 3) == 0) sh.red[wave][lane >> 2] = mine;
      const float s0 = wave_sum_fast(sc0);
      if (lane == 0) {
        sh.red[wave][16] = s0;
        sh.red[wave][17] = s0 * dc;  // sum r^2 / d (constant diagonal)
        // the next member rides on the same all-reduce: drawn by the group's first workgroup (exact below 2^24)
        sh.red[wave][18] = (wig == 0 && wave == 0) ? (float)(ngroups + atomicAdd(a.next_member, 1)) : 0.f;
      }
    }
    const PfSlots ps = pf_publish<GW>(sh, 19, g);
    // while the partial sums travel: store r, update x
    float pv[NRK];
#pragma unroll
    for (int q = 0; q < NRK; ++q) {
      const int lr = t + R4_TPB * q;
      pv[q] = 0.f;
      if (lr < nv) {
        pv[q] = a.p[mb + lr];
        a.r[mb + lr] = rv[q];
        a.x[mb + lr] = fmaf(al, pv[q], a.x[mb + lr]);  // x += alpha p      :31
      }
    }
    pf_collect<GW>(sh, 19, g, ps);
    // u = F w (every lane: all 16 entries, F rows from LDS as broadcast reads), r.z = sum r^2/d - w.u
    float cm[16];  // b_m[i2] u_m / sigma
    float uu = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float* row = kf_s + m * 16;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        const float4 f4 = *reinterpret_cast<const float4*>(row + j);
        const float4 w4 = *reinterpret_cast<const float4*>(&sh.res[j]);
        s2 = fmaf(f4.x, w4.x, s2); s2 = fmaf(f4.y, w4.y, s2); s2 = fmaf(f4.z, w4.z, s2); s2 = fmaf(f4.w, w4.w, s2);
      }
      uu = fmaf(sh.res[m], s2, uu);
      cm[m] = bm[m] * (s2 * dc);
    }
    const float srr = sh.res[16], srz = sh.res[17] - uu;
    const float rzo = rzo_s;
    const float beta = (rzo < a.eps) ? 0.f : srz / rzo;  // the control step's rule (:39-42) on the same numbers
    // (uniform by construction; readfirstlane tells the compiler, so that the a_.[i1] loads stay scalar)
    const int64_t bnext = (int64_t)__builtin_amdgcn_readfirstlane((int)sh.res[18]);
    int cf_old = -1;
    if (a.cf.on && wig == 0 && t == 0) {  // control step of this member: see k_precond_fused
      float rn = sqrtf(srr);
      if (a.cf.rhs_is_zero[b]) rn = 0.f;
      a.cf.rz[b] = srz;
      a.cf.beta[b] = beta;
      a.cf.resid_norm[b] = rn;
      a.cf.has_conv[b] = rn < a.cf.stop_after;
      __hip_atomic_store(a.cf.gran + b, ((unsigned long long)(unsigned)(a.launch + 1) << 32) | __float_as_uint(rn),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cf_old = atomicAdd(a.cf.done, 1);
    }
#pragma unroll
    for (int q = 0; q < NRK; ++q) {
      const int lr = t + R4_TPB * q;
      const cfloat_ptr ar = as_const_space(kab + (size_t)min(i1w + q * i1step, i1max) * 16);
      float zq = 0.f;
#pragma unroll
      for (int m = 0; m < 16; ++m) zq = fmaf(ar[m], cm[m], zq);
      // z = r/sigma - KP F KP^T r / sigma^2 goes straight into p <- z + beta p (:46)
      if (lr < nv) a.p[mb + lr] = fmaf(pv[q], beta, fmaf(rv[q], dc, -zq));
    }
    if (wig == 0 && t < a.S) {  // the control step sums S partials per member: the totals go to slot 0
      a.rr_part[(size_t)b * a.S + t] = (t == 0) ? srr : 0.f;
      a.rz_part[(size_t)b * a.S + t] = (t == 0) ? srz : 0.f;
    }
    if (a.cf.on && wig == 0 && t < 64) {
      const int last = __builtin_amdgcn_readfirstlane((cf_old == (int)a.B - 1) ? 1 : 0);
      if (last) {  // every member of the batch is recorded: the batch-global decisions (cg_ctrl_body), fixed order
        float ls = 0.f;
        const unsigned want = (unsigned)(a.launch + 1);
        for (int64_t i = t; i < a.B; i += 64) {
          unsigned long long gq;
          unsigned spin = 0;
          do {
            gq = __hip_atomic_load(a.cf.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned)(gq >> 32) != want && ++spin < R4_MAXSPIN);
          if ((unsigned)(gq >> 32) != want) atomicExch(a.err, 1);
          ls += __uint_as_float((unsigned)(gq & 0xffffffffull));
        }
        const float mean = wave_sum_fast(ls) / (float)a.B;
        if (t == 0) {
          const int k = a.launch;
          a.cf.ctrl->iterations = k + 1;
          a.cf.ctrl->mean_resid = mean;
          if (k >= a.cf.kfloor && mean < a.cf.tol) {  // :302-306 (no tridiagonal columns on this path)
            a.cf.ctrl->tol_reached = 1;
            a.cf.ctrl->stop = 1;
          }
        }
      }
    }
    __syncthreads();  // (sh.res / alpha_s / rzo_s / kf_s are reused by the next member)
    b = bnext;
  }
}

// second-factor sizes the kernel takes: n2 divides 256 (a thread's rows share i2) and n2 >= 64 (a wave's rows share i1)
bool precond_fused_kron_eligible(int64_t n1, int64_t n2) {
  return n1 >= 1 && (n2 == 64 || n2 == 128 || n2 == 256) && n1 * n2 < (1 << 24) && !getenv("LO_NO_KRON_ROOT");
}

static int pf_group_size(int64_t N) { return N <= 16 * (int64_t)R4_ROWS ? 16 : (N <= 32 * (int64_t)R4_ROWS ? 32 : 64); }

bool precond_fused_eligible(int64_t B, int64_t N, int64_t c, int ldq, int S) {
  return c == 1 && ldq == 16 && N >= 8 * (int64_t)R4_ROWS && N <= 64 * (int64_t)R4_ROWS && B < (1 << 24) - 4096 &&
         S <= R4_TPB && !getenv("LO_NO_FUSED_PRECOND");
}

size_t precond_fused_gbuf_bytes() { return (size_t)64 * 2 * 65 * R4_SLOT * sizeof(unsigned long long) + 256; }

template <int GW, int OCC, int NRK>  // NRK = 0: the Q form; 4 / 8: Kronecker root form with NRK rows per thread
static int pf_go(PfArgs& a, int ncu, hipStream_t st) {
  int per_cu = 0;
  constexpr bool KR = NRK > 0;
  constexpr int NK = KR ? NRK : 4;
  const hipError_t oe =
      KR ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_precond_fused_kron<GW, OCC, NK>, R4_TPB, 0)
         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_precond_fused<GW, OCC>, R4_TPB, 0);
  if (oe != hipSuccess || per_cu < 1) return LO_ERR_UNSUPPORTED;
  per_cu = std::min(per_cu, OCC);
  const int nwg = per_cu * ncu;
  if ((nwg / 8) < GW) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN(KR ? "precond_fused_kron" : "precond_fused", st);
  ResidentLaunch guard(st);
  if (KR) hipLaunchKernelGGL((k_precond_fused_kron<GW, OCC, NK>), dim3(nwg), dim3(R4_TPB), 0, st, a);
  else hipLaunchKernelGGL((k_precond_fused<GW, OCC>), dim3(nwg), dim3(R4_TPB), 0, st, a);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// gbuf: precond_fused_gbuf_bytes() of memory the CALLER zeroed once per solve, next_member: base of one zeroed int PER
// LAUNCH of the solve (max_launch + 1 of them); `launch` = index of this launch within the solve: its tags
// start at launch * (B + 2), so nothing has to be cleared between the iterations (two memset launches per CG iteration
// were 3 % of the cfg4 iteration).  LO_ERR_UNSUPPORTED when the tag space would wrap (the caller falls back).
int precond_fused_rupdate(const float* Q, const float* dinv, int dinv_mode, float* r, const float* Ap, float* p,
                          float* x, float* z, const float* pAp_part, int S_dot, const float* rz, const int* has_conv,
                          float eps, float* alpha_out, float* rr_part, float* rz_part, int S, int64_t B, int64_t N,
                          unsigned long long* gbuf, int* err, int* next_member, int launch, const int* iter_ptr,
                          int max_launch, const int* stop, int ncu, const PfCtrl* cf, const PfKron* kr,
                          hipStream_t st) {
  PfArgs a;
  const bool kron = kr && kr->a && kr->b && kr->F && dinv_mode == LO_DIAG_CONST &&
                    precond_fused_kron_eligible(kr->n1, kr->n2);
  a.ka = kron ? kr->a : nullptr; a.kb = kron ? kr->b : nullptr; a.kF = kron ? kr->F : nullptr;
  a.n1 = kron ? kr->n1 : 0; a.n2 = kron ? kr->n2 : 1;
  a.inv_n2 = 1.0f / (float)a.n2;
  a.launch = launch;
  if (cf) {
    a.cf = *cf;
    if (a.cf.on && !iter_ptr) a.cf.done += launch;
  } else {
    a.cf.on = 0;
  }
  a.Q = Q; a.dinv = dinv; a.dinv_mode = dinv_mode; a.r = r; a.Ap = Ap; a.p = p; a.x = x; a.z = z;
  a.pAp_part = pAp_part; a.S_dot = S_dot; a.rz = rz; a.has_conv = has_conv; a.eps = eps; a.alpha_out = alpha_out;
  a.rr_part = rr_part; a.rz_part = rz_part; a.S = S; a.B = B; a.N = (int)N;
  int GW = pf_group_size(N);
  a.RW = (int)((N + GW - 1) / GW);
  // Kronecker root form: 8 rows per thread (groups of 16 / 32 workgroups of 2048 rows) once the batch fills the device
  // that way, else 4 (cfg4 shard: 74.7 us with 4 rows, 61.7 with 8, 62.2 with 16); a workgroup's first row is a
  // multiple of 256 (wave-uniform first-factor index)
  const bool kron8 = kron && N > 8 * (int64_t)2048 && B * ((N + 2047) / 2048) >= 1024 && !getenv("LO_KRON_NR4");
  if (kron8) GW = N <= 16 * (int64_t)2048 ? 16 : 32;
  if (kron) a.RW = (int)(((N + (int64_t)GW * 256 - 1) / ((int64_t)GW * 256)) * 256);
  a.gbuf = gbuf; a.err = err; a.next_member = next_member; a.stop = stop;
  a.allow_l2_handoff = onchip_l2_handoff_allowed();
  // next_member: base of the per-launch counters; launch: index of this launch (iter_ptr: read on the device instead)
  const unsigned long long base = (unsigned long long)launch * (unsigned long long)(B + 2);
  if ((unsigned long long)(max_launch + 1) * (unsigned long long)(B + 2) >= 0xffff0000ull) return LO_ERR_UNSUPPORTED;
  a.tag_base = (unsigned)base;
  a.iter_ptr = iter_ptr;
  if (!iter_ptr) a.next_member = next_member + launch;
  // four workgroups per CU (110 VGPRs): cfg4 153 us per call against 188 us with two or three
  if (kron8) return GW == 16 ? pf_go<16, 4, 8>(a, ncu, st) : pf_go<32, 4, 8>(a, ncu, st);
  if (kron) {
    if (GW == 16) return pf_go<16, 4, 4>(a, ncu, st);
    if (GW == 32) return pf_go<32, 4, 4>(a, ncu, st);
    return pf_go<64, 4, 4>(a, ncu, st);
  }
  if (GW == 16) return pf_go<16, 4, 0>(a, ncu, st);
  if (GW == 32) return pf_go<32, 4, 0>(a, ncu, st);
  return pf_go<64, 4, 0>(a, ncu, st);
}

}  // namespace lo

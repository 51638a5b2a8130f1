// lo_prof.hip -- opt-in per-kernel timing with HIP events on the launch stream (used by bench.py to report
// the live average duration of the dominant kernel for the roofline line; off by default, zero cost then).
#include <map>
#include <string>
#include <vector>

#include "lo_internal.h"

namespace lo {

bool g_prof_on = false;

namespace {
struct Rec {
  const char* name;
  hipEvent_t a, b;
};
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void prof_start(const char* name, hipStream_t st) {
  Rec r;
  r.name = name;
  r.a = take_event();
  r.b = take_event();
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}
void prof_stop(hipStream_t st) { (void)hipEventRecord(g_recs.back().b, st); }

// BabelStream-style kernels over n floats: the achievable HBM rate of this box, reported next to the 8 TB/s spec peak
// (SURVEY 8(d)).  Every thread moves U float4 that lie one workgroup-width apart (a wave instruction touches 1 KiB of
// consecutive bytes), all U loads are issued before the first use, the grid covers the arrays exactly once (no
// grid-stride loop: a fixed grid of 8192 workgroups left the 1 GiB arrays to 32 serial trips per thread with one
// 16-byte request in flight -- 4.4 TB/s where the dense matvec of this library streams at 5.3), and the streams are
// marked non-temporal (nothing is reused: keep them out of the L2's way).
//   MODE 0: a = b + s c (triad, 12 bytes per element)   1: a = b (copy, 8 bytes)   2: read b, c only (8 bytes; the
//   sum is written by the threads that see a NaN-poisoned value, i.e. never)
typedef float vf4 __attribute__((ext_vector_type(4)));  // (the non-temporal builtins want a native vector type)
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(256) void k_stream(vf4* __restrict__ a, const vf4* __restrict__ b,
                                                const vf4* __restrict__ c, float s, size_t n4) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  vf4 x[U], y[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = base + (size_t)u * 256;
    x[u] = y[u] = vf4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      if (NT) {
        x[u] = __builtin_nontemporal_load(b + i);
        if (MODE != 1) y[u] = __builtin_nontemporal_load(c + i);
      } else {
        x[u] = b[i];
        if (MODE != 1) y[u] = c[i];
      }
    }
  }
  if (MODE == 2) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (x[u].x + x[u].y + x[u].z + x[u].w) * (y[u].x + y[u].y + y[u].z + y[u].w);
    if (acc != acc) a[base] = vf4{acc, acc, acc, acc};
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n4) {
      vf4 r = x[u];
      if (MODE == 0) r = r + s * y[u];
      if (NT)
        __builtin_nontemporal_store(r, a + i);
      else
        a[i] = r;
    }
  }
}

template <int MODE, int U>
int launch_stream(bool nt, float* a, const float* b, const float* c, float s, size_t n4, hipStream_t st) {
  const size_t per = 256 * (size_t)U;
  const dim3 grid((unsigned)((n4 + per - 1) / per));
  if (nt)
    hipLaunchKernelGGL((k_stream<MODE, U, true>), grid, dim3(256), 0, st, (vf4*)a, (const vf4*)b, (const vf4*)c, s, n4);
  else
    hipLaunchKernelGGL((k_stream<MODE, U, false>), grid, dim3(256), 0, st, (vf4*)a, (const vf4*)b, (const vf4*)c, s, n4);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

template <int MODE>
int launch_stream_u(int unroll, bool nt, float* a, const float* b, const float* c, float s, size_t n4, hipStream_t st) {
  switch (unroll) {
    case 1: return launch_stream<MODE, 1>(nt, a, b, c, s, n4, st);
    case 2: return launch_stream<MODE, 2>(nt, a, b, c, s, n4, st);
    case 4: return launch_stream<MODE, 4>(nt, a, b, c, s, n4, st);
    case 8: return launch_stream<MODE, 8>(nt, a, b, c, s, n4, st);
    default: return LO_ERR_BADARG;
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

// the shapes measured best on MI355X (tools/mb_stream.py sweeps unroll x non-temporal for the three modes)
#define LO_STREAM_UNROLL 4
#define LO_STREAM_NT true
int lo_hbm_triad_f32(float* a, const float* b, const float* c, float s, size_t n, void* stream) {
  if (!a || !b || !c || (n & 3)) return LO_ERR_BADARG;
  return launch_stream<0, LO_STREAM_UNROLL>(LO_STREAM_NT, a, b, c, s, n / 4, (hipStream_t)stream);
}

int lo_hbm_copy_f32(float* a, const float* b, size_t n, void* stream) {
  if (!a || !b || (n & 3)) return LO_ERR_BADARG;
  return launch_stream<1, LO_STREAM_UNROLL>(LO_STREAM_NT, a, b, b, 0.f, n / 4, (hipStream_t)stream);
}

// measurement aid of the measurement aid: mode 0 triad / 1 copy / 2 read-only, unroll in {1, 2, 4, 8}, nt 0 / 1
int lo_hbm_stream_dev(int mode, int unroll, int nt, float* a, const float* b, const float* c, float s, size_t n,
                      void* stream) {
  if (!a || !b || !c || (n & 3)) return LO_ERR_BADARG;
  switch (mode) {
    case 0: return launch_stream_u<0>(unroll, nt != 0, a, b, c, s, n / 4, (hipStream_t)stream);
    case 1: return launch_stream_u<1>(unroll, nt != 0, a, b, c, s, n / 4, (hipStream_t)stream);
    case 2: return launch_stream_u<2>(unroll, nt != 0, a, b, c, s, n / 4, (hipStream_t)stream);
    default: return LO_ERR_BADARG;
  }
}

int lo_prof_enable(int on) {
  g_prof_on = on != 0;
  return LO_OK;
}

// Writes "name count total_ms\n" lines (sorted by name) into buf, resets the records.  Synchronises.
int lo_prof_report(char* buf, size_t buflen) {
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = agg[r.name];
      e.first += 1;
      e.second += ms;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  std::string out;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && buflen) {
    snprintf(buf, buflen, "%s", out.c_str());
  }
  return (int)out.size();
}

}  // extern "C"

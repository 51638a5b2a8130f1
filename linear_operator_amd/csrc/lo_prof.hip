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

// BabelStream-style triad a = b + s c over n floats (float4 per thread, grid-stride): the achievable HBM rate of
// this box, reported next to the 8 TB/s spec peak
__global__ __launch_bounds__(256) void k_triad(float4* __restrict__ a, const float4* __restrict__ b,
                                               const float4* __restrict__ c, float s, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_hbm_triad_f32(float* a, const float* b, const float* c, float s, size_t n, void* stream) {
  if (!a || !b || !c || (n & 3)) return LO_ERR_BADARG;
  hipLaunchKernelGGL(k_triad, dim3(256 * 32), dim3(256), 0, (hipStream_t)stream, (float4*)a, (const float4*)b,
                     (const float4*)c, s, n / 4);
  LO_LAUNCH_CHECK();
  return LO_OK;
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

// How fast can a dense [B, N, N] fp32 operator be streamed in the TILE order of a matrix-vector kernel (a workgroup owns
// R rows and walks along k in slabs of W floats per row), against a flat linear read?  Loads only (summed so that they
// are not removed), no LDS traffic, no MFMA: the HBM-side ceiling of csrc/lo_dense_mfma.hip's access pattern.
//   rows R x slab W (bytes contiguous per row = 4 W), U float4 loads in flight per thread, workgroups per CU capped by a
//   dynamic LDS request, plain / non-temporal loads.
// Build: hipcc --offload-arch=gfx950 -O2 -o dense_stream_pattern dense_stream_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4* p, bool nt) {
  const f4 v = nt ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)) : *reinterpret_cast<const f4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int R, int W, int U, bool NT>
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ K, float* __restrict__ out, int N, int rot) {
  extern __shared__ float pad[];
  const int tile = blockIdx.x, b = blockIdx.y;
  const float* Kb = K + (size_t)b * N * N + (size_t)tile * R * N;
  constexpr int QW = W / 4;             // float4 per row and slab
  constexpr int PER = R * QW / 256;     // float4 per thread and slab
  static_assert(PER >= 1 && (R * QW) % 256 == 0, "slab must be a multiple of the workgroup");
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int SL = (U >= PER) ? U / PER : 1;  // one slab = PER loads per thread; SL slabs in flight
  const int nsl = N / (SL * W);
  const int start = rot ? (int)((blockIdx.x * (unsigned)rot) % (unsigned)nsl) : 0;  // each tile starts its sweep elsewhere
  for (int it = 0; it < nsl; ++it) {
    int kb = (start + it) * SL * W;
    if (kb >= N) kb -= N;
    float4 v[SL * PER];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = threadIdx.x + 256 * u;
        const int r = f / QW, q = f % QW;
        const float4* p = reinterpret_cast<const float4*>(Kb + (size_t)r * N + kb + s * W + 4 * q);
        v[s * PER + u] = (kb + s * W < N) ? ld4(p, NT) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int i = 0; i < SL * PER; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = pad[0];
}

template <bool NT>
__global__ __launch_bounds__(256) void k_flat(const float* __restrict__ K, float* __restrict__ out, size_t n4) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = reinterpret_cast<const float4*>(K);
  const size_t per = n4 / gridDim.x;
  const size_t base = (size_t)blockIdx.x * per;
  for (size_t i = threadIdx.x; i < per; i += 256 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t j = i + 256 * (size_t)u;
      v[u] = j < per ? ld4(p + base + j, NT) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

static float* K; static float* out; static int N = 16384, B = 4;
static hipEvent_t e0, e1;

template <class F>
static void timeit(const char* name, F launch) {
  launch(); launch();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %8.3f ms  %7.1f GB/s\n", name, best, 4.0 * B * N * (double)N / best / 1e6);
  fflush(stdout);
}

template <int R, int W, int U, bool NT>
static void tile(int wg_per_cu, int rot = 0) {
  char name[128];
  snprintf(name, sizeof(name), "tile %3d rows x %4d B, %2d x 16 B in flight, %d WG/CU%s", R, 4 * W, U, wg_per_cu, NT ? ", nt" : "");
  if (rot) snprintf(name + strlen(name), sizeof(name) - strlen(name), ", rot %d", rot);
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu) - 1024;
  hipFuncSetAttribute((const void*)k_tile<R, W, U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  timeit(name, [&] { hipLaunchKernelGGL((k_tile<R, W, U, NT>), dim3(N / R, B), dim3(256), lds, 0, K, out, N, rot); });
}

int main() {
  hipMalloc(&K, (size_t)B * N * N * 4); hipMalloc(&out, 1024);
  hipMemset(K, 0, (size_t)B * N * N * 4);
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n4 = (size_t)B * N * N / 4;
  timeit("flat, 2048 workgroups", [&] { hipLaunchKernelGGL(k_flat<false>, dim3(2048), dim3(256), 0, 0, K, out, n4); });
  timeit("flat, 2048 workgroups, nt", [&] { hipLaunchKernelGGL(k_flat<true>, dim3(2048), dim3(256), 0, 0, K, out, n4); });
  timeit("flat, 16384 workgroups, nt", [&] { hipLaunchKernelGGL(k_flat<true>, dim3(16384), dim3(256), 0, 0, K, out, n4); });
  tile<64, 128, 8, true>(3);   // the shape of k_dense_mv_mfma16 today
  tile<64, 128, 8, true>(3, 1);
  tile<64, 128, 8, true>(3, 5);
  tile<64, 128, 8, true>(3, 37);
  tile<64, 128, 16, true>(3, 1);
  tile<64, 128, 16, true>(3, 37);
  tile<128, 64, 8, true>(3);
  tile<128, 64, 8, true>(3, 1);
  tile<128, 64, 8, true>(3, 37);
  tile<128, 64, 16, true>(3);
  tile<128, 64, 16, true>(3, 1);
  tile<128, 64, 16, true>(3, 37);
  tile<128, 64, 16, true>(2, 37);
  tile<128, 128, 16, true>(3, 37);
  tile<128, 128, 16, true>(2, 37);
  tile<64, 64, 8, true>(3, 37);
  tile<64, 64, 8, true>(4, 1);
  tile<32, 256, 16, true>(3, 37);
  return 0;
}

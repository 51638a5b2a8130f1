// lo_cg_onchip.hip -- what the operator-resident kernels share on the host side: the ordering of resident launches across
// streams (ResidentLaunch), the workgroup count they are sized by (onchip_num_workgroups, LO_OC_RESERVE_CUS), the opt-in of
// the same-XCD plain-store hand-off, the granule budget of the serial kernels.
// (Rounds 1 - 5 also kept the FIRST generation of the resident CG here -- k_cg_onchip: one row per thread, C in LDS, four
// group all-reduces per iteration, 0.95 ms per headline solve.  Nothing had selected it since k_cg_onchip4 (round 1) and
// k_cg_onchip5 (round 2); removed in round 6 with its plan rows.  The design notes of the hand-off -- tagged 8-byte
// granules, two slot sets by phase parity, bounded polls -- live in lo_group_reduce.h and DESIGN.md 4.2.)
#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include "lo_device.h"
#include <string.h>

#include "lo_internal.h"
#include "lo_cg_onchip.h"

namespace lo {

constexpr int OC_GW = 8;  // (granule budget of the serial resident kernels: groups of up to 8 x 40 granules per parity)

size_t onchip_gbuf_bytes(int ngroups) { return (size_t)ngroups * 2 * OC_GW * 40 * sizeof(unsigned long long); }

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

}  // namespace lo

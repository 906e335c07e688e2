// Optional in-library kernel timing with HIP events recorded on the SAME stream the kernels are
// launched on (bench.py's roofline leg; torch.cuda.Event only sees torch's current stream and
// cannot bracket one kernel inside tbe_backward_fused_f32).  Off by default: when disabled no
// event is created or recorded.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/tbe_hip.h"

static constexpr int kProfileRowSlots = 64;  // must equal tbe::kProfileRowSlots (common.hpp)

namespace tbe {

struct Span {
  hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Span> g_spans[TBE_PROFILE_NUM_SLOTS];

bool profile_enabled() { return g_on; }

static unsigned long long* g_rows_dev = nullptr;
unsigned long long* profile_unique_rows_counter() { return g_on ? g_rows_dev : nullptr; }

// Returns an event to record before the launch (and registers its partner, returned via *after).
void profile_begin(int slot, hipStream_t st, hipEvent_t* after) {
  *after = nullptr;
  if (!g_on || slot < 0 || slot >= TBE_PROFILE_NUM_SLOTS) return;
  Span s;
  if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_spans[slot].size() >= 65536) {
      (void)hipEventDestroy(s.a);
      (void)hipEventDestroy(s.b);
      return;
    }
    g_spans[slot].push_back(s);
  }
  (void)hipEventRecord(s.a, st);
  *after = s.b;
}

void profile_end(hipEvent_t after, hipStream_t st) {
  if (after != nullptr) (void)hipEventRecord(after, st);
}

}  // namespace tbe

extern "C" int tbe_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(tbe::g_mu);
  if (on && tbe::g_rows_dev == nullptr) {
    // kProfileRowSlots counters, one per 128-B line: the update kernel's workgroups add to different lines
    if (hipMalloc(&tbe::g_rows_dev, kProfileRowSlots * 128) != hipSuccess) return TBE_ERR_LAUNCH;
    (void)hipMemset(tbe::g_rows_dev, 0, kProfileRowSlots * 128);
  }
  tbe::g_on = on != 0;
  return TBE_OK;
}

extern "C" int tbe_profile_read(int32_t slot, double* total_ms, int64_t* count) {
  if (slot < 0 || slot >= TBE_PROFILE_NUM_SLOTS || !total_ms || !count) return TBE_ERR_INVALID_ARGUMENT;
  std::vector<tbe::Span> spans;
  {
    std::lock_guard<std::mutex> lk(tbe::g_mu);
    spans.swap(tbe::g_spans[slot]);
  }
  double tot = 0.0;
  int64_t n = 0;
  for (auto& s : spans) {
    float ms = 0.f;
    if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      tot += ms;
      ++n;
    }
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  *total_ms = tot;
  *count = n;
  return TBE_OK;
}

extern "C" int tbe_profile_read_rows(int64_t* rows_updated) {
  if (!rows_updated) return TBE_ERR_INVALID_ARGUMENT;
  *rows_updated = 0;
  if (tbe::g_rows_dev == nullptr) return TBE_OK;
  std::vector<unsigned long long> v(kProfileRowSlots * 16);
  if (hipDeviceSynchronize() != hipSuccess) return TBE_ERR_LAUNCH;
  if (hipMemcpy(v.data(), tbe::g_rows_dev, kProfileRowSlots * 128, hipMemcpyDeviceToHost) != hipSuccess) return TBE_ERR_LAUNCH;
  (void)hipMemset(tbe::g_rows_dev, 0, kProfileRowSlots * 128);
  unsigned long long tot = 0;
  for (int i = 0; i < kProfileRowSlots; ++i) tot += v[i * 16];
  *rows_updated = static_cast<int64_t>(tot);
  return TBE_OK;
}

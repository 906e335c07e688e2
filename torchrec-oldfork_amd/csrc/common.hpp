// Shared helpers for the gfx950 TBE kernels.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/tbe_hip.h"

namespace tbe {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define TBE_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::tbe::set_error(__VA_ARGS__);      \
      return TBE_ERR_INVALID_ARGUMENT;    \
    }                                     \
  } while (0)

#define TBE_CHECK_LAUNCH(what)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      ::tbe::set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return TBE_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// profile.cpp: optional HIP-event bracketing of a launch (no-op unless tbe_profile_enable(1)).
bool profile_enabled();
void profile_begin(int slot, hipStream_t st, hipEvent_t* after);
void profile_end(hipEvent_t after, hipStream_t st);
constexpr int kProfileRowSlots = 64;  // counters of 128 B each
unsigned long long* profile_unique_rows_counter();  // kProfileRowSlots device counters (one per 128 B), nullptr unless profiling
struct ProfileSpan {
  hipEvent_t after = nullptr;
  hipStream_t st;
  ProfileSpan(int slot, hipStream_t s) : st(s) {
    if (profile_enabled()) profile_begin(slot, s, &after);
  }
  ~ProfileSpan() { profile_end(after, st); }
};

// Carves 256-byte aligned sub-buffers out of a caller workspace.
struct Carver {
  char* base;
  size_t used = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    used = align_up(used, 256);
    T* r = reinterpret_cast<T*>(base ? base + used : nullptr);
    used += count * sizeof(T);
    return r;
  }
  void* take_bytes(size_t bytes) {
    used = align_up(used, 256);
    void* r = base ? base + used : nullptr;
    used += bytes;
    return r;
  }
  size_t total() const { return align_up(used, 256); }
};

// Row ownership of a feature's table shard (row-wise sharding without a bucketize pass: every rank sees every
// id of the feature).  With a window the ids are GLOBAL rows of the table and the shard holds
// [lo, lo + rows); without one (feat_window == nullptr) lo = 0 and global_rows = rows.
struct RowWindow {
  int64_t lo, rows, global_rows;
};
__device__ __forceinline__ RowWindow load_window(const int64_t* feat_rows, const int64_t* feat_window, int f) {
  RowWindow w;
  w.rows = feat_rows[f];
  w.lo = feat_window != nullptr ? feat_window[2 * f] : 0;
  w.global_rows = feat_window != nullptr ? feat_window[2 * f + 1] : w.rows;
  return w;
}
constexpr int kIdLocal = 0;    // a row of this shard: `local` = id - lo
constexpr int kIdForeign = 1;  // another shard's row, or TBE_ID_SKIP: contributes nothing, silently
constexpr int kIdBad = 2;      // outside [0, global_rows): contributes nothing and is counted in bounds_errors
__device__ __forceinline__ int classify_id(const RowWindow& w, int64_t id, int64_t& local) {
  local = static_cast<int64_t>(static_cast<uint64_t>(id) - static_cast<uint64_t>(w.lo));
  if (static_cast<uint64_t>(local) < static_cast<uint64_t>(w.rows)) return kIdLocal;
  if (id == TBE_ID_SKIP || static_cast<uint64_t>(id) < static_cast<uint64_t>(w.global_rows)) return kIdForeign;
  return kIdBad;
}

// Broadcast lane `src` (0..63) of a 64-bit value.
__device__ __forceinline__ int64_t shfl64(int64_t v, int src) {
  int lo = __shfl(static_cast<int>(v & 0xffffffffll), src, kWave);
  int hi = __shfl(static_cast<int>(v >> 32), src, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}
__device__ __forceinline__ uint64_t shflu64(uint64_t v, int src) {
  return static_cast<uint64_t>(shfl64(static_cast<int64_t>(v), src));
}

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}

}  // namespace tbe

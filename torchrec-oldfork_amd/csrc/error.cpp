// Thread-local error message for the C ABI (include/tbe_hip.h) and the library's FAULT WORD.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "../../include/tbe_hip.h"

namespace tbe {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The fault word: one 64-byte line of pinned, GPU-mapped, coherent host memory that kernels write when they give up
// on something that makes their RESULT wrong (today: a spin-wait of the pair sort that outlived kSpinLimit).  It lives
// in host memory so that the host can look at it at any time without a copy, a stream operation or a sync: the
// normal path pays nothing, the fault path is one system-scope store + add.  word 0 = sticky flag (plain store: works
// without PCIe atomics), word 1 = count (system-scope atomic add, best effort).
static std::mutex g_fault_mu;
static uint32_t* g_fault_host = nullptr;    // hipHostMalloc'ed line, or the fallback below when there is no device
static uint32_t g_fault_fallback[16] = {};  // no HIP device in this process (CPU-only container): host-side counting only
static bool g_fault_tried = false;

static uint32_t* fault_alloc_locked() {
  if (!g_fault_tried) {
    g_fault_tried = true;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && p != nullptr) {
      g_fault_host = static_cast<uint32_t*>(p);
      for (int i = 0; i < 16; ++i) g_fault_host[i] = 0u;
    } else {
      (void)hipGetLastError();
    }
  }
  return g_fault_host;
}

// Device-visible address of the fault word (unified addressing: the pinned line is mapped at its host address), or
// nullptr when it cannot be allocated — callers that launch a kernel which may need it must fail then.
uint32_t* fault_word_device() {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  uint32_t* h = fault_alloc_locked();
  if (h == nullptr) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || d == nullptr) {
    (void)hipGetLastError();
    return nullptr;
  }
  return static_cast<uint32_t*>(d);
}

static volatile uint32_t* fault_word_host() {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  uint32_t* h = fault_alloc_locked();
  return h != nullptr ? h : g_fault_fallback;
}
}  // namespace tbe

extern "C" const char* tbe_last_error(void) { return tbe::g_err; }
extern "C" int32_t tbe_abi_version(void) { return 3; }

extern "C" int tbe_fault_status(int64_t* sort_giveups) {
  if (sort_giveups == nullptr) {
    tbe::set_error("tbe_fault_status: null pointer");
    return TBE_ERR_INVALID_ARGUMENT;
  }
  volatile uint32_t* w = tbe::fault_word_host();
  const uint32_t flag = w[0], count = w[1];
  *sort_giveups = count != 0u ? count : (flag != 0u ? 1 : 0);
  return TBE_OK;
}

extern "C" int tbe_debug_inject_fault_host(void) {
  volatile uint32_t* w = tbe::fault_word_host();
  w[0] = 1u;
  w[1] = w[1] + 1u;
  return TBE_OK;
}

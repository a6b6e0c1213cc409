// Error channel + version of libsrganst.so (SURVEY.md 8b "Error convention").
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

int sst_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

SST_API const char* sst_last_error(void) { return g_err; }

// ---- dev switches, read once (common.h: sst_env)
#include <cstdlib>
#include <mutex>
namespace {
struct EnvEntry { char name[40]; char value[24]; bool set; };
constexpr int ENV_MAX = 64;
EnvEntry g_env[ENV_MAX];
int g_env_n = 0;
std::mutex g_env_mu;
}  // namespace
const char* sst_env(const char* name) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  for (int i = 0; i < g_env_n; ++i)
    if (!strcmp(g_env[i].name, name)) return g_env[i].set ? g_env[i].value : nullptr;
  const char* v = std::getenv(name);
  if (g_env_n == ENV_MAX || strlen(name) >= sizeof(g_env[0].name)) return v;      // table full: uncached (never happens with the switches in tree)
  EnvEntry& e = g_env[g_env_n++];
  snprintf(e.name, sizeof(e.name), "%s", name);
  e.set = v != nullptr;
  snprintf(e.value, sizeof(e.value), "%s", v ? v : "");
  return e.set ? e.value : nullptr;
}
SST_API int sst_reload_env(void) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  g_env_n = 0;
  return SST_OK;
}
SST_API int sst_version(void) { return 100; }  // 0.1.0
SST_API const char* sst_arch(void) { return "gfx950"; }
// Forget the runtime's sticky "last error" (hipGetLastError reads AND clears it): a failed stream capture leaves one behind, and
// the launch check of the next - perfectly good - eager launch would report it.  Returns the code that was pending (0 = none).
SST_API int sst_clear_error(void) { g_err[0] = 0; return (int)hipGetLastError(); }

// ---- measurement hook (tools/mfma_peak.py): what the chip sustains on the fp32 MFMA the conv kernels are built on, with no
// memory traffic at all - the clock a long MFMA-dense launch really holds sets the ceiling the conv kernels are read against.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  f32x16_t acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  unsigned ua = threadIdx.x * 2654435761u + blockIdx.x, ub = ua ^ 0x9e3779b9u;
  for (int i = 0; i < iters; ++i) {
    // fresh pseudo-random operands in [-0.5, 0.5) every step (an LCG on the mantissa bits): the clock the chip holds depends
    // on operand toggling, constant operands would flatter it
    ua = ua * 1664525u + 1013904223u;
    ub = ub * 22695477u + 1u;
    const float a = __uint_as_float((ua >> 9) | 0x3f800000u) - 1.5f, b = __uint_as_float((ub >> 9) | 0x3f800000u) - 1.5f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) out[0] = s;
}
SST_API int sst_debug_mfma_peak(float* out, int blocks, int iters, void* stream) {
  SST_REQUIRE(out && blocks > 0 && iters > 0, "sst_debug_mfma_peak: bad argument");
  mfma_peak_kernel<0><<<blocks, 256, 0, sst_stream(stream)>>>(out, iters);
  SST_LAUNCH_CHECK("mfma_peak_kernel");
  return SST_OK;
}

// ---- measurement hook (tools/stamp_step.py): one thread writes the device's constant-rate wall clock (100 MHz) into out[slot].
// Dropped into a captured step at the points of interest, it shows when each branch of the launch DAG really starts and ends in
// an UNPROFILED replay (the profiler's own launch overhead moves exactly those points).
__global__ void stamp_kernel(unsigned long long* out, int slot) { out[slot] = wall_clock64(); }
SST_API int sst_debug_stamp(unsigned long long* out, int slot, void* stream) {
  SST_REQUIRE(out && slot >= 0, "sst_debug_stamp: bad argument");
  stamp_kernel<<<1, 1, 0, sst_stream(stream)>>>(out, slot);
  SST_LAUNCH_CHECK("stamp_kernel");
  return SST_OK;
}

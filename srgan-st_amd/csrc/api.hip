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

// ---- measurement hook (tools/bf16x3_probe.py): the split-operand proposal of DESIGN.md section 8, measured instead of argued.  An fp32
// operand is split into three bf16 terms x = x1 + x2 + x3 (each the round-to-nearest bf16 of the running residual: 3 x 8 significand
// bits); a product keeps the six cross terms of order >= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1); the bf16 MFMA multiplies exactly
// and accumulates in fp32.  (a) accuracy: C = A[32 x K] * B[K x 32] by v_mfma_f32_32x32x2_f32 and by the six-MFMA split form of
// v_mfma_f32_32x32x16_bf16, one wave each, both written out for a host comparison against fp64; (b) rate: the six-MFMA group in a
// register-only loop, to be compared with sst_debug_mfma_peak.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}
__global__ __launch_bounds__(64) void bf16x3_probe_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C32,
                                                          float* __restrict__ C3, int K) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16_t acc32, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc32[r] = acc3[r] = 0.f;
  for (int k = 0; k < K; k += 2)                           // fp32 MFMA: lane (li, lh) holds A[li][k + lh], B[k + lh][li]
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + lh], B[(k + lh) * 32 + li], acc32, 0, 0, 0);
  for (int k = 0; k < K; k += 16) {                        // bf16 MFMA: lane (li, lh) holds 8 consecutive k: k + 8 lh + j
    bf16x8_t a[3], b[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 t0, t1, t2;
      split3(A[li * K + k + 8 * lh + j], t0, t1, t2);
      a[0][j] = t0; a[1][j] = t1; a[2][j] = t2;
      split3(B[(k + 8 * lh + j) * 32 + li], t0, t1, t2);
      b[0][j] = t0; b[1][j] = t1; b[2][j] = t2;
    }
    // smallest terms first
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc3, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {                           // acc[r]: row (r & 3) + 8 (r >> 2) + 4 lh, column li
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C32[row * 32 + li] = acc32[r];
    C3[row * 32 + li] = acc3[r];
  }
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void bf16x3_rate_kernel(float* out, int iters) {
  f32x16_t acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  // operands as raw bf16 bit patterns in [1, 2) / (-2, -1]; every iteration flips sign / mantissa bits of two of the 24 operand words
  // (three vector instructions per 24 MFMAs: the loop measures the matrix pipe, not the operand generator - a first version that
  // built fresh operands with conversions every step spent more vector cycles than MFMA cycles and read 200 TFLOP/s)
  u32x4_t a[3], b[3];
  unsigned ua = threadIdx.x * 2654435761u + blockIdx.x;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ua = ua * 1664525u + 1013904223u;
      a[t][j] = (ua & 0x807F807Fu) | 0x3F803F80u;
      ua = ua * 1664525u + 1013904223u;
      b[t][j] = (ua & 0x807F807Fu) | 0x3F803F80u;
    }
  for (int i = 0; i < iters; i += 12) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      ua = ua * 1664525u + 1013904223u;
      a[u % 3][u / 3] ^= ua & 0x807F807Fu;
      b[(u + 1) % 3][u / 3] ^= (ua >> 3) & 0x807F807Fu;
      bf16x8_t av[3], bv[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        av[t] = __builtin_bit_cast(bf16x8_t, a[t]);
        bv[t] = __builtin_bit_cast(bf16x8_t, b[t]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {                        // one fp32-equivalent 32 x 32 x 16 step on each of 4 accumulators
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc[k], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) out[0] = s;
}
// mode 0: accuracy probe (A [32][K], B [K][32], C32 / C3 [32][32], K % 16 == 0); mode 1: rate probe (A = out buffer, K = iterations,
// C3's low bits = blocks)
SST_API int sst_debug_bf16x3(const float* A, const float* B, float* C32, float* C3, int K, int mode, int blocks, void* stream) {
  if (mode == 0) {
    SST_REQUIRE(A && B && C32 && C3 && K > 0 && K % 16 == 0, "sst_debug_bf16x3: bad argument");
    bf16x3_probe_kernel<<<1, 64, 0, sst_stream(stream)>>>(A, B, C32, C3, K);
  } else {
    SST_REQUIRE(C32 && K > 0 && blocks > 0, "sst_debug_bf16x3: bad argument");
    bf16x3_rate_kernel<<<blocks, 256, 0, sst_stream(stream)>>>(C32, K);
  }
  SST_LAUNCH_CHECK("bf16x3 probe");
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

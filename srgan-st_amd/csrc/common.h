// Internal helpers shared by every translation unit of libsrganst.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#define SST_API extern "C" __attribute__((visibility("default")))

// ---- error convention: every entry returns int, 0 = ok, message via sst_last_error()
enum : int {
  SST_OK = 0,
  SST_ERR_ARG = -1,        // bad argument (null pointer, unsupported shape)
  SST_ERR_UNSUPPORTED = -2,
  SST_ERR_HIP = -3,        // a HIP runtime call / launch failed
};

int sst_set_error(int code, const char* fmt, ...);

#define SST_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return sst_set_error(SST_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define SST_LAUNCH_CHECK(name)                                                                   \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess) return sst_set_error(SST_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Dev switches (SST_* environment variables): each is read from the environment ONCE, at its first use, and served from a table
// afterwards, so that a shape query (slab / workspace size) and the launch it sizes always see the same value; sst_reload_env()
// forgets the table (tests that toggle a switch call it).  Returns the value or nullptr (unset).
const char* sst_env(const char* name);

static inline hipStream_t sst_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a workgroup of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
// `red` is LDS scratch of at least NT/64 floats.  Deterministic order.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// The same in fp64 (signed sums that largely cancel: bias / BatchNorm / slope gradients).
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NT>
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// Agent-scope hand-off used by "last block reduces" epilogues (cdna guide, Guideline 16):
// the single publishing lane stores its partial(s), releases, then takes a ticket.
__device__ __forceinline__ unsigned publish_and_ticket(unsigned* counter) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void acquire_after_ticket() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ float load_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Last-arriver total of per-workgroup partials (fixed order => reproducible).  Call from ALL threads of the
// workgroup after thread 0 has stored partials[blockIdx] (plain store).  Returns true in the last-arriving
// workgroup, where `total` (valid in every thread) is the sum of partials[0..nblk).  `s_flag`/`red` are LDS.
template <int NT>
__device__ __forceinline__ bool last_block_total(const float* partials, unsigned* counter, unsigned nblk, unsigned* s_flag,
                                                 float* red, float& total) {
  if (threadIdx.x == 0) *s_flag = (publish_and_ticket(counter) == nblk - 1) ? 1u : 0u;
  __syncthreads();
  if (!*s_flag) return false;
  if (threadIdx.x == 0) acquire_after_ticket();
  __syncthreads();
  float t = 0.f;
  for (unsigned i = threadIdx.x; i < nblk; i += NT) t += load_agent(partials + i);
  total = block_sum<NT>(t, red);
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Adam over ONE flat parameter buffer (replaces the multi-tensor launches of torch.optim.Adam(fused=True) that the
// step engines used before; reference train.py:62-75 / warmup.py:34-40 build torch.optim.Adam with eps 1e-4).
//
// The parameters of a network live in one flat fp32 buffer (srganst/ops.py: flatten_params), the gradients in a flat
// buffer of the same layout (flat_grads), so the update is a single streaming pass: 4 reads + 3 writes per element,
// HBM-bound.  Arithmetic follows torch's fused Adam kernel (non-amsgrad, maximize = False):
//   g' = g + wd * p;  m = lerp(m, g', 1 - b1);  v = b2 * v + (1 - b2) * g'^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// lr and the step counters are device memory, so the launch is hipGraph-capturable and LR schedulers keep working.
#include "common.h"
#include <cmath>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// steps[i] += 1 for the per-parameter step counters torch's state_dict exposes (all equal); separate tiny launch so that
// the update kernel's workgroups all see the same value.
__global__ void adam_tick_kernel(float* steps, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) steps[i] += 1.f;
}

// Constants are doubles and the moment updates are evaluated in double before rounding to fp32, as in torch's kernel
// (its betas / eps / lr are double arguments): with float constants (1 - 0.999f) is already off by 5e-5 relative.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n4, const float* __restrict__ lr,
                                                        const float* __restrict__ steps, double b1, double b2, double eps,
                                                        double wd) {
  __shared__ float s_c[2];
  if (threadIdx.x == 0) {
    const double t = (double)steps[0];
    const float bc1 = (float)(1.0 - pow(b1, t)), bc2 = (float)(1.0 - pow(b2, t));
    s_c[0] = (float)((double)lr[0] / (double)bc1);       // step size
    s_c[1] = sqrtf(bc2);
  }
  __syncthreads();
  const float step_size = s_c[0], bc2_sqrt = s_c[1];
  const double w1 = 1.0 - b1, w2 = 1.0 - b2;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gg = gv[j];
      if (wd != 0.0) gg = (float)((double)gg + (double)pv[j] * wd);
      mv[j] = (float)(b1 * (double)mv[j] + w1 * (double)gg);
      vv[j] = (float)(b2 * (double)vv[j] + w2 * (double)gg * (double)gg);
      const float denom = (float)((double)(sqrtf(vv[j]) / bc2_sqrt) + eps);
      pv[j] -= step_size * mv[j] / denom;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}

}  // namespace

// p, g, m, v: flat fp32 buffers of n floats (n % 4 == 0, 16-byte aligned).  Pad words between the parameter slots are updated
// like any other element and are don't-care: flatten_params zeroes them in p, flat_grads leaves them uninitialised in g, and
// whatever they hold (NaN included) stays confined to the pad words of p / m / v - no parameter element ever reads one.  steps: nsteps device floats (all incremented by one,
// steps[0] is the t of this update).  lr: device scalar.
SST_API int sst_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float* steps, int nsteps,
                          double beta1, double beta2, double eps, double weight_decay, void* stream) {
  SST_REQUIRE(p && g && m && v && lr && steps && n > 0 && (n & 3) == 0 && nsteps > 0, "sst_adam_flat: bad argument");
  hipStream_t st = sst_stream(stream);
  adam_tick_kernel<<<1, 256, 0, st>>>(steps, nsteps);
  SST_LAUNCH_CHECK("adam_tick_kernel");
  const int64_t n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  adam_flat_kernel<<<blocks, 256, 0, st>>>(p, g, m, v, n4, lr, steps, beta1, beta2, eps, weight_decay);
  SST_LAUNCH_CHECK("adam_flat_kernel");
  return SST_OK;
}

// Small layout / loss kernels around the generator and discriminator graphs (gfx950).
//
//   pixel (MSE / L1) criterion          reference config.py:88-90, warmup.py:88-93, train.py:138
//   BCE-with-logits criterion           reference config.py:71-73, train.py:59,113-114,135-136,155-161
//   clamp backward + NCHW<->NHWC        reference model.py:150 (clamp_), model.py:138-152 (NCHW surface)
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NT = 256;

// dst NHWC [B,H,W,C] <- src NCHW [B,C,H,W]   (dir 0)   or the inverse (dir 1).  Small C (3): plain gather.
// optional per-channel affine v*sc[c] + sh[c] (ImageNet normalisation of loss.py:52 and its backward)
__global__ __launch_bounds__(NT) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C,
                                                       int H, int W, int dir, const float* __restrict__ sc,
                                                       const float* __restrict__ sh) {
  const int64_t total = (int64_t)B * C * H * W, hw = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    int c;
    float v;
    if (dir == 0) {  // i indexes dst NHWC
      c = (int)(i % C);
      const int64_t p = i / C, b = p / hw, r = p - b * hw;
      v = src[(b * C + c) * hw + r];
    } else {         // i indexes dst NCHW
      const int64_t r = i % hw, bc = i / hw, b = bc / C;
      c = (int)(bc - b * C);
      v = src[(b * hw + r) * C + c];
    }
    if (sc) v = fmaf(v, sc[c], sh ? sh[c] : 0.f);
    dst[i] = v;
  }
}

// out[b,oy,ox,c] = max over the 2x2 window of relu(y[b,2oy+i,2ox+j,c])      (nn.ReLU + nn.MaxPool2d(2) of VGG19)
__global__ __launch_bounds__(NT) void maxpool_relu_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int H,
                                                              int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, c4n = C >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int c4 = (int)(i % c4n);
    int64_t p = i / c4n;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int64_t b = p / Ho;
    const f32x4* src = reinterpret_cast<const f32x4*>(y) + (((b * H + 2 * oy) * W + 2 * ox) * (int64_t)c4n + c4);
    f32x4 m = {0.f, 0.f, 0.f, 0.f};   // relu: max with 0
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 v = src[((k >> 1) * W + (k & 1)) * (int64_t)c4n];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
    }
    reinterpret_cast<f32x4*>(out)[i] = m;
  }
}

// dy[b,y,x,c] = g[b,y/2,x/2,c] if (y,x) is the first arg-max of relu(y) in its window and y > 0, else 0
__global__ __launch_bounds__(NT) void maxpool_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                              float* __restrict__ dy, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, c4n = C >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int c4 = (int)(i % c4n);
    int64_t p = i / c4n;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int64_t b = p / Ho;
    const int64_t base = ((b * H + 2 * oy) * W + 2 * ox) * (int64_t)c4n + c4;
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4*>(y)[base + ((k >> 1) * W + (k & 1)) * (int64_t)c4n];
    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int arg = 0;
      float m = fmaxf(v[0][j], 0.f);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float r = fmaxf(v[k][j], 0.f);
        if (r > m) { m = r; arg = k; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == arg && v[k][j] > 0.f) ? gv[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<f32x4*>(dy)[base + ((k >> 1) * W + (k & 1)) * (int64_t)c4n] = o[k];
  }
}

// g3[b,y,x,c] (NHWC) = (g[b,c,y,x] (+ g2)) * (0 <= pre <= 1)        clamp_ backward, model.py:150
// and per-block partial column sums (-> bias gradient of the producing conv):  partial[blk][c]
__global__ __launch_bounds__(NT) void clamp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pre,
                                                       float* __restrict__ out, float* __restrict__ partial, int B, int C,
                                                       int H, int W) {
  // column sums in fp64 (here and in colsum_finalize_kernel), as torch's CPU reductions accumulate them (acc_type<float> = double):
  // the bias gradient is a signed sum over every pixel of the batch; the cost is not measurable (the kernel is bound by its loads)
  __shared__ double red[NT / 64];
  const int64_t hw = (int64_t)H * W, npx = (int64_t)B * hw;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t p = blockIdx.x * (int64_t)NT + threadIdx.x; p < npx; p += (int64_t)gridDim.x * NT) {
    const int64_t b = p / hw, r = p - b * hw;
    for (int c = 0; c < C; ++c) {
      const int64_t s = (b * C + c) * hw + r;
      const float pv = pre[s];
      const float v = (pv >= 0.f && pv <= 1.f) ? g[s] : 0.f;
      out[p * C + c] = v;
      if (c < 4) acc[c] += (double)v;
    }
  }
  for (int c = 0; c < C && c < 4; ++c) {
    const double t = block_sum_d<NT>(acc[c], red);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * C + c] = (float)t;
  }
}

__global__ __launch_bounds__(NT) void colsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int nblk, int C, int accumulate) {
  __shared__ double red[NT / 64];
  const int c = blockIdx.x;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblk; b += NT) t += (double)partial[(size_t)b * C + c];
  t = block_sum_d<NT>(t, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)t : (float)t;
}

// ---- pixel criterion: mode 0 = MSE, 1 = L1.  Two-stage fixed-order reduction (last block finishes).
__global__ __launch_bounds__(NT) void pixel_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                            float* __restrict__ loss, float* __restrict__ partials,
                                                            unsigned* __restrict__ counter, int64_t n, int mode_) {
  __shared__ float red[NT / 64];
  const int mode = mode_ & 1;
  const bool relu = (mode_ & 2) != 0;   // criterion on relu(x), relu(gt): VGG feature taps (loss.py:66-68)
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i], b = reinterpret_cast<const f32x4*>(gt)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = relu ? fmaxf(a[j], 0.f) - fmaxf(b[j], 0.f) : a[j] - b[j];
      s += mode == 0 ? d * d : fabsf(d);
    }
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += NT) {
      const float d = relu ? fmaxf(x[i], 0.f) - fmaxf(gt[i], 0.f) : x[i] - gt[i];
      s += mode == 0 ? d * d : fabsf(d);
    }
  }
  s = block_sum<NT>(s, red);
  __shared__ unsigned s_flag;
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
  float tot;
  if (last_block_total<NT>(partials, counter, gridDim.x, &s_flag, red, tot) && threadIdx.x == 0) loss[0] = tot / (float)n;
}

// dx (+)= scale * d loss/dx ;  scale = scale_host * (scale_dev ? *scale_dev : 1)
__global__ __launch_bounds__(NT) void pixel_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                            float* __restrict__ dx, const float* __restrict__ scale_dev,
                                                            float scale_host, int accumulate, int64_t n, int mode_) {
  const int mode = mode_ & 1;
  const bool relu = (mode_ & 2) != 0;
  float sc = scale_host / (float)n;
  if (scale_dev) sc *= scale_dev[0];
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const float d = relu ? fmaxf(x[i], 0.f) - fmaxf(gt[i], 0.f) : x[i] - gt[i];
    float g = mode == 0 ? 2.f * d * sc : (d > 0.f ? sc : (d < 0.f ? -sc : 0.f));
    if (relu && !(x[i] > 0.f)) g = 0.f;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}

// ---- feature criterion on activated, per-channel-affine features (ContentLossDiscriminator, loss.py:231-289: taps after
// BatchNorm(eval) + LeakyReLU of the discriminator):  f(v) = act(v*scale[c] + shift[c]),  act(z) = z > 0 ? z : slope*z,
// loss = mean crit(f(x) - f(gt)) over [rows, C] tensors (mode 0 = MSE, 1 = L1); the taps hold the conv outputs v.
__global__ __launch_bounds__(NT) void feat_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float slope, int C, float* __restrict__ loss, float* __restrict__ partials,
                                                           unsigned* __restrict__ counter, int64_t n, int mode) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int c = (int)(i % C);
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    float a = fmaf(x[i], sc, sh), b = fmaf(gt[i], sc, sh);
    a = a > 0.f ? a : a * slope;
    b = b > 0.f ? b : b * slope;
    const float d = a - b;
    s += mode == 0 ? d * d : fabsf(d);
  }
  s = block_sum<NT>(s, red);
  __shared__ unsigned s_flag;
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
  float tot;
  if (last_block_total<NT>(partials, counter, gridDim.x, &s_flag, red, tot) && threadIdx.x == 0) loss[0] = tot / (float)n;
}

// dx (+)= scale_host * scale_dev * d(loss)/dx   (gradient w.r.t. the conv output x; gt is a constant)
__global__ __launch_bounds__(NT) void feat_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float slope, int C, float* __restrict__ dx, const float* __restrict__ scale_dev,
                                                           float scale_host, int accumulate, int64_t n, int mode) {
  float k = scale_host / (float)n;
  if (scale_dev) k *= scale_dev[0];
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int c = (int)(i % C);
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    const float za = fmaf(x[i], sc, sh), zb = fmaf(gt[i], sc, sh);
    const float a = za > 0.f ? za : za * slope, b = zb > 0.f ? zb : zb * slope;
    const float d = a - b;
    float g = mode == 0 ? 2.f * d * k : (d > 0.f ? k : (d < 0.f ? -k : 0.f));
    g *= (za > 0.f ? 1.f : slope) * sc;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}

// ---- BCEWithLogits(mean) against a constant target t:  l = max(x,0) - x t + log1p(exp(-|x|))
// single workgroup (logits are [B,1]); dlogit = scale * (sigmoid(x) - t) / n
__global__ __launch_bounds__(NT) void bce_fwd_bwd_kernel(const float* __restrict__ x, float target, float* __restrict__ loss,
                                                         float* __restrict__ dx, const float* __restrict__ scale_dev,
                                                         float scale_host, int n) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  float sc = scale_host / (float)n;
  if (scale_dev) sc *= scale_dev[0];
  for (int i = threadIdx.x; i < n; i += NT) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    if (dx) dx[i] = sc * (1.f / (1.f + expf(-v)) - target);
  }
  s = block_sum<NT>(s, red);
  if (threadIdx.x == 0 && loss) loss[0] = s / (float)n;
}

// out = sum_i w[i] * terms[i][0]     (weighted sum of up to 8 scalar losses; also copies each weighted term)
struct ScalarPtrs { const float* p[8]; float w[8]; };
__global__ void weighted_sum_kernel(ScalarPtrs sp, int n, float* __restrict__ out, float* __restrict__ weighted) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < n; ++i) {
      const float v = sp.p[i][0] * sp.w[i];
      if (weighted) weighted[i] = v;
      t += v;
    }
    out[0] = t;
  }
}

inline int grid_for(int64_t items) {
  int64_t b = (items + NT - 1) / NT;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_transpose(const float* src, float* dst, int B, int C, int H, int W, int to_nchw, void* stream) {
  SST_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "sst_transpose: bad argument");
  transpose_kernel<<<grid_for((int64_t)B * C * H * W), NT, 0, sst_stream(stream)>>>(src, dst, B, C, H, W, to_nchw, nullptr, nullptr);
  SST_LAUNCH_CHECK("transpose_kernel");
  return SST_OK;
}

// NCHW <-> NHWC with a per-channel affine v*scale[c] + shift[c] (shift may be null)
SST_API int sst_transpose_affine(const float* src, float* dst, int B, int C, int H, int W, int to_nchw, const float* scale,
                                 const float* shift, void* stream) {
  SST_REQUIRE(src && dst && scale && B > 0 && C > 0 && H > 0 && W > 0, "sst_transpose_affine: bad argument");
  transpose_kernel<<<grid_for((int64_t)B * C * H * W), NT, 0, sst_stream(stream)>>>(src, dst, B, C, H, W, to_nchw, scale, shift);
  SST_LAUNCH_CHECK("transpose_kernel");
  return SST_OK;
}

// ReLU + MaxPool2d(2) of the VGG19 feature stack (loss.py:46-49), NHWC, C % 4 == 0, H and W even
SST_API int sst_maxpool_relu_fwd(const float* y, float* out, int B, int H, int W, int C, void* stream) {
  SST_REQUIRE(y && out && B > 0 && H > 1 && W > 1 && !(H & 1) && !(W & 1) && C > 0 && !(C & 3), "sst_maxpool_relu_fwd: bad argument");
  maxpool_relu_fwd_kernel<<<grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4)), NT, 0, sst_stream(stream)>>>(y, out, B, H, W, C);
  SST_LAUNCH_CHECK("maxpool_relu_fwd_kernel");
  return SST_OK;
}

SST_API int sst_maxpool_relu_bwd(const float* g, const float* y, float* dy, int B, int H, int W, int C, void* stream) {
  SST_REQUIRE(g && y && dy && B > 0 && H > 1 && W > 1 && !(H & 1) && !(W & 1) && C > 0 && !(C & 3), "sst_maxpool_relu_bwd: bad argument");
  maxpool_relu_bwd_kernel<<<grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4)), NT, 0, sst_stream(stream)>>>(g, y, dy, B, H, W, C);
  SST_LAUNCH_CHECK("maxpool_relu_bwd_kernel");
  return SST_OK;
}

SST_API int sst_clamp_bwd_blocks(int B, int H, int W) { return grid_for((int64_t)B * H * W); }

// g, pre NCHW [B,C,H,W] (C <= 4) -> out NHWC [B,H,W,C]; dbias[C] (+)= column sums.  partial: blocks*C floats.
SST_API int sst_clamp_bwd(const float* g, const float* pre, float* out, float* partial, float* dbias, int accumulate, int B,
                          int C, int H, int W, void* stream) {
  SST_REQUIRE(g && pre && out && partial && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0, "sst_clamp_bwd: bad argument");
  const int nb = sst_clamp_bwd_blocks(B, H, W);
  clamp_bwd_kernel<<<nb, NT, 0, sst_stream(stream)>>>(g, pre, out, partial, B, C, H, W);
  SST_LAUNCH_CHECK("clamp_bwd_kernel");
  if (dbias) {
    colsum_finalize_kernel<<<C, NT, 0, sst_stream(stream)>>>(partial, dbias, nb, C, accumulate);
    SST_LAUNCH_CHECK("colsum_finalize_kernel");
  }
  return SST_OK;
}

SST_API int sst_pixel_loss_blocks(int64_t n) {
  const int b = grid_for(n / 4 + 1);
  return b > 256 ? 256 : b;
}

SST_API int sst_pixel_loss_fwd(const float* x, const float* gt, float* loss, float* partials, unsigned* counter, int64_t n,
                               int mode, void* stream) {
  SST_REQUIRE(x && gt && loss && partials && counter && n > 0 && mode >= 0 && mode <= 3, "sst_pixel_loss_fwd: bad argument");
  pixel_loss_fwd_kernel<<<sst_pixel_loss_blocks(n), NT, 0, sst_stream(stream)>>>(x, gt, loss, partials, counter, n, mode);
  SST_LAUNCH_CHECK("pixel_loss_fwd_kernel");
  return SST_OK;
}

SST_API int sst_pixel_loss_bwd(const float* x, const float* gt, float* dx, const float* scale_dev, float scale_host,
                               int accumulate, int64_t n, int mode, void* stream) {
  SST_REQUIRE(x && gt && dx && n > 0 && mode >= 0 && mode <= 3, "sst_pixel_loss_bwd: bad argument");
  pixel_loss_bwd_kernel<<<grid_for(n), NT, 0, sst_stream(stream)>>>(x, gt, dx, scale_dev, scale_host, accumulate, n, mode);
  SST_LAUNCH_CHECK("pixel_loss_bwd_kernel");
  return SST_OK;
}

SST_API int sst_feat_loss_fwd(const float* x, const float* gt, const float* scale, const float* shift, float slope, int C, float* loss,
                              float* partials, unsigned* counter, int64_t n, int mode, void* stream) {
  SST_REQUIRE(x && gt && loss && partials && counter && n > 0 && C > 0 && n % C == 0 && (mode == 0 || mode == 1) &&
                  ((scale == nullptr) == (shift == nullptr)), "sst_feat_loss_fwd: bad argument");
  feat_loss_fwd_kernel<<<sst_pixel_loss_blocks(n), NT, 0, sst_stream(stream)>>>(x, gt, scale, shift, slope, C, loss, partials, counter, n,
                                                                                 mode);
  SST_LAUNCH_CHECK("feat_loss_fwd_kernel");
  return SST_OK;
}

SST_API int sst_feat_loss_bwd(const float* x, const float* gt, const float* scale, const float* shift, float slope, int C, float* dx,
                              const float* scale_dev, float scale_host, int accumulate, int64_t n, int mode, void* stream) {
  SST_REQUIRE(x && gt && dx && n > 0 && C > 0 && n % C == 0 && (mode == 0 || mode == 1) && ((scale == nullptr) == (shift == nullptr)),
              "sst_feat_loss_bwd: bad argument");
  feat_loss_bwd_kernel<<<grid_for(n), NT, 0, sst_stream(stream)>>>(x, gt, scale, shift, slope, C, dx, scale_dev, scale_host, accumulate, n,
                                                                   mode);
  SST_LAUNCH_CHECK("feat_loss_bwd_kernel");
  return SST_OK;
}

SST_API int sst_bce_logits(const float* logits, float target, float* loss, float* dlogits, const float* scale_dev,
                           float scale_host, int n, void* stream) {
  SST_REQUIRE(logits && n > 0 && (loss || dlogits), "sst_bce_logits: bad argument");
  bce_fwd_bwd_kernel<<<1, NT, 0, sst_stream(stream)>>>(logits, target, loss, dlogits, scale_dev, scale_host, n);
  SST_LAUNCH_CHECK("bce_fwd_bwd_kernel");
  return SST_OK;
}

// ---- MATLAB-style antialiased bicubic resampling (reference bicubic.py:15-105, used by dataset.py:28 to synthesise the LR
// input).  Separable gather-multiply-sum with host-built tap tables (weights normalised, border indices clamped):
//   V[oy][x] = sum_ty wy[oy][ty] * in[iy[oy][ty]][x];   out[oy][ox] = round(255 * sum_tx wx[ox][tx] * V[oy][ix[ox][tx]]) / 255
// (vertical pass first, taps in table order - the reference's order; rounding to the 1/255 grid, no clamp).
// One thread per output pixel; the input plane (36.9 KB at 96x96) is L1/L2-resident, the op is launch/HBM-bound.
namespace {
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ wy,
                                                      const int* __restrict__ iy, const float* __restrict__ wx,
                                                      const int* __restrict__ ix, int64_t planes, int H, int W, int oh, int ow, int Ty,
                                                      int Tx, int round_grid) {
  const int64_t total = planes * oh * ow;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const int64_t t = i / ow;
    const int oy = (int)(t % oh);
    const float* src = x + (t / oh) * (int64_t)H * W;
    float acc = 0.f;
    for (int tx = 0; tx < Tx; ++tx) {
      const int cx = ix[ox * Tx + tx];
      float v = 0.f;
      for (int ty = 0; ty < Ty; ++ty) v += src[(int64_t)iy[oy * Ty + ty] * W + cx] * wy[oy * Ty + ty];
      acc += v * wx[ox * Tx + tx];
    }
    y[i] = round_grid ? rintf(255.f * acc) / 255.f : acc;
  }
}
}  // namespace

// x [planes, H, W] -> y [planes, oh, ow]; wy/iy [oh, Ty], wx/ix [ow, Tx] (fp32 weights, int32 0-based indices, device memory).
SST_API int sst_bicubic(const float* x, float* y, const float* wy, const int* iy, const float* wx, const int* ix, int64_t planes,
                        int H, int W, int oh, int ow, int Ty, int Tx, int round_grid, void* stream) {
  SST_REQUIRE(x && y && wy && iy && wx && ix && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && Ty > 0 && Tx > 0,
              "sst_bicubic: bad argument");
  const int64_t total = planes * oh * ow;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  bicubic_kernel<<<blocks, 256, 0, sst_stream(stream)>>>(x, y, wy, iy, wx, ix, planes, H, W, oh, ow, Ty, Tx, round_grid);
  SST_LAUNCH_CHECK("bicubic_kernel");
  return SST_OK;
}

SST_API int sst_weighted_sum(const float* const* terms, const float* weights, int n, float* out, float* weighted,
                             void* stream) {
  SST_REQUIRE(terms && weights && out && n > 0 && n <= 8, "sst_weighted_sum: 1..8 terms");
  ScalarPtrs sp;
  for (int i = 0; i < n; ++i) {
    SST_REQUIRE(terms[i], "sst_weighted_sum: null term");
    sp.p[i] = terms[i];
    sp.w[i] = weights[i];
  }
  weighted_sum_kernel<<<1, 64, 0, sst_stream(stream)>>>(sp, n, out, weighted);
  SST_LAUNCH_CHECK("weighted_sum_kernel");
  return SST_OK;
}

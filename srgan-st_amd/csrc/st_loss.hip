// Structure-tensor loss, forward and backward, for gfx950.
//
// Replaces (reference file:line)
//   StructureTensorLoss.forward / st_loss          loss.py:399-413
//   structure_tensor                                utils.py:212-233   (10 separable 'same' zero-padded correlations)
//   get_gaussian_kernel                             utils.py:194-208   (taps computed on the host, passed by value)
//   normalize / compute_invS1xS2 / compute_eigenvalues / compute_distance   utils.py:236-280
//   torchvision Grayscale                           loss.py:400-401    (0.2989 R + 0.587 G + 0.114 B)
//
// Layout: sr, gt, dsr are NCHW fp32 (the Python surface layout, model.py:138-152 returns NCHW);
// gS (saved d loss / d(Jxx,Jyy,Jxy) of the SR image, unit upstream gradient) is planar [B,3,H,W].
//
// One workgroup (256 threads) per 32x32 output tile; every separable pass is staged through LDS
// with the halo it needs (forward: 2+8 px, backward: 8+2+2 px).  HBM-bound by construction:
// algorithmic bytes = read sr + read gt + write d(sr) = 3*3*H*W*4 B per image.
#include "common.h"

namespace {

constexpr int T = 32;     // output tile edge
constexpr int NT = 1024;  // threads per workgroup (measured fwd+bwd at 96 px, B = 16: 256 -> 50 us, 512 -> 35 us, 1024 -> 29 us: the passes are latency-bound and there are only 144 tiles)
constexpr int PPT = T * T / NT;   // output pixels per thread

template <int R1, int R2>
struct StTaps {
  float g[2 * R1 + 1];
  float dg[2 * R1 + 1];
  float k[2 * R2 + 1];
};

__device__ __forceinline__ float gray_at(const float* __restrict__ img, int H, int W, int y, int x) {
  if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0.f;
  const size_t hw = (size_t)H * W;
  const float* p = img + (size_t)y * W + x;
  return 0.2989f * p[0] + 0.587f * p[hw] + 0.114f * p[2 * hw];
}

// ---------------------------------------------------------------------------------------------
// Computes the structure tensor (Jxx,Jyy,Jxy) of one image on the tile; 4 pixels per thread.
// LDS use (floats): GW*GW + 2*IW*GW + 2*IW*IW, with the 17-tap H-pass output overlaying the
// (dead by then) gray/A buffers.
template <int R1, int R2>
__device__ __forceinline__ void tile_structure_tensor(const float* __restrict__ img, int H, int W, int y0, int x0,
                                                      const StTaps<R1, R2>& tp, float* lds, float (&J)[PPT][3]) {
  constexpr int R = R1 + R2;
  constexpr int GW = T + 2 * R;    // gray patch edge
  constexpr int IW = T + 2 * R2;   // Ix/Iy region edge
  float* sG = lds;
  float* sA1 = sG + GW * GW;
  float* sA2 = sA1 + IW * GW;
  float* sIx = sA2 + IW * GW;
  float* sIy = sIx + IW * IW;
  float* sQ = lds;  // overlays sG/sA1/sA2: 3*T*IW <= GW*GW + 2*IW*GW
  static_assert(3 * T * IW <= GW * GW + 2 * IW * GW, "overlay");
  const int tid = threadIdx.x;

  __syncthreads();  // previous user of the LDS is done
  for (int i = tid; i < GW * GW; i += NT) {
    const int pr = i / GW, pc = i - pr * GW;
    sG[i] = gray_at(img, H, W, y0 - R + pr, x0 - R + pc);
  }
  __syncthreads();
  // 5-tap pass along H: A1 = dg (x)_H gray, A2 = g (x)_H gray                      utils.py:219,221
  for (int i = tid; i < IW * GW; i += NT) {
    const int ar = i / GW, pc = i - ar * GW;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      const float v = sG[(ar + t) * GW + pc];
      a1 = fmaf(tp.dg[t], v, a1);
      a2 = fmaf(tp.g[t], v, a2);
    }
    sA1[i] = a1;
    sA2[i] = a2;
  }
  __syncthreads();
  // 5-tap pass along W: Ix = g (x)_W A1, Iy = dg (x)_W A2; zero outside the image      utils.py:220,222
  for (int i = tid; i < IW * IW; i += NT) {
    const int ar = i / IW, ac = i - ar * IW;
    float ix = 0.f, iy = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      ix = fmaf(tp.g[t], sA1[ar * GW + ac + t], ix);
      iy = fmaf(tp.dg[t], sA2[ar * GW + ac + t], iy);
    }
    const int y = y0 - R2 + ar, x = x0 - R2 + ac;
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    sIx[i] = in ? ix : 0.f;
    sIy[i] = in ? iy : 0.f;
  }
  __syncthreads();
  // (2R2+1)-tap pass along H on the products                                          utils.py:225,227,229
  for (int i = tid; i < T * IW; i += NT) {
    const int qr = i / IW, ac = i - qr * IW;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R2; ++t) {
      const float ix = sIx[(qr + t) * IW + ac], iy = sIy[(qr + t) * IW + ac];
      q0 = fmaf(tp.k[t], ix * ix, q0);
      q1 = fmaf(tp.k[t], iy * iy, q1);
      q2 = fmaf(tp.k[t], ix * iy, q2);
    }
    sQ[i] = q0;
    sQ[T * IW + i] = q1;
    sQ[2 * T * IW + i] = q2;
  }
  __syncthreads();
  // pass along W                                                                     utils.py:226,228,230
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + j * NT, qr = p >> 5, qc = p & 31;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R2; ++t) {
      const float kk = tp.k[t];
      s0 = fmaf(kk, sQ[qr * IW + qc + t], s0);
      s1 = fmaf(kk, sQ[T * IW + qr * IW + qc + t], s1);
      s2 = fmaf(kk, sQ[2 * T * IW + qr * IW + qc + t], s2);
    }
    J[j][0] = s0;
    J[j][1] = s1;
    J[j][2] = s2;
  }
}

template <int R1, int R2>
constexpr int st_fwd_lds_floats() {
  constexpr int R = R1 + R2, GW = T + 2 * R, IW = T + 2 * R2;
  return GW * GW + 2 * IW * GW + 2 * IW * IW;
}

// ---------------------------------------------------------------------------------------------
// Pixel criterion riding along (reference train.py:129-140 evaluates "Pixel" and "ST" on the same sr / gt): mode < 0 none, 0 MSE,
// 1 L1.  The tile's own pixels are re-read per channel (cache hits: the gray patch was just built from them); the partial sums
// share the structure-tensor term's last-block ticket.
struct StPix { float* loss; float* partials; int mode; };

template <int R1, int R2>
__global__ __launch_bounds__(NT) void st_loss_fwd_kernel(const float* __restrict__ sr, const float* __restrict__ gt,
                                                         float* __restrict__ gS, float* __restrict__ loss,
                                                         float* __restrict__ partials, unsigned* __restrict__ counter,
                                                         int B, int H, int W, int normalize, StTaps<R1, R2> tp, StPix pix) {
  __shared__ float lds[st_fwd_lds_floats<R1, R2>()];
  __shared__ float red[NT / 64];
  const int b = blockIdx.z, y0 = blockIdx.y * T, x0 = blockIdx.x * T;
  const size_t img_off = (size_t)b * 3 * H * W;
  float J1[PPT][3], J2[PPT][3];
  tile_structure_tensor<R1, R2>(sr + img_off, H, W, y0, x0, tp, lds, J1);
  tile_structure_tensor<R1, R2>(gt + img_off, H, W, y0, x0, tp, lds, J2);

  const float eps = 1e-12f;
  float lsum = 0.f, psum = 0.f;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = threadIdx.x + j * NT, y = y0 + (p >> 5), x = x0 + (p & 31);
    if (y >= H || x >= W) continue;
    if (pix.mode >= 0) {
      const size_t o = img_off + (size_t)y * W + x, hw = (size_t)H * W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = sr[o + c * hw] - gt[o + c * hw];
        psum += pix.mode == 0 ? d * d : fabsf(d);
      }
    }
    const float a1 = J1[j][0], b1 = J1[j][1], c1 = J1[j][2];
    const float a2 = J2[j][0], b2 = J2[j][1], c2 = J2[j][2];
    // normalize                                                             utils.py:236-239
    float n1 = 1.f, n2 = 1.f;
    if (normalize) {
      n1 = 1.f / sqrtf(a1 * b1 - c1 * c1 + eps);
      n2 = 1.f / sqrtf(a2 * b2 - c2 * c2 + eps);
    }
    const float ah1 = a1 * n1, bh1 = b1 * n1, ch1 = c1 * n1;
    const float ah2 = a2 * n2, bh2 = b2 * n2, ch2 = c2 * n2;
    // inv(S1)*S2                                                            utils.py:248-251
    const float A = bh1 * ah2 - ch1 * ch2;
    const float Bm = ah1 * bh2 - ch1 * ch2;
    const float C = bh1 * ch2 - ch1 * bh2;
    const float D = ah1 * ch2 - ch1 * ah2;
    // eigenvalues                                                           utils.py:260-265
    const float ApB = A + Bm;
    const float disc = ApB * ApB - 4.f * (A * Bm - C * D);
    const float discc = fmaxf(disc, eps);
    const float r = sqrtf(discc);
    const float l1 = 0.5f * (ApB - r), l2 = 0.5f * (ApB + r);
    // distance                                                              utils.py:275-280
    const float L1 = fmaxf(l1, 1.f), L2 = fmaxf(l2, 1.f);
    const float g1 = logf(L1), g2 = logf(L2);
    const float d = sqrtf(g1 * g1 + g2 * g2 + eps);
    lsum += d;

    // ---- analytic gradient of d wrt (a1,b1,c1), unit upstream
    const float dq = 0.5f / d;
    const float dl1 = (l1 >= 1.f) ? dq * 2.f * g1 / L1 : 0.f;
    const float dl2 = (l2 >= 1.f) ? dq * 2.f * g2 / L2 : 0.f;
    float dApB = 0.5f * (dl1 + dl2);
    const float dr = 0.5f * (dl2 - dl1);
    const float ddisc = (disc >= eps) ? dr / (2.f * r) : 0.f;
    dApB += ddisc * 2.f * ApB;
    const float dt = -4.f * ddisc;  // d wrt (A*Bm - C*D)
    const float dA = dApB + dt * Bm;
    const float dB = dApB + dt * A;
    const float dC = -dt * D;
    const float dD = -dt * C;
    const float dah = dB * bh2 + dD * ch2;
    const float dbh = dA * ah2 + dC * ch2;
    const float dch = -(dA + dB) * ch2 - dC * bh2 - dD * ah2;
    float ga = dah * n1, gb = dbh * n1, gc = dch * n1;
    if (normalize) {
      const float dn = a1 * dah + b1 * dbh + c1 * dch;
      const float ddet = dn * (-0.5f * n1 * n1 * n1);
      ga += ddet * b1;
      gb += ddet * a1;
      gc -= 2.f * ddet * c1;
    }
    const size_t o = img_off + (size_t)y * W + x, hw = (size_t)H * W;
    gS[o] = ga;
    gS[o + hw] = gb;
    gS[o + 2 * hw] = gc;
  }
  const float bsum = block_sum<NT>(lsum, red);
  __shared__ unsigned s_flag;
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const unsigned me = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  float pb = 0.f;
  if (pix.mode >= 0) pb = block_sum<NT>(psum, red);
  if (threadIdx.x == 0) {
    partials[me] = bsum;
    if (pix.mode >= 0) pix.partials[me] = pb;
  }
  float tot;
  if (last_block_total<NT>(partials, counter, nblk, &s_flag, red, tot)) {
    if (threadIdx.x == 0) loss[0] = tot / ((float)B * (float)H * (float)W);
    if (pix.mode >= 0) {                                   // same ticket: every workgroup's stores are visible here
      float t2 = 0.f;
      for (unsigned i = threadIdx.x; i < nblk; i += NT) t2 += load_agent(pix.partials + i);
      t2 = block_sum<NT>(t2, red);
      if (threadIdx.x == 0) pix.loss[0] = t2 / (3.f * (float)B * (float)H * (float)W);
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int R1, int R2>
constexpr int st_bwd_lds_floats() {
  constexpr int R = R1 + R2, EW = T + 2 * R1, SW = T + 2 * R;
  return 3 * SW * SW + 3 * EW * SW;
}

// d(sr) (+)= gray_w[c] * scale * adjoint(structure tensor)(gS).   scale = scale_host * (scale_dev ? *scale_dev : 1)
// pix_gt != null: dsr also receives pix_scale * (scale_dev ? *scale_dev : 1) * d pixel_criterion / d sr (pix_scale = weight / numel)
template <int R1, int R2>
__global__ __launch_bounds__(NT) void st_loss_bwd_kernel(const float* __restrict__ sr, const float* __restrict__ gS,
                                                         float* __restrict__ dsr, const float* __restrict__ scale_dev,
                                                         float scale_host, int accumulate, int B, int H, int W,
                                                         StTaps<R1, R2> tp, const float* __restrict__ pix_gt, float pix_scale,
                                                         int pix_mode) {
  constexpr int R = R1 + R2;
  constexpr int EW = T + 2 * R1;    // region where dIx,dIy are needed
  constexpr int SW = T + 2 * R;     // gS patch edge
  constexpr int GW2 = T + 4 * R1;   // gray patch edge for recomputing Ix,Iy on the EW region
  __shared__ float lds[st_bwd_lds_floats<R1, R2>()];
  float* sS = lds;                       // region X: [3][SW][SW]   gS patch
  float* sU = sS + 3 * SW * SW;          // region Y: [3][EW][SW]   after the H pass
  float* sV = sS;                        // X again : [3][EW][EW]   after the W pass (gS patch is dead)
  float* sG = sU;                        // Y again : [GW2][GW2]    gray(sr) patch (U is dead)
  float* sA1 = sG + GW2 * GW2;           //           [EW][GW2]
  float* sA2 = sA1 + EW * GW2;           //           [EW][GW2]
  static_assert(GW2 * GW2 + 2 * EW * GW2 <= 3 * EW * SW, "overlay Y");
  float* sDIx = sV;                      // in place over V plane 0 / plane 1
  float* sDIy = sV + EW * EW;
  float* sDA1 = sU;                      // Y again : [EW][T] x2 (gray/A are dead)
  float* sDA2 = sDA1 + EW * T;
  static_assert(2 * EW * T <= 3 * EW * SW, "overlay Y2");

  const int tid = threadIdx.x;
  const int b = blockIdx.z, y0 = blockIdx.y * T, x0 = blockIdx.x * T;
  const size_t hw = (size_t)H * W, img_off = (size_t)b * 3 * hw;

  for (int i = tid; i < 3 * SW * SW; i += NT) {
    const int c = i / (SW * SW), rem = i - c * SW * SW, pr = rem / SW, pc = rem - pr * SW;
    const int y = y0 - R + pr, x = x0 - R + pc;
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    sS[i] = in ? gS[img_off + c * hw + (size_t)y * W + x] : 0.f;
  }
  __syncthreads();
  // adjoint of the integration passes (k symmetric): along H ...
  for (int i = tid; i < 3 * EW * SW; i += NT) {
    const int c = i / (EW * SW), rem = i - c * EW * SW, er = rem / SW, sc = rem - er * SW;
    float u = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R2; ++t) u = fmaf(tp.k[t], sS[c * SW * SW + (er + t) * SW + sc], u);
    sU[i] = u;
  }
  __syncthreads();
  // ... along W (into sV, overlaying sS); then the gray patch of sr into the dead U region
  for (int i = tid; i < 3 * EW * EW; i += NT) {
    const int c = i / (EW * EW), rem = i - c * EW * EW, er = rem / EW, ec = rem - er * EW;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R2; ++t) v = fmaf(tp.k[t], sU[c * EW * SW + er * SW + ec + t], v);
    sV[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < GW2 * GW2; i += NT) {
    const int pr = i / GW2, pc = i - pr * GW2;
    sG[i] = gray_at(sr + img_off, H, W, y0 - 2 * R1 + pr, x0 - 2 * R1 + pc);
  }
  __syncthreads();
  for (int i = tid; i < EW * GW2; i += NT) {
    const int er = i / GW2, gc = i - er * GW2;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      const float v = sG[(er + t) * GW2 + gc];
      a1 = fmaf(tp.dg[t], v, a1);
      a2 = fmaf(tp.g[t], v, a2);
    }
    sA1[i] = a1;
    sA2[i] = a2;
  }
  __syncthreads();
  // Ix, Iy on the EW region -> dIx = 2 Ix gJxx + Iy gJxy, dIy = 2 Iy gJyy + Ix gJxy
  for (int i = tid; i < EW * EW; i += NT) {
    const int er = i / EW, ec = i - er * EW;
    float ix = 0.f, iy = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      ix = fmaf(tp.g[t], sA1[er * GW2 + ec + t], ix);
      iy = fmaf(tp.dg[t], sA2[er * GW2 + ec + t], iy);
    }
    const int y = y0 - R1 + er, x = x0 - R1 + ec;
    const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const float v0 = sV[i], v1 = sV[EW * EW + i], v2 = sV[2 * EW * EW + i];
    sDIx[i] = in ? (2.f * ix * v0 + iy * v2) : 0.f;
    sDIy[i] = in ? (2.f * iy * v1 + ix * v2) : 0.f;
  }
  __syncthreads();
  // adjoint of the W passes: dA1 = g (x)_W dIx ; dA2 = flip(dg) (x)_W dIy = -(dg (x)_W dIy)
  for (int i = tid; i < EW * T; i += NT) {
    const int er = i / T, oc = i - er * T;
    float d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      d1 = fmaf(tp.g[t], sDIx[er * EW + oc + t], d1);
      d2 = fmaf(-tp.dg[t], sDIy[er * EW + oc + t], d2);
    }
    sDA1[i] = d1;
    sDA2[i] = d2;
  }
  __syncthreads();
  float scale = scale_host;
  if (scale_dev) scale *= scale_dev[0];
  // adjoint of the H passes: dG = flip(dg) (x)_H dA1 + g (x)_H dA2
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + j * NT, orow = p >> 5, oc = p & 31, y = y0 + orow, x = x0 + oc;
    float dg_ = 0.f;
#pragma unroll
    for (int t = 0; t <= 2 * R1; ++t) {
      dg_ = fmaf(-tp.dg[t], sDA1[(orow + t) * T + oc], dg_);
      dg_ = fmaf(tp.g[t], sDA2[(orow + t) * T + oc], dg_);
    }
    if (y < H && x < W) {
      const size_t o = img_off + (size_t)y * W + x;
      const float v = dg_ * scale;
      float pv[3] = {0.f, 0.f, 0.f};
      if (pix_gt) {
        float psc = pix_scale;
        if (scale_dev) psc *= scale_dev[0];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = sr[o + c * hw] - pix_gt[o + c * hw];
          pv[c] = pix_mode == 0 ? 2.f * d * psc : (d > 0.f ? psc : (d < 0.f ? -psc : 0.f));
        }
      }
      if (accumulate) {
        dsr[o] += 0.2989f * v + pv[0];
        dsr[o + hw] += 0.587f * v + pv[1];
        dsr[o + 2 * hw] += 0.114f * v + pv[2];
      } else {
        dsr[o] = 0.2989f * v + pv[0];
        dsr[o + hw] = 0.587f * v + pv[1];
        dsr[o + 2 * hw] = 0.114f * v + pv[2];
      }
    }
  }
}

// host: taps exactly like utils.py:194-208 (fp32 exp, fp32 normalisation)
template <int R>
void gaussian_taps(float sigma, float* g, float* dg) {
  const float sigma2 = (float)((double)sigma * (double)sigma + 1e-12);
  const float c = (float)(-0.5 / ((double)sigma * (double)sigma + 1e-12));
  float s = 0.f;
  for (int i = 0; i <= 2 * R; ++i) {
    const float x = (float)(i - R);
    g[i] = expf(c * (x * x));
    s += g[i];
  }
  for (int i = 0; i <= 2 * R; ++i) {
    g[i] /= s;
    if (dg) dg[i] = g[i] * -(float)(i - R) / sigma2;
  }
}

int radius_of(float s) {
  int r = (int)(4.0 * (double)s + 0.5);
  return r < 1 ? 1 : r;
}

template <int R1, int R2>
StTaps<R1, R2> make_taps(float sigma, float rho) {
  StTaps<R1, R2> tp;
  gaussian_taps<R1>(sigma, tp.g, tp.dg);
  gaussian_taps<R2>(rho, tp.k, nullptr);
  return tp;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_st_loss_workspace(int B, int H, int W, int64_t* partial_floats) {
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && partial_floats, "sst_st_loss_workspace: bad shape");
  *partial_floats = (int64_t)B * ((H + T - 1) / T) * ((W + T - 1) / T);
  return SST_OK;
}

static int st_loss_fwd_impl(const float* sr, const float* gt, float* loss, float* gS, float* partials, unsigned* counter, int B, int H,
                            int W, float sigma, float rho, int normalize, StPix pix, void* stream) {
  SST_REQUIRE(sr && gt && loss && gS && partials && counter, "sst_st_loss_fwd: null pointer");
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535, "sst_st_loss_fwd: bad shape B=%d H=%d W=%d", B, H, W);
  const int r1 = radius_of(sigma), r2 = radius_of(rho);
  dim3 grid((W + T - 1) / T, (H + T - 1) / T, B);
  if (r1 == 2 && r2 == 8) {
    st_loss_fwd_kernel<2, 8><<<grid, NT, 0, sst_stream(stream)>>>(sr, gt, gS, loss, partials, counter, B, H, W,
                                                                    normalize, make_taps<2, 8>(sigma, rho), pix);
  } else if (r1 == 4 && r2 == 10) {
    st_loss_fwd_kernel<4, 10><<<grid, NT, 0, sst_stream(stream)>>>(sr, gt, gS, loss, partials, counter, B, H, W,
                                                                     normalize, make_taps<4, 10>(sigma, rho), pix);
  } else {
    return sst_set_error(SST_ERR_UNSUPPORTED, "sst_st_loss_fwd: (sigma,rho)=(%g,%g) -> radii (%d,%d) not built", sigma,
                         rho, r1, r2);
  }
  SST_LAUNCH_CHECK("st_loss_fwd_kernel");
  return SST_OK;
}

SST_API int sst_st_loss_fwd(const float* sr, const float* gt, float* loss, float* gS, float* partials,
                            unsigned* counter, int B, int H, int W, float sigma, float rho, int normalize,
                            void* stream) {
  return st_loss_fwd_impl(sr, gt, loss, gS, partials, counter, B, H, W, sigma, rho, normalize, StPix{nullptr, nullptr, -1}, stream);
}

// Structure-tensor loss + pixel criterion (pix_mode 0 = MSE, 1 = L1: reference config.py:88-90) in ONE launch: pix_loss[0] = the
// criterion's mean over all B*3*H*W elements; pix_partials = as many floats as `partials`.
SST_API int sst_st_pixel_loss_fwd(const float* sr, const float* gt, float* loss, float* gS, float* partials, unsigned* counter,
                                  float* pix_loss, float* pix_partials, int pix_mode, int B, int H, int W, float sigma, float rho,
                                  int normalize, void* stream) {
  SST_REQUIRE(pix_loss && pix_partials && (pix_mode == 0 || pix_mode == 1), "sst_st_pixel_loss_fwd: bad pixel-criterion argument");
  return st_loss_fwd_impl(sr, gt, loss, gS, partials, counter, B, H, W, sigma, rho, normalize, StPix{pix_loss, pix_partials, pix_mode},
                          stream);
}

static int st_loss_bwd_impl(const float* sr, const float* gS, float* dsr, const float* scale_dev, float scale_host, int accumulate, int B,
                            int H, int W, float sigma, float rho, const float* pix_gt, float pix_scale, int pix_mode, void* stream) {
  SST_REQUIRE(sr && gS && dsr, "sst_st_loss_bwd: null pointer");
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535, "sst_st_loss_bwd: bad shape");
  const int r1 = radius_of(sigma), r2 = radius_of(rho);
  dim3 grid((W + T - 1) / T, (H + T - 1) / T, B);
  const float s = scale_host / ((float)B * (float)H * (float)W);
  if (r1 == 2 && r2 == 8) {
    st_loss_bwd_kernel<2, 8><<<grid, NT, 0, sst_stream(stream)>>>(sr, gS, dsr, scale_dev, s, accumulate, B, H, W,
                                                                    make_taps<2, 8>(sigma, rho), pix_gt, pix_scale, pix_mode);
  } else if (r1 == 4 && r2 == 10) {
    st_loss_bwd_kernel<4, 10><<<grid, NT, 0, sst_stream(stream)>>>(sr, gS, dsr, scale_dev, s, accumulate, B, H, W,
                                                                     make_taps<4, 10>(sigma, rho), pix_gt, pix_scale, pix_mode);
  } else {
    return sst_set_error(SST_ERR_UNSUPPORTED, "sst_st_loss_bwd: (sigma,rho)=(%g,%g) not built", sigma, rho);
  }
  SST_LAUNCH_CHECK("st_loss_bwd_kernel");
  return SST_OK;
}

SST_API int sst_st_loss_bwd(const float* sr, const float* gS, float* dsr, const float* scale_dev, float scale_host,
                            int accumulate, int B, int H, int W, float sigma, float rho, void* stream) {
  return st_loss_bwd_impl(sr, gS, dsr, scale_dev, scale_host, accumulate, B, H, W, sigma, rho, nullptr, 0.f, 0, stream);
}

// ... and both gradients in one launch: dsr (+)= (scale_dev ? *scale_dev : 1) * (scale_host * d ST-loss / d sr + pix_weight * d pixel
// criterion / d sr).
SST_API int sst_st_pixel_loss_bwd(const float* sr, const float* gt, const float* gS, float* dsr, const float* scale_dev, float scale_host,
                                  float pix_weight, int pix_mode, int accumulate, int B, int H, int W, float sigma, float rho,
                                  void* stream) {
  SST_REQUIRE(gt && (pix_mode == 0 || pix_mode == 1), "sst_st_pixel_loss_bwd: bad pixel-criterion argument");
  return st_loss_bwd_impl(sr, gS, dsr, scale_dev, scale_host, accumulate, B, H, W, sigma, rho, gt,
                          pix_weight / (3.f * (float)B * (float)H * (float)W), pix_mode, stream);
}

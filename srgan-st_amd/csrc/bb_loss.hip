// Best-buddy losses on the GPU: reference loss.py:78-142 BestBuddyLoss, loss.py:145-228 GramLoss and loss.py:292-375
// PatchwiseStructureTensorLoss (matching via utils.py:157-191 batch_pairwise_distance).
//
// The SR and GT images are cut into non-overlapping 3 x 3 patches (27-vectors F in unfold order c*9 + ky*3 + kx).  A patch's
// FEATURE is the vector itself (BestBuddyLoss, mode 0), its 3x3 gram matrix G = F F^T / 27 with F viewed as [3 channels][9]
// (GramLoss, mode 1, 9-vector) or the normalised structure tensor of the 3x3 gray patch (PatchwiseStructureTensorLoss, mode 2:
// utils.py:212-239 on a 3x3 image = three fixed 9x9 linear maps: Ix = Ax g, Iy = Ay g, J = K (Ix^2, Iy^2, Ix Iy), then
// S / sqrt(det S + 1e-12); 27-vector (Jxx, Jyy, Jxy) x 9 pixels).  The candidate set is the GT features at scales 1, 1/2, 1/4; SR patch i is paired with the candidate j
// minimising   alpha * max(|f_sr_i|^2 + |c_j|^2 - 2 f_sr_i.c_j, 0) + beta * max(|f_gt_i|^2 + |c_j|^2 - 2 f_gt_i.c_j, 0)
// (the reference's expanded squared distance, clamped; first minimum wins like torch.min) and the loss is the mean L1 (or L2)
// between the SR features and their buddies.  Only that last criterion is differentiated (through the gram map for GramLoss).
//   bb_patches_kernel<GRAM> : image [B,3,H,W] -> features [B, nP, D] + squared norms, written into the candidate table
//   bb_match_kernel<D,GRAM> : 32 query patches x 8 candidate splits per workgroup (all lanes of a wave read the SAME candidate
//                             row from LDS: broadcast), argmin combined, then criterion term + gradient of the patch's 27
//                             pixels; per-workgroup loss partials
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BB_P = 27;                                       // raw patch: 3 channels x 3 x 3
constexpr int BB_Q = 32, BB_SPLIT = 8, BB_NT = BB_Q * BB_SPLIT;   // query patches x candidate splits per workgroup
constexpr int BB_CH = 128;                                     // candidates per LDS chunk

constexpr float BB_GW0 = 0.2989f, BB_GW1 = 0.587f, BB_GW2 = 0.114f;   // torchvision Grayscale (ITU-R 601), loss.py:341

// normalised structure tensor of a 3x3 patch; mats = [Ax 81][Ay 81][K 81] (row-major [out pixel][in pixel]).
// Also returns what the backward pass needs when `keep` is given: Ix, Iy, Jxx, Jyy, Jxy, r (9 each).
__device__ __forceinline__ void bb_st_forward(const float (&p)[BB_P], const float* mats, float* f, float* keep) {
  float g[9], ix[9], iy[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = BB_GW0 * p[k] + BB_GW1 * p[9 + k] + BB_GW2 * p[18 + k];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a = fmaf(mats[i * 9 + k], g[k], a);
      b = fmaf(mats[81 + i * 9 + k], g[k], b);
    }
    ix[i] = a;
    iy[i] = b;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float jxx = 0.f, jyy = 0.f, jxy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float kk = mats[162 + i * 9 + k];
      jxx = fmaf(kk, ix[k] * ix[k], jxx);
      jyy = fmaf(kk, iy[k] * iy[k], jyy);
      jxy = fmaf(kk, ix[k] * iy[k], jxy);
    }
    const float r = 1.f / sqrtf(jxx * jyy - jxy * jxy + 1e-12f);
    f[i] = jxx * r;
    f[9 + i] = jyy * r;
    f[18 + i] = jxy * r;
    if (keep) {
      keep[i] = ix[i]; keep[9 + i] = iy[i]; keep[18 + i] = jxx; keep[27 + i] = jyy; keep[36 + i] = jxy; keep[45 + i] = r;
    }
  }
}

// gp = (d feature / d patch)^T gf for the structure-tensor features (keep from bb_st_forward)
__device__ __forceinline__ void bb_st_backward(const float* keep, const float* mats, const float* gf, float* gp) {
  float dxx[9], dyy[9], dxy[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const float jxx = keep[18 + i], jyy = keep[27 + i], jxy = keep[36 + i], r = keep[45 + i];
    const float a = gf[i], b = gf[9 + i], c = gf[18 + i];
    const float t = -0.5f * (a * jxx + b * jyy + c * jxy) * r * r * r;        // through r = (det + eps)^(-1/2)
    dxx[i] = a * r + t * jyy;
    dyy[i] = b * r + t * jxx;
    dxy[i] = c * r - 2.f * t * jxy;
  }
  float dix[9], diy[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float pxx = 0.f, pyy = 0.f, pxy = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float kk = mats[162 + i * 9 + k];                               // K^T
      pxx = fmaf(kk, dxx[i], pxx);
      pyy = fmaf(kk, dyy[i], pyy);
      pxy = fmaf(kk, dxy[i], pxy);
    }
    const float ix = keep[k], iy = keep[9 + k];
    dix[k] = 2.f * ix * pxx + iy * pxy;
    diy[k] = 2.f * iy * pyy + ix * pxy;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float dg = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) dg = fmaf(mats[i * 9 + k], dix[i], fmaf(mats[81 + i * 9 + k], diy[i], dg));   // Ax^T, Ay^T
    gp[k] = BB_GW0 * dg;
    gp[9 + k] = BB_GW1 * dg;
    gp[18 + k] = BB_GW2 * dg;
  }
}

template <int GRAM>
__device__ __forceinline__ void bb_features(const float (&p)[BB_P], float* f, const float* mats) {   // f: 27 (raw / st) or 9 (gram)
  if (GRAM == 2) {
    bb_st_forward(p, mats, f, nullptr);
  } else if (GRAM == 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s = fmaf(p[a * 9 + k], p[b * 9 + k], s);
        f[a * 3 + b] = s / 27.f;
      }
  } else {
#pragma unroll
    for (int d = 0; d < BB_P; ++d) f[d] = p[d];
  }
}

template <int GRAM>
__global__ __launch_bounds__(256) void bb_patches_kernel(const float* __restrict__ img, float* __restrict__ out, float* __restrict__ nrm,
                                                         int B, int H, int W, int ncand_total, int cand_off,
                                                         const float* __restrict__ mats_g) {
  constexpr int D = GRAM == 1 ? 9 : BB_P;
  __shared__ float mats[243];
  if (GRAM == 2) {
    for (int i = threadIdx.x; i < 243; i += 256) mats[i] = mats_g[i];
    __syncthreads();
  }
  const int ph = H / 3, pw = W / 3, np = ph * pw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * np; i += gridDim.x * 256) {
    const int b = i / np, pidx = i - b * np, py = pidx / pw, px = pidx - py * pw;
    float p[BB_P], f[D];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) p[c * 9 + ky * 3 + kx] = img[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx];
    bb_features<GRAM>(p, f, mats);
    float* o = out + ((size_t)b * ncand_total + cand_off + pidx) * D;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      o[d] = f[d];
      s = fmaf(f[d], f[d], s);
    }
    nrm[(size_t)b * ncand_total + cand_off + pidx] = s;
  }
}

template <int GRAM>
__global__ __launch_bounds__(BB_NT) void bb_match_kernel(const float* __restrict__ sr, const float* __restrict__ cand,
                                                         const float* __restrict__ cnrm, int* __restrict__ ind_out,
                                                         float* __restrict__ dsr, float* __restrict__ partials, int B, int H, int W,
                                                         int ncand, float alpha, float beta, int l2, float inv_n,
                                                         const float* __restrict__ mats_g, int dist_l1) {
  constexpr int D = GRAM == 1 ? 9 : BB_P, DP = (D + 3) / 4 * 4;     // feature length, padded to 16-B rows in LDS
  __shared__ float mats[243];
  if (GRAM == 2) {
    for (int i = threadIdx.x; i < 243; i += BB_NT) mats[i] = mats_g[i];
    __syncthreads();
  }
  __shared__ __attribute__((aligned(16))) float sc[BB_CH][DP];
  __shared__ float sn[BB_CH];
  __shared__ float sbest[BB_SPLIT][BB_Q];
  __shared__ int sbi[BB_SPLIT][BB_Q];
  __shared__ float red[BB_NT / 64];
  const int ph = H / 3, pw = W / 3, np = ph * pw;
  const int nblk_img = (np + BB_Q - 1) / BB_Q;
  const int b = blockIdx.x / nblk_img;
  const int ql = threadIdx.x % BB_Q, part = threadIdx.x / BB_Q;
  const int q = (blockIdx.x - b * nblk_img) * BB_Q + ql;
  const bool live = q < np;
  const int py = live ? q / pw : 0, px = live ? q - py * pw : 0;
  float p[BB_P];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        p[c * 9 + ky * 3 + kx] = live ? sr[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx] : 0.f;
  float f1[DP], f2[DP];
  bb_features<GRAM>(p, f1, mats);
  float n1 = 0.f, n2 = 0.f;
#pragma unroll
  for (int d = 0; d < DP; ++d) {
    if (d >= D) f1[d] = 0.f;
    f2[d] = (live && d < D) ? cand[((size_t)b * ncand + q) * D + d] : 0.f;      // GT features are the first np candidates
    n1 = fmaf(f1[d], f1[d], n1);
    n2 = fmaf(f2[d], f2[d], n2);
  }
  float best = 3.4e38f;
  int bi = 0x7fffffff;
  for (int j0 = 0; j0 < ncand; j0 += BB_CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < BB_CH * DP; i += BB_NT) {
      const int j = i / DP, d = i - j * DP;
      sc[j][d] = (j0 + j < ncand && d < D) ? cand[((size_t)b * ncand + j0 + j) * D + d] : 0.f;
    }
    for (int j = threadIdx.x; j < BB_CH; j += BB_NT) sn[j] = (j0 + j < ncand) ? cnrm[(size_t)b * ncand + j0 + j] : 0.f;
    __syncthreads();
    const int nj = min(BB_CH, ncand - j0);
    for (int j = part; j < nj; j += BB_SPLIT) {
      float d1 = 0.f, d2 = 0.f;
      float s;
      if (dist_l1) {                                       // utils.py:166-172: sum |x - y| (padding entries are zero on both sides)
#pragma unroll
        for (int d4 = 0; d4 < DP / 4; ++d4) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(&sc[j][4 * d4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d1 += fabsf(f1[4 * d4 + e] - cv[e]);
            d2 += fabsf(f2[4 * d4 + e] - cv[e]);
          }
        }
        s = alpha * d1 + beta * d2;
      } else {
#pragma unroll
        for (int d4 = 0; d4 < DP / 4; ++d4) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(&sc[j][4 * d4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d1 = fmaf(f1[4 * d4 + e], cv[e], d1);
            d2 = fmaf(f2[4 * d4 + e], cv[e], d2);
          }
        }
        const float cn = sn[j];
        s = alpha * fmaxf(n1 + cn - 2.f * d1, 0.f) + beta * fmaxf(n2 + cn - 2.f * d2, 0.f);
      }
      if (s < best) { best = s; bi = j0 + j; }          // ascending j within a split: first minimum of the split
    }
  }
  sbest[part][ql] = best;
  sbi[part][ql] = bi;
  __syncthreads();
  float lsum = 0.f;
  if (part == 0 && live) {
#pragma unroll
    for (int k = 1; k < BB_SPLIT; ++k) {
      const float s = sbest[k][ql];
      const int i2 = sbi[k][ql];
      if (s < best || (s == best && i2 < bi)) { best = s; bi = i2; }
    }
    ind_out[(size_t)b * np + q] = bi;
    const float* sel = cand + ((size_t)b * ncand + bi) * D;
    float gf[D];                                          // d(loss)/d(feature)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float df = f1[d] - sel[d];
      if (l2) { lsum = fmaf(df, df, lsum); gf[d] = 2.f * df * inv_n; }
      else { lsum += fabsf(df); gf[d] = (df > 0.f ? inv_n : (df < 0.f ? -inv_n : 0.f)); }
    }
    float gp[BB_P];                                       // d(loss)/d(patch pixels)
    if (GRAM == 2) {
      float keep[54], ftmp[BB_P];
      bb_st_forward(p, mats, ftmp, keep);                 // recomputed here: only one thread in 8 needs the intermediates
      bb_st_backward(keep, mats, gf, gp);
    } else if (GRAM == 1) {                               // G = F F^T / 27  ->  dF = (dG + dG^T) F / 27
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) s = fmaf(gf[a * 3 + c] + gf[c * 3 + a], p[c * 9 + k], s);
          gp[a * 9 + k] = s / 27.f;
        }
    } else {
#pragma unroll
      for (int d = 0; d < BB_P; ++d) gp[d] = gf[d < D ? d : 0];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          dsr[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx] = gp[c * 9 + ky * 3 + kx];
  }
  const float tot = block_sum<BB_NT>(lsum, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot * inv_n;
}

// ---------------------------------------------------------------------------------------------
// General geometry (BestBuddyLoss with any ksize <= 6, pad, stride - reference loss.py:86,116-129 F.unfold - and dist_norm 'l1' or
// 'l2', utils.py:157-191): patch features live in global tables [B, n, D = 3 k k] (unfold order c*k*k + ky*k + kx, zero padding),
// the matching is the tiled loop above on runtime-sized rows (LDS: 32 query pairs + 64 candidates), the gradient goes through a
// per-patch gradient table that bbg_fold_kernel gathers into d(sr) - overlapping patches (stride < ksize) included, fixed order.
__global__ __launch_bounds__(256) void bbg_unfold_kernel(const float* __restrict__ img, float* __restrict__ out, float* __restrict__ nrm,
                                                         int B, int H, int W, int k, int pad, int stride, int oh, int ow,
                                                         int nrows_total, int row_off) {
  const int np = oh * ow, D = 3 * k * k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * np; i += gridDim.x * 256) {
    const int b = i / np, pidx = i - b * np, py = pidx / ow, px = pidx - py * ow;
    float* o = out + ((size_t)b * nrows_total + row_off + pidx) * D;
    float s = 0.f;
    for (int c = 0; c < 3; ++c)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          const int y = py * stride - pad + ky, x = px * stride - pad + kx;
          const float v = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? img[(((size_t)b * 3 + c) * H + y) * W + x] : 0.f;
          o[(c * k + ky) * k + kx] = v;
          s = fmaf(v, v, s);
        }
    if (nrm) nrm[(size_t)b * nrows_total + row_off + pidx] = s;
  }
}

constexpr int BBG_CH = 64;      // candidates per LDS chunk
__global__ __launch_bounds__(BB_NT) void bbg_match_kernel(const float* __restrict__ srf, const float* __restrict__ cand,
                                                          const float* __restrict__ cnrm, int* __restrict__ ind_out,
                                                          float* __restrict__ gfeat, float* __restrict__ partials, int B, int np,
                                                          int ncand, int D, float alpha, float beta, int crit_l2, int dist_l1,
                                                          float inv_n) {
  extern __shared__ float bbg_lds[];
  const int DS = D + 1;                                   // row stride: lanes of a wave read different query rows at the same d
  float* fq1 = bbg_lds;                                   // [BB_Q][DS] SR features of the block's queries
  float* fq2 = fq1 + BB_Q * DS;                           // [BB_Q][DS] GT features (the first np candidates)
  float* sc = fq2 + BB_Q * DS;                            // [BBG_CH][DS]
  __shared__ float sn[BBG_CH];
  __shared__ float sbest[BB_SPLIT][BB_Q];
  __shared__ int sbi[BB_SPLIT][BB_Q];
  __shared__ float red[BB_NT / 64];
  const int nblk_img = (np + BB_Q - 1) / BB_Q;
  const int b = blockIdx.x / nblk_img, q0 = (blockIdx.x - b * nblk_img) * BB_Q;
  const int ql = threadIdx.x % BB_Q, part = threadIdx.x / BB_Q;
  const int q = q0 + ql;
  const bool live = q < np;
  for (int i = threadIdx.x; i < BB_Q * D; i += BB_NT) {
    const int r = i / D, d = i - r * D;
    const bool ok = q0 + r < np;
    fq1[r * DS + d] = ok ? srf[((size_t)b * np + q0 + r) * D + d] : 0.f;
    fq2[r * DS + d] = ok ? cand[((size_t)b * ncand + q0 + r) * D + d] : 0.f;
  }
  __syncthreads();
  float n1 = 0.f, n2 = 0.f;
  for (int d = 0; d < D; ++d) {
    n1 = fmaf(fq1[ql * DS + d], fq1[ql * DS + d], n1);
    n2 = fmaf(fq2[ql * DS + d], fq2[ql * DS + d], n2);
  }
  float best = 3.4e38f;
  int bi = 0x7fffffff;
  for (int j0 = 0; j0 < ncand; j0 += BBG_CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < BBG_CH * D; i += BB_NT) {
      const int j = i / D, d = i - j * D;
      sc[j * DS + d] = (j0 + j < ncand) ? cand[((size_t)b * ncand + j0 + j) * D + d] : 0.f;
    }
    for (int j = threadIdx.x; j < BBG_CH; j += BB_NT) sn[j] = (j0 + j < ncand) ? cnrm[(size_t)b * ncand + j0 + j] : 0.f;
    __syncthreads();
    const int nj = min(BBG_CH, ncand - j0);
    for (int j = part; j < nj; j += BB_SPLIT) {
      float d1 = 0.f, d2 = 0.f;
      if (dist_l1) {
        for (int d = 0; d < D; ++d) {
          const float cv = sc[j * DS + d];
          d1 += fabsf(fq1[ql * DS + d] - cv);
          d2 += fabsf(fq2[ql * DS + d] - cv);
        }
      } else {
        for (int d = 0; d < D; ++d) {
          const float cv = sc[j * DS + d];
          d1 = fmaf(fq1[ql * DS + d], cv, d1);
          d2 = fmaf(fq2[ql * DS + d], cv, d2);
        }
        const float cn = sn[j];
        d1 = fmaxf(n1 + cn - 2.f * d1, 0.f);
        d2 = fmaxf(n2 + cn - 2.f * d2, 0.f);
      }
      const float s = alpha * d1 + beta * d2;
      if (s < best) { best = s; bi = j0 + j; }
    }
  }
  sbest[part][ql] = best;
  sbi[part][ql] = bi;
  __syncthreads();
  float lsum = 0.f;
  if (part == 0 && live) {
    for (int kk = 1; kk < BB_SPLIT; ++kk) {
      const float s = sbest[kk][ql];
      const int i2 = sbi[kk][ql];
      if (s < best || (s == best && i2 < bi)) { best = s; bi = i2; }
    }
    ind_out[(size_t)b * np + q] = bi;
    const float* sel = cand + ((size_t)b * ncand + bi) * D;
    float* g = gfeat + ((size_t)b * np + q) * D;
    for (int d = 0; d < D; ++d) {
      const float df = fq1[ql * DS + d] - sel[d];
      if (crit_l2) { lsum = fmaf(df, df, lsum); g[d] = 2.f * df * inv_n; }
      else { lsum += fabsf(df); g[d] = (df > 0.f ? inv_n : (df < 0.f ? -inv_n : 0.f)); }
    }
  }
  const float tot = block_sum<BB_NT>(lsum, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot * inv_n;
}

// d(sr)[b,c,y,x] = sum over the patches that contain the pixel of their gradient entry (F.unfold's adjoint), row-major patch order
__global__ __launch_bounds__(256) void bbg_fold_kernel(const float* __restrict__ gfeat, float* __restrict__ dsr, int B, int H, int W, int k,
                                                       int pad, int stride, int oh, int ow) {
  const int D = 3 * k * k, np = oh * ow;
  const size_t total = (size_t)B * 3 * H * W;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    size_t t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % 3), b = (int)(t / 3);
    float s = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int yy = y + pad - ky;
      if (yy < 0 || yy % stride) continue;
      const int py = yy / stride;
      if (py >= oh) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int xx = x + pad - kx;
        if (xx < 0 || xx % stride) continue;
        const int px = xx / stride;
        if (px >= ow) continue;
        s += gfeat[((size_t)b * np + py * ow + px) * D + (c * k + ky) * k + kx];
      }
    }
    dsr[i] = s;
  }
}

}  // namespace

// ---- general patch geometry for BestBuddyLoss (loss.py:86): k = ksize (1..6), pad, stride; D = 3 k k; rows per image
// sst_bbg_patches(H, W, k, pad, stride) = F.unfold's patch count.  sst_bbg_unfold writes rows [row_off, row_off + patches) of a
// [B, nrows_total, D] table (+ squared norms, or null); sst_bbg_match pairs the SR rows srf [B, np, D] with the candidate table
// (its first np rows = full-resolution GT patches), dist_l1 = utils.py:166-172, else the clamped expanded squared distance
// utils.py:173-187; ind [B, np], gfeat [B, np, D] = d(loss)/d(SR patch entries), partials [sst_bbg_blocks(B, np)];
// sst_bbg_fold = the adjoint of the unfold: d(sr) [B,3,H,W].
SST_API int sst_bbg_patches(int H, int W, int k, int pad, int stride) {
  if (k < 1 || stride < 1 || pad < 0 || H + 2 * pad < k || W + 2 * pad < k) return 0;
  return ((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
}
SST_API int sst_bbg_blocks(int B, int np) { return B * ((np + BB_Q - 1) / BB_Q); }
SST_API int sst_bbg_unfold(const float* img, float* table, float* nrm, int B, int H, int W, int k, int pad, int stride, int nrows_total,
                           int row_off, void* stream) {
  const int np = sst_bbg_patches(H, W, k, pad, stride);
  SST_REQUIRE(img && table && B > 0 && np > 0 && k <= 6 && row_off >= 0 && row_off + np <= nrows_total, "sst_bbg_unfold: bad argument");
  const int oh = (H + 2 * pad - k) / stride + 1, ow = (W + 2 * pad - k) / stride + 1;
  const int total = B * np;
  bbg_unfold_kernel<<<(total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096, 256, 0, sst_stream(stream)>>>(img, table, nrm, B, H, W, k, pad,
                                                                                                            stride, oh, ow, nrows_total, row_off);
  SST_LAUNCH_CHECK("bbg_unfold_kernel");
  return SST_OK;
}
SST_API int sst_bbg_match(const float* srf, const float* cand, const float* cnrm, int* ind, float* gfeat, float* partials, int B, int np,
                          int ncand, int D, float alpha, float beta, int criterion_l2, int dist_l1, void* stream) {
  SST_REQUIRE(srf && cand && cnrm && ind && gfeat && partials && B > 0 && np > 0 && ncand >= np && D > 0 && D <= 108,
              "sst_bbg_match: bad argument (D = %d, at most 108 = ksize 6)", D);
  const float inv_n = 1.f / ((float)B * np * D);
  const size_t lds = (size_t)(2 * BB_Q + BBG_CH) * (D + 1) * sizeof(float);
  bbg_match_kernel<<<sst_bbg_blocks(B, np), BB_NT, lds, sst_stream(stream)>>>(srf, cand, cnrm, ind, gfeat, partials, B, np, ncand, D, alpha, beta,
                                                                             criterion_l2, dist_l1, inv_n);
  SST_LAUNCH_CHECK("bbg_match_kernel");
  return SST_OK;
}
SST_API int sst_bbg_fold(const float* gfeat, float* dsr, int B, int H, int W, int k, int pad, int stride, void* stream) {
  const int np = sst_bbg_patches(H, W, k, pad, stride);
  SST_REQUIRE(gfeat && dsr && B > 0 && np > 0 && k <= 6, "sst_bbg_fold: bad argument");
  const int oh = (H + 2 * pad - k) / stride + 1, ow = (W + 2 * pad - k) / stride + 1;
  const size_t total = (size_t)B * 3 * H * W;
  bbg_fold_kernel<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, sst_stream(stream)>>>(gfeat, dsr, B, H, W, k, pad,
                                                                                                                   stride, oh, ow);
  SST_LAUNCH_CHECK("bbg_fold_kernel");
  return SST_OK;
}

SST_API int sst_bb_blocks(int B, int H, int W) { return B * (((H / 3) * (W / 3) + BB_Q - 1) / BB_Q); }
SST_API int sst_bb_feature_dim(int gram) { return gram == 1 ? 9 : BB_P; }

// img [B,3,H,W] (H, W multiples of 3) -> rows [cand_off, cand_off + (H/3)(W/3)) of cand [B, ncand_total, D] and cnrm
// [B, ncand_total]; D = sst_bb_feature_dim(gram): raw 27-vector (0, BestBuddyLoss), 3x3 gram matrix (1, GramLoss) or the
// normalised 3x3-patch structure tensor (2, PatchwiseStructureTensorLoss; st_mats = device [Ax 81][Ay 81][K 81]).
SST_API int sst_bb_patches(const float* img, float* cand, float* cnrm, int B, int H, int W, int ncand_total, int cand_off, int gram,
                           const float* st_mats, void* stream) {
  SST_REQUIRE(gram >= 0 && gram <= 2 && (gram != 2 || st_mats), "sst_bb_patches: feature mode 0..2, mode 2 needs st_mats");
  SST_REQUIRE(img && cand && cnrm && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  cand_off >= 0 && cand_off + (H / 3) * (W / 3) <= ncand_total, "sst_bb_patches: bad argument");
  const int total = B * (H / 3) * (W / 3);
  if (gram == 2)
    bb_patches_kernel<2><<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off, st_mats);
  else if (gram == 1)
    bb_patches_kernel<1><<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off, nullptr);
  else
    bb_patches_kernel<0><<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off, nullptr);
  SST_LAUNCH_CHECK("bb_patches_kernel");
  return SST_OK;
}

SST_API int sst_bb_match_dist(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B, int H,
                              int W, int ncand, float alpha, float beta, int criterion_l2, int gram, const float* st_mats, int dist_l1,
                              void* stream);
// sr [B,3,H,W]; cand / cnrm: candidate table whose first (H/3)(W/3) rows are the full-resolution GT features.
// ind [B, nP] int32; dsr [B,3,H,W] = d(loss)/d(sr) for loss = sum(partials) (criterion mean over B*nP*D elements);
// partials [sst_bb_blocks(B,H,W)].
SST_API int sst_bb_match(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B, int H,
                         int W, int ncand, float alpha, float beta, int criterion_l2, int gram, const float* st_mats, void* stream) {
  return sst_bb_match_dist(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand, alpha, beta, criterion_l2, gram, st_mats, 0, stream);
}
// ... with the matching distance of utils.py:157-191 selectable: dist_l1 = 0 the (clamped, expanded) squared L2, 1 the L1 distance
SST_API int sst_bb_match_dist(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B, int H,
                              int W, int ncand, float alpha, float beta, int criterion_l2, int gram, const float* st_mats, int dist_l1,
                              void* stream) {
  SST_REQUIRE(gram >= 0 && gram <= 2 && (gram != 2 || st_mats), "sst_bb_match: feature mode 0..2, mode 2 needs st_mats");
  SST_REQUIRE(sr && cand && cnrm && ind && dsr && partials && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  ncand >= (H / 3) * (W / 3), "sst_bb_match: bad argument");
  const float inv_n = 1.f / ((float)B * (H / 3) * (W / 3) * sst_bb_feature_dim(gram));
  if (gram == 2)
    bb_match_kernel<2><<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand,
                                                                                 alpha, beta, criterion_l2, inv_n, st_mats, dist_l1);
  else if (gram == 1)
    bb_match_kernel<1><<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand,
                                                                                 alpha, beta, criterion_l2, inv_n, nullptr, dist_l1);
  else
    bb_match_kernel<0><<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand,
                                                                                 alpha, beta, criterion_l2, inv_n, nullptr, dist_l1);
  SST_LAUNCH_CHECK("bb_match_kernel");
  return SST_OK;
}

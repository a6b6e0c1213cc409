// Best-buddy losses on the GPU: reference loss.py:78-142 BestBuddyLoss and loss.py:145-228 GramLoss (matching via
// utils.py:157-191 batch_pairwise_distance).
//
// The SR and GT images are cut into non-overlapping 3 x 3 patches (27-vectors F in unfold order c*9 + ky*3 + kx).  A patch's
// FEATURE is the vector itself (BestBuddyLoss) or its 3x3 gram matrix G = F F^T / 27 with F viewed as [3 channels][9]
// (GramLoss, 9-vector).  The candidate set is the GT features at scales 1, 1/2, 1/4; SR patch i is paired with the candidate j
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

template <int GRAM>
__device__ __forceinline__ void bb_features(const float (&p)[BB_P], float* f) {   // f: 27 (raw) or 9 (gram) values
  if (GRAM) {
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
                                                         int B, int H, int W, int ncand_total, int cand_off) {
  constexpr int D = GRAM ? 9 : BB_P;
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
    bb_features<GRAM>(p, f);
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
                                                         int ncand, float alpha, float beta, int l2, float inv_n) {
  constexpr int D = GRAM ? 9 : BB_P, DP = (D + 3) / 4 * 4;     // feature length, padded to 16-B rows in LDS
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
  bb_features<GRAM>(p, f1);
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
      const float s = alpha * fmaxf(n1 + cn - 2.f * d1, 0.f) + beta * fmaxf(n2 + cn - 2.f * d2, 0.f);
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
    if (GRAM) {                                           // G = F F^T / 27  ->  dF = (dG + dG^T) F / 27
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

}  // namespace

SST_API int sst_bb_blocks(int B, int H, int W) { return B * (((H / 3) * (W / 3) + BB_Q - 1) / BB_Q); }
SST_API int sst_bb_feature_dim(int gram) { return gram ? 9 : BB_P; }

// img [B,3,H,W] (H, W multiples of 3) -> rows [cand_off, cand_off + (H/3)(W/3)) of cand [B, ncand_total, D] and cnrm
// [B, ncand_total]; D = sst_bb_feature_dim(gram): the raw 27-vector (BestBuddyLoss) or the 3x3 gram matrix (GramLoss).
SST_API int sst_bb_patches(const float* img, float* cand, float* cnrm, int B, int H, int W, int ncand_total, int cand_off, int gram,
                           void* stream) {
  SST_REQUIRE(img && cand && cnrm && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  cand_off >= 0 && cand_off + (H / 3) * (W / 3) <= ncand_total, "sst_bb_patches: bad argument");
  const int total = B * (H / 3) * (W / 3);
  if (gram)
    bb_patches_kernel<1><<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off);
  else
    bb_patches_kernel<0><<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off);
  SST_LAUNCH_CHECK("bb_patches_kernel");
  return SST_OK;
}

// sr [B,3,H,W]; cand / cnrm: candidate table whose first (H/3)(W/3) rows are the full-resolution GT features.
// ind [B, nP] int32; dsr [B,3,H,W] = d(loss)/d(sr) for loss = sum(partials) (criterion mean over B*nP*D elements);
// partials [sst_bb_blocks(B,H,W)].
SST_API int sst_bb_match(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B, int H,
                         int W, int ncand, float alpha, float beta, int criterion_l2, int gram, void* stream) {
  SST_REQUIRE(sr && cand && cnrm && ind && dsr && partials && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  ncand >= (H / 3) * (W / 3), "sst_bb_match: bad argument");
  const float inv_n = 1.f / ((float)B * (H / 3) * (W / 3) * sst_bb_feature_dim(gram));
  if (gram)
    bb_match_kernel<1><<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand,
                                                                                 alpha, beta, criterion_l2, inv_n);
  else
    bb_match_kernel<0><<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand,
                                                                                 alpha, beta, criterion_l2, inv_n);
  SST_LAUNCH_CHECK("bb_match_kernel");
  return SST_OK;
}

// Best-buddy loss (reference loss.py:78-142 BestBuddyLoss, utils.py:157-191 batch_pairwise_distance) on the GPU.
//
// The SR and GT images are cut into non-overlapping k x k patches (k = 3: 27-vectors in unfold order c*9 + ky*3 + kx); the
// candidate set is the GT patches at scales 1, 1/2, 1/4; SR patch i is paired with the candidate j that minimises
//   alpha * max(|p_sr_i|^2 + |c_j|^2 - 2 p_sr_i.c_j, 0) + beta * max(|p_gt_i|^2 + |c_j|^2 - 2 p_gt_i.c_j, 0)
// (the reference's expanded squared distance, clamped) and the loss is the mean L1 (or L2) between the SR patches and their
// buddies.  Only that last criterion is differentiated.
//   bb_patches_kernel : image [B,3,H,W] -> patches [B, nP, 27] + squared norms, written into the candidate table
//   bb_match_kernel   : 8 threads per SR patch scan interleaved slices of the candidates (LDS chunks, broadcast reads), the
//                       argmin is combined, then the criterion term and the (unscaled) gradient of the patch's 27 pixels;
//                       per-workgroup loss partials
#include "common.h"

namespace {

constexpr int BB_D = 27;          // 3 channels x 3 x 3
constexpr int BB_Q = 32, BB_SPLIT = 8, BB_NT = BB_Q * BB_SPLIT;   // query patches x candidate splits per workgroup
constexpr int BB_CH = 128;        // candidates per LDS chunk

__global__ __launch_bounds__(256) void bb_patches_kernel(const float* __restrict__ img, float* __restrict__ out, float* __restrict__ nrm,
                                                         int B, int H, int W, int ncand_total, int cand_off) {
  const int ph = H / 3, pw = W / 3, np = ph * pw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * np; i += gridDim.x * 256) {
    const int b = i / np, pidx = i - b * np, py = pidx / pw, px = pidx - py * pw;
    float* o = out + ((size_t)b * ncand_total + cand_off + pidx) * BB_D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = img[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx];
          o[c * 9 + ky * 3 + kx] = v;
          s = fmaf(v, v, s);
        }
    nrm[(size_t)b * ncand_total + cand_off + pidx] = s;
  }
}

// 32 query patches x 8 candidate splits per workgroup: lane = query, the 8 waves... (4 waves x 2) scan interleaved slices of
// the candidate chunk (all lanes of a wave read the SAME candidate row: LDS broadcast, 7 x 16 B per row); the 8 partial
// argmins of a query are combined with "lower score, then lower index" = torch.min's first minimum.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(BB_NT) void bb_match_kernel(const float* __restrict__ sr, const float* __restrict__ cand,
                                                         const float* __restrict__ cnrm, int* __restrict__ ind_out,
                                                         float* __restrict__ dsr, float* __restrict__ partials, int B, int H, int W,
                                                         int ncand, float alpha, float beta, int l2, float inv_n) {
  __shared__ __attribute__((aligned(16))) float sc[BB_CH][BB_D + 1];
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
  float p1[BB_D + 1], p2[BB_D + 1];
  float n1 = 0.f, n2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int d = c * 9 + ky * 3 + kx;
        p1[d] = live ? sr[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx] : 0.f;
        p2[d] = live ? cand[((size_t)b * ncand + q) * BB_D + d] : 0.f;          // GT patches are the first np candidates
        n1 = fmaf(p1[d], p1[d], n1);
        n2 = fmaf(p2[d], p2[d], n2);
      }
  p1[BB_D] = p2[BB_D] = 0.f;
  float best = 3.4e38f;
  int bi = 0x7fffffff;
  for (int j0 = 0; j0 < ncand; j0 += BB_CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < BB_CH * (BB_D + 1); i += BB_NT) {
      const int j = i / (BB_D + 1), d = i - j * (BB_D + 1);
      sc[j][d] = (j0 + j < ncand && d < BB_D) ? cand[((size_t)b * ncand + j0 + j) * BB_D + d] : 0.f;
    }
    for (int j = threadIdx.x; j < BB_CH; j += BB_NT) sn[j] = (j0 + j < ncand) ? cnrm[(size_t)b * ncand + j0 + j] : 0.f;
    __syncthreads();
    const int nj = min(BB_CH, ncand - j0);
    for (int j = part; j < nj; j += BB_SPLIT) {
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < (BB_D + 1) / 4; ++d4) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(&sc[j][4 * d4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d1 = fmaf(p1[4 * d4 + e], cv[e], d1);
          d2 = fmaf(p2[4 * d4 + e], cv[e], d2);
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
    const float* sel = cand + ((size_t)b * ncand + bi) * BB_D;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int d = c * 9 + ky * 3 + kx;
          const float df = p1[d] - sel[d];
          float gv;
          if (l2) { lsum = fmaf(df, df, lsum); gv = 2.f * df * inv_n; }
          else { lsum += fabsf(df); gv = (df > 0.f ? inv_n : (df < 0.f ? -inv_n : 0.f)); }
          dsr[(((size_t)b * 3 + c) * H + py * 3 + ky) * W + px * 3 + kx] = gv;
        }
  }
  const float tot = block_sum<BB_NT>(lsum, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot * inv_n;
}

}  // namespace

SST_API int sst_bb_blocks(int B, int H, int W) { return B * (((H / 3) * (W / 3) + BB_Q - 1) / BB_Q); }

// img [B,3,H,W] (H, W multiples of 3) -> rows [cand_off, cand_off + (H/3)(W/3)) of cand [B, ncand_total, 27] and cnrm [B, ncand_total]
SST_API int sst_bb_patches(const float* img, float* cand, float* cnrm, int B, int H, int W, int ncand_total, int cand_off,
                           void* stream) {
  SST_REQUIRE(img && cand && cnrm && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  cand_off >= 0 && cand_off + (H / 3) * (W / 3) <= ncand_total, "sst_bb_patches: bad argument");
  const int total = B * (H / 3) * (W / 3);
  bb_patches_kernel<<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(img, cand, cnrm, B, H, W, ncand_total, cand_off);
  SST_LAUNCH_CHECK("bb_patches_kernel");
  return SST_OK;
}

// sr [B,3,H,W]; cand / cnrm: candidate table whose first (H/3)(W/3) rows are the full-resolution GT patches.
// ind [B, nP] int32; dsr [B,3,H,W] = d(loss)/d(sr) for loss = sum(partials) (criterion mean over B*nP*27 elements);
// partials [sst_bb_blocks(B,H,W)].
SST_API int sst_bb_match(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B, int H,
                         int W, int ncand, float alpha, float beta, int criterion_l2, void* stream) {
  SST_REQUIRE(sr && cand && cnrm && ind && dsr && partials && B > 0 && H >= 3 && W >= 3 && H % 3 == 0 && W % 3 == 0 &&
                  ncand >= (H / 3) * (W / 3), "sst_bb_match: bad argument");
  const float inv_n = 1.f / ((float)B * (H / 3) * (W / 3) * BB_D);
  bb_match_kernel<<<sst_bb_blocks(B, H, W), BB_NT, 0, sst_stream(stream)>>>(sr, cand, cnrm, ind, dsr, partials, B, H, W, ncand, alpha,
                                                                            beta, criterion_l2, inv_n);
  SST_LAUNCH_CHECK("bb_match_kernel");
  return SST_OK;
}

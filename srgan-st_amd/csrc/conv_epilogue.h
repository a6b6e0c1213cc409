// Shared by the conv forward / data-gradient kernels (conv_fwd.hip: 32 px x 32 ch tiles, conv_fwd2.hip: 64 x 64):
// launch arguments, store modes and the per-(32x32)-tile epilogue (K-partial reduction through LDS, bias, residual,
// store pattern, BatchNorm forward statistics, BatchNorm/activation backward partials).
#pragma once
#include "conv_common.h"

// (types at namespace scope: they cross translation units through sst_launch_conv_fwd2)
constexpr int TWO = 8, THO = 4;   // output tile (pixels)

struct Conv3Args {
  const float* x;          // [B,H,W,Cin]
  const float* wp;         // packed weights
  float* y;                // [B,Ho,Wo,Cout]
  const float* bias;       // [Cout] or null
  const float* in_scale;   // [Cin] or null : x <- x*scale + shift (per channel) before the activation
  const float* in_shift;
  const float* in_slope;   // device scalar or null
  float in_slope_const;    // used when in_slope == null
  int in_act;              // ACT_*
  const float* residual;   // [B,Ho,Wo,Cout] or null
  float* stats;            // [n_mtiles][2][Cout] (sum, M2) or null
  float* stats_cnt;        // [n_mtiles] valid pixels per tile (written when stats != null)
  float* y_pre;            // OUT_NCHW_CLAMP: pre-clamp copy (saved for backward) or null
  // optional fused BatchNorm-backward APPLY on the input (this conv is then the data-gradient of the layer below):
  //   the staged value is dy = cA*gz + cB*y2 + cC with gz = in_act ? (y2*in_scale+in_shift > 0 ? x : x*slope) : x,
  //   x = upstream gradient, in2 = saved conv output y2 [B,H,W,Cin]; dy of the tile's own pixels goes to side_out.
  const float* in2; const float* in_cA; const float* in_cB; const float* in_cC; float* side_out;
  // optional BatchNorm/activation BACKWARD partials of the stored value g (this conv is then a data-gradient):
  //   with z = epi_y*epi_scale+epi_shift (or epi_y), gz = epi_act ? (z>0 ? g : g*slope) : g
  //   epi_partial[mtile][0..2][c] = sum over the tile's pixels of (gz, gz*epi_y, g*min(z,0))   (layout of bwd_reduce)
  const float* epi_y; const float* epi_scale; const float* epi_shift; const float* epi_slope;
  float epi_slope_const; int epi_act; float* epi_partial;
  int out_mode;            // OUT_*
  int ksy, ksx, pad_y, pad_x;   // runtime tap window (<= KS x KS) and padding: KS,KS,KS/2,KS/2 for a plain conv
  int sub_y, sub_x;        // OUT_STRIDE2: parity class of the scattered output pixels
  int dbg;                 // ablation bits for tools/ablate_conv.py (0 in production): 1 skip staging, 2 skip K loop, 4 skip epilogue
  int B, H, W, Cin, Cout, Ho, Wo;
  int Hy, Wy;              // OUT_STRIDE2: full size of y
};

// "Accumulator" mode of the BatchNorm statistics (optional second kernel argument): instead of per-band partial tiles that a
// separate finalize launch reduces, the producer conv ADDS its per-band (sum, M2 + n*mean^2) into per-image fp64 accumulators
// acc[rep = image % nrep][channel][2] with hardware fp64 atomics, and the CONSUMER conv derives the batch statistics and the
// BatchNorm affine of its input from those accumulators in its own prologue (16 lanes per wave, one channel each).  One
// launch per BatchNorm layer disappears.  fp64 sums of fp32-precise terms are order-independent up to 1e-16 relative, so the
// fp32 results are reproducible in practice; Var = (Bq - A^2/n)/n is evaluated in fp64 (no cancellation problem).
struct BandAcc {
  double* st_acc;            // producer: accumulators of THIS conv's output statistics, or null
  const double* in_acc;      // consumer: accumulators of the input's BatchNorm, or null
  int nrep;                  // replicas (images are spread over them to keep same-address atomics short)
  int in_target;             // 0: the affine is the (scale, shift) of the activation prologue; 1: it is (kB, kC) of
                             //    staged = x + in2*kB + kC  (residual sum of the previous block, kA = 1)
  const float* in_gamma; const float* in_beta;
  float in_n, in_eps, momentum;
  float* o_mean; float* o_rstd; float* o_scale; float* o_shift;   // written once (workgroup 0) for the backward pass, may be null
  float* run_mean; float* run_var;                                 // running statistics (train mode), may be null
  // ---- backward: the BatchNorm/activation backward partial sums (sum gz, sum gz*y, sum g*min(z,0)) the same way:
  // producer adds its per-band sums into bw_st_acc[nrep][Cout][4] (4th word unused), the consumer derives the coefficients of
  //   dy = cA*gz + cB*y + cC   from bw_in_acc + (mean, rstd, gamma) in its prologue and publishes dgamma / dbeta / dslope once.
  double* bw_st_acc;
  const double* bw_in_acc;
  const float* bw_mean; const float* bw_rstd; const float* bw_gamma;
  float bw_n;
  float* o_dgamma; float* o_dbeta; float* o_dslope;                // o_dslope may be null (BatchNorm without activation)
};

// how the epilogue stores the [B,Ho,Wo,Cout] result
enum : int {
  OUT_NHWC = 0,
  OUT_SHUFFLE = 1,      // PixelShuffle(2): y[b, 2oy+i, 2ox+j, c] = out[b,oy,ox,4c+2i+j]      (model.py:160)
  OUT_NCHW_CLAMP = 2,   // y[b,co,oy,ox] = clamp(out,0,1), y_pre = out                          (model.py:148-150)
  OUT_UNSHUFFLE = 3,    // inverse of OUT_SHUFFLE: y[b, oy/2, ox/2, 4c + 2(oy&1) + (ox&1)] = out[b,oy,ox,c]
  OUT_STRIDE2 = 4,      // y[b, 2oy+sub_y, 2ox+sub_x, c] = out (y is [B,Hy,Wy,Cout]: data-gradient of a stride-2 conv, one parity class)
};


namespace {

// Epilogue of ONE 32-pixel (8 wide x 4 high at (oy0, ox0)) x 32-channel (block nf) tile whose K-partials sit in the
// 4 waves' accumulators.  mt = statistics-tile index (8x4 tiling of the output); tile_ok = the tile has valid pixels.
// Must be called by all 256 threads (contains workgroup barriers); `lds` needs 4*32*33 floats.
__device__ __forceinline__ void conv_tile_epilogue(const Conv3Args& a, float* lds, float (*sstat)[3][32], const f32x16& acc,
                                                   int b, int oy0, int ox0, int nf, int mt, bool tile_ok, int tid, int wave,
                                                   int lane) {
  const int li = lane & 31, lh = lane >> 5;
  // ---- reduce the 4 K-partials through LDS (overlays the patch)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    lds[(wave * 32 + row) * 33 + li] = acc[r];
  }
  __syncthreads();
  const int p = tid >> 3, cq = (tid & 7) * 4;      // pixel of the tile, first of 4 output channels
  const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
  const int n0 = nf * 32 + cq;
  const bool pix_ok = oy < a.Ho && ox < a.Wo;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = lds[(0 * 32 + p) * 33 + cq + j];
    s += lds[(1 * 32 + p) * 33 + cq + j];
    s += lds[(2 * 32 + p) * 33 + cq + j];
    s += lds[(3 * 32 + p) * 33 + cq + j];
    if (a.bias && n0 + j < a.Cout) s += a.bias[n0 + j];
    v[j] = s;
  }
  const size_t obase = (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + n0;
  if (pix_ok) {
    if (a.residual) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < a.Cout) v[j] += a.residual[obase + j];
    }
    if (a.out_mode == OUT_NHWC) {
      if ((a.Cout & 3) == 0 && n0 + 3 < a.Cout) {
        *reinterpret_cast<f32x4*>(a.y + obase) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < a.Cout) a.y[obase + j] = v[j];
      }
    } else if (a.out_mode == OUT_SHUFFLE) {
      const int Cs = a.Cout >> 2, c = n0 >> 2;   // n0 is a multiple of 4: the 4 values are the 2x2 sub-pixels of channel c
      if (n0 < a.Cout) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a.y[(((size_t)b * 2 * a.Ho + 2 * oy + (j >> 1)) * 2 * a.Wo + 2 * ox + (j & 1)) * Cs + c] = v[j];
      }
    } else if (a.out_mode == OUT_NCHW_CLAMP) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < a.Cout) {
          const size_t o = (((size_t)b * a.Cout + n0 + j) * a.Ho + oy) * a.Wo + ox;
          if (a.y_pre) a.y_pre[o] = v[j];
          a.y[o] = fminf(fmaxf(v[j], 0.f), 1.f);
        }
    } else if (a.out_mode == OUT_UNSHUFFLE) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < a.Cout)
          a.y[(((size_t)b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * (4 * a.Cout) + 4 * (n0 + j) +
              2 * (oy & 1) + (ox & 1)] = v[j];
    } else {  // OUT_STRIDE2: (Ho,Wo) is this parity class's sub-grid of the [Hy,Wy] tensor
      const int Y = 2 * oy + a.sub_y, X = 2 * ox + a.sub_x;
      float* d = a.y + (((size_t)b * a.Hy + Y) * a.Wy + X) * a.Cout + n0;
      if ((a.Cout & 3) == 0 && n0 + 3 < a.Cout) {
        *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < a.Cout) d[j] = v[j];
      }
    }
  }
  if (a.stats) {
    // per-tile (sum, centred M2) per output channel over the tile's valid pixels
    const int nvalid = tile_ok ? min(THO, a.Ho - oy0) * min(TWO, a.Wo - ox0) : 1;
    float s1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = pix_ok ? v[j] : 0.f;
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      s1[j] = s;
    }
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sstat[wave][0][lane * 4 + j] = s1[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tot = sstat[0][0][cq + j] + sstat[1][0][cq + j] + sstat[2][0][cq + j] + sstat[3][0][cq + j];
      const float mean = tot / (float)nvalid;
      float d = pix_ok ? (v[j] - mean) : 0.f;
      d = d * d;
      d += __shfl_xor(d, 8, 64);
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      m2[j] = d;
      s1[j] = tot;
    }
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sstat[wave][1][lane * 4 + j] = m2[j];
    }
    __syncthreads();
    if (tile_ok && tid < 32 && nf * 32 + tid < a.Cout) {
      const float tot = sstat[0][0][tid] + sstat[1][0][tid] + sstat[2][0][tid] + sstat[3][0][tid];
      const float m2t = sstat[0][1][tid] + sstat[1][1][tid] + sstat[2][1][tid] + sstat[3][1][tid];
      float* st = a.stats + (size_t)mt * 2 * a.Cout;
      st[nf * 32 + tid] = tot;
      st[a.Cout + nf * 32 + tid] = m2t;
      if (tid == 0 && nf == 0) a.stats_cnt[mt] = (float)nvalid;
    }
  }
  if (a.epi_partial) {
    // backward partials for the BatchNorm / activation that produced this conv's consumer-side input (see Conv3Args)
    const float eslope = a.epi_slope ? a.epi_slope[0] : a.epi_slope_const;
    // epi_y is addressed like the STORED tensor: for the parity-class store that is the full-resolution pixel
    const size_t ebase = a.out_mode == OUT_STRIDE2
                             ? (((size_t)b * a.Hy + 2 * oy + a.sub_y) * a.Wy + 2 * ox + a.sub_x) * a.Cout + n0
                             : obase;
    float q[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = 0.f, yv = 0.f, z = 0.f;
      if (pix_ok && n0 + j < a.Cout) {
        g = v[j];
        yv = a.epi_y[ebase + j];
        z = a.epi_scale ? fmaf(yv, a.epi_scale[n0 + j], a.epi_shift[n0 + j]) : yv;
      }
      float gz = g;
      float q2 = 0.f;
      if (a.epi_act) {
        q2 = g * fminf(z, 0.f);
        gz = z > 0.f ? g : g * eslope;
      }
      q[0][j] = gz;
      q[1][j] = gz * yv;
      q[2][j] = q2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = q[k][j];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        q[k][j] = t;
      }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sstat[wave][k][lane * 4 + j] = q[k][j];
    }
    __syncthreads();
    if (tile_ok && tid < 96) {
      const int k = tid >> 5, c = tid & 31;
      if (nf * 32 + c < a.Cout)
        a.epi_partial[((size_t)mt * 3 + k) * a.Cout + nf * 32 + c] =
            sstat[0][k][c] + sstat[1][k][c] + sstat[2][k][c] + sstat[3][k][c];
    }
  }
}

}  // namespace

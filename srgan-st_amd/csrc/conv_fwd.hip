// KSxKS convolution (KS = 3 or 9, pad KS/2, stride 1 or 2), NHWC, fp32 MFMA implicit GEMM - forward and data-gradient.
//
// Replaces the cuDNN/oneDNN calls behind (reference file:line)
//   _ResidualConvBlock convs            model.py:173,176      (64->64, no bias)
//   Generator.conv2                     model.py:113
//   _UpsampleBlock conv                 model.py:159          (64->256, bias)
//   Discriminator.features convs        model.py:32-56        (stride 1/2, up to 512 channels)
//   Generator.conv1 / conv3 (9x9)       model.py:101,127      (+ PixelShuffle model.py:160 and clamp model.py:150
//                                                              folded into the store)
// and their autograd data-gradients (dgrad of a stride-1 conv = the same kernel on weights packed
// with mode 1: transposed + rotated by 180 degrees).
//
// GEMM view: M = output pixels (B*Ho*Wo), N = Cout, K = KS*KS*Cin.
// Workgroup = 256 threads = 4 waves; output tile = 32 pixels (8 wide x 4 high) x 32 channels:
// ONE v_mfma_f32_32x32x2_f32 accumulator per wave, the 4 waves split K and are summed through LDS.
//   A (pixels x k): input patch (tile + halo) x 64-channel block staged ONCE in LDS, re-used by all KS*KS taps;
//                   the producer's BatchNorm-apply + PReLU/LeakyReLU is applied while staging.
//   B (k x cout)  : pre-packed weights, each lane loads its 16-B fragment straight from L2 (1 KiB per wave-load).
// Epilogue: + bias, + residual, per-tile BatchNorm partial statistics (sum, centred M2; combined
// with Chan's formula by bn_finalize - no atomics, bit-reproducible).
#include "conv_common.h"

namespace {

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
  int out_mode;            // OUT_*
  int B, H, W, Cin, Cout, Ho, Wo;
};

// how the epilogue stores the [B,Ho,Wo,Cout] result
enum : int {
  OUT_NHWC = 0,
  OUT_SHUFFLE = 1,      // PixelShuffle(2): y[b, 2oy+i, 2ox+j, c] = out[b,oy,ox,4c+2i+j]      (model.py:160)
  OUT_NCHW_CLAMP = 2,   // y[b,co,oy,ox] = clamp(out,0,1), y_pre = out                          (model.py:148-150)
  OUT_UNSHUFFLE = 3,    // inverse of OUT_SHUFFLE: y[b, oy/2, ox/2, 4c + 2(oy&1) + (ox&1)] = out[b,oy,ox,c]
};

template <int KS, int S>
__global__ __launch_bounds__(CONV_NT) void conv_fwd_kernel(Conv3Args a) {
  constexpr int KK = KS * KS, PAD = KS / 2;
  constexpr int PW = (TWO - 1) * S + KS, PH = (THO - 1) * S + KS, NP = PW * PH;
  constexpr int LDS_FLOATS = (NP * LDSC > 4 * 32 * 33) ? NP * LDSC : 4 * 32 * 33;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  __shared__ float sstat[4][2][32];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_x = (a.Wo + TWO - 1) / TWO, tiles_y = (a.Ho + THO - 1) / THO;
  const int mt = blockIdx.x;
  const int b = mt / (tiles_x * tiles_y), rt = mt - b * tiles_x * tiles_y;
  const int oy0 = (rt / tiles_x) * THO, ox0 = (rt % tiles_x) * TWO;
  const int nf = blockIdx.y;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
  const int ncb = (a.Cin + CB - 1) / CB;
  const int li = lane & 31, lh = lane >> 5;
  const int a_base = (((li >> 3) * S) * PW + (li & 7) * S) * LDSC + 4 * lh;
  const bool vec_ok = (a.Cin & 3) == 0;
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int cb = 0; cb < ncb; ++cb) {
    const int c0 = cb * CB;
    const int cin_blk = min(CB, a.Cin - c0);
    const int nks = (cin_blk + 7) >> 3;
    const int nchunks = KK * nks;
    if (cb) __syncthreads();
    // ---- stage the input patch for this channel block (zero padding stays zero: transform only in-image pixels)
    for (int q = tid; q < NP * 16; q += CONV_NT) {
      const int p = q >> 4, c4 = (q & 15) * 4;
      const int py = p / PW, px = p - py * PW;
      const int iy = iy0 + py, ix = ix0 + px, c = c0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c < a.Cin) {
        const float* src = a.x + (((size_t)b * a.H + iy) * a.W + ix) * a.Cin + c;
        if (vec_ok) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < a.Cin) v[j] = src[j];
        }
        if (a.in_scale) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < a.Cin) v[j] = fmaf(v[j], a.in_scale[c + j], a.in_shift[c + j]);
        }
        if (a.in_act == ACT_SLOPE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
        }
      }
      *reinterpret_cast<f32x4*>(&lds[p * LDSC + c4]) = v;
    }
    __syncthreads();

    // ---- this wave's share of the K chunks (chunk = one tap x 8 input channels = 4 MFMAs)
    const int cbeg = (nchunks * wave) >> 2, cend = (nchunks * (wave + 1)) >> 2;
    const float* wblk = a.wp + ((size_t)(nf * ncb + cb) * KK * 8) * 256 + lane * 4;
    auto bptr = [&](int c) {
      c = c < cend ? c : cend - 1;  // clamp: never read past this wave's range
      const int tap = c / nks, ks = c - tap * nks;
      return reinterpret_cast<const f32x4*>(wblk + (size_t)(tap * 8 + ks) * 256);
    };
    auto aread = [&](int c) {
      const int tap = c / nks, ks = c - tap * nks;
      const int dy = tap / KS, dx = tap - dy * KS;
      return *reinterpret_cast<const f32x4*>(&lds[a_base + (dy * PW + dx) * LDSC + ks * 8]);
    };
    if (cbeg < cend) {
      f32x4 b0 = *bptr(cbeg), b1 = *bptr(cbeg + 1), b2 = *bptr(cbeg + 2), b3 = *bptr(cbeg + 3);
      for (int c = cbeg; c < cend; c += 4) {
        {
          const f32x4 av = aread(c);
          const f32x4 bn = *bptr(c + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0[j], acc, 0, 0, 0);
          b0 = bn;
        }
        if (c + 1 < cend) {
          const f32x4 av = aread(c + 1);
          const f32x4 bn = *bptr(c + 5);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1[j], acc, 0, 0, 0);
          b1 = bn;
        }
        if (c + 2 < cend) {
          const f32x4 av = aread(c + 2);
          const f32x4 bn = *bptr(c + 6);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b2[j], acc, 0, 0, 0);
          b2 = bn;
        }
        if (c + 3 < cend) {
          const f32x4 av = aread(c + 3);
          const f32x4 bn = *bptr(c + 7);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b3[j], acc, 0, 0, 0);
          b3 = bn;
        }
      }
    }
  }

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
    } else {  // OUT_UNSHUFFLE
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < a.Cout)
          a.y[(((size_t)b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * (4 * a.Cout) + 4 * (n0 + j) +
              2 * (oy & 1) + (ox & 1)] = v[j];
    }
  }
  if (a.stats) {
    // per-tile (sum, centred M2) per output channel over the tile's valid pixels
    const int nvalid = min(THO, a.Ho - oy0) * min(TWO, a.Wo - ox0);
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
    if (tid < 32 && nf * 32 + tid < a.Cout) {
      const float tot = sstat[0][0][tid] + sstat[1][0][tid] + sstat[2][0][tid] + sstat[3][0][tid];
      const float m2t = sstat[0][1][tid] + sstat[1][1][tid] + sstat[2][1][tid] + sstat[3][1][tid];
      float* st = a.stats + (size_t)mt * 2 * a.Cout;
      st[nf * 32 + tid] = tot;
      st[a.Cout + nf * 32 + tid] = m2t;
      if (tid == 0 && nf == 0) a.stats_cnt[mt] = (float)nvalid;
    }
  }
}

// w [Cout][Cin][3][3] (reference layout) -> packed.  mode 0: forward.  mode 1: data-gradient of a stride-1
// conv (outputs = Cin, inputs = Cout, taps rotated 180 degrees).
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KK,
                                 int mode, int64_t total) {
  const int O = mode ? Cin : Cout, I = mode ? Cout : Cin;
  const int ncb = (I + 63) / 64;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    // decode packed index
    const int j = idx & 3;
    const int l = (idx >> 2) & 63;
    const int ks = (idx >> 8) & 7;
    int64_t rest = idx >> 11;
    const int tap = rest % KK;
    rest /= KK;
    const int cbk = rest % ncb;
    const int of = rest / ncb;
    const int o = of * 32 + (l & 31);
    const int i = cbk * 64 + ks * 8 + (l >> 5) * 4 + j;
    float v = 0.f;
    if (o < O && i < I) {
      if (mode == 0)
        v = w[((size_t)o * Cin + i) * KK + tap];
      else
        v = w[((size_t)i * Cin + o) * KK + (KK - 1 - tap)];
    }
    wp[idx] = v;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int64_t sst_conv_packed_floats(int Cout, int Cin, int ksize) { return packed_floats(Cout, Cin, ksize * ksize); }

SST_API int sst_conv_pack(const float* w, float* wp, int Cout, int Cin, int ksize, int mode, void* stream) {
  SST_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 3 || ksize == 9) && (mode == 0 || mode == 1),
              "sst_conv_pack: bad argument");
  const int O = mode ? Cin : Cout, I = mode ? Cout : Cin;
  const int64_t total = packed_floats(O, I, ksize * ksize);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  pack_conv_kernel<<<blocks, 256, 0, sst_stream(stream)>>>(w, wp, Cout, Cin, ksize * ksize, mode, total);
  SST_LAUNCH_CHECK("pack_conv_kernel");
  return SST_OK;
}

SST_API int sst_conv_mtiles(int B, int Ho, int Wo) { return B * ((Ho + THO - 1) / THO) * ((Wo + TWO - 1) / TWO); }

// y = conv(act(x*in_scale+in_shift), w) (+bias) (+residual), stored per out_mode; optional BN partial stats.
SST_API int sst_conv_fwd(const float* x, const float* wp, float* y, float* y_pre, const float* bias, const float* in_scale,
                         const float* in_shift, const float* in_slope, float in_slope_const, int in_act,
                         const float* residual, float* stats, float* stats_cnt, int out_mode, int B, int H, int W, int Cin,
                         int Cout, int ksize, int stride, void* stream) {
  SST_REQUIRE(x && wp && y, "sst_conv_fwd: null pointer");
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) && (ksize == 3 || ksize == 9),
              "sst_conv_fwd: bad shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d stride=%d", B, H, W, Cin, Cout, ksize, stride);
  SST_REQUIRE(!(ksize == 9 && stride == 2), "sst_conv_fwd: 9x9 stride 2 not built");
  SST_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sst_conv_fwd: in_scale/in_shift must come together");
  SST_REQUIRE(!stats || stats_cnt, "sst_conv_fwd: stats needs stats_cnt");
  SST_REQUIRE(out_mode >= 0 && out_mode <= 3, "sst_conv_fwd: bad out_mode");
  SST_REQUIRE(out_mode != OUT_SHUFFLE || (Cout & 3) == 0, "sst_conv_fwd: shuffle store needs Cout %% 4 == 0");
  SST_REQUIRE(out_mode == OUT_NHWC || (!residual && !stats), "sst_conv_fwd: residual/stats only with NHWC store");
  Conv3Args a;
  a.x = x; a.wp = wp; a.y = y; a.y_pre = y_pre; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift;
  a.in_slope = in_slope; a.in_slope_const = in_slope_const; a.in_act = in_act; a.residual = residual; a.stats = stats;
  a.stats_cnt = stats_cnt; a.out_mode = out_mode;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Ho = (H + 2 * (ksize / 2) - ksize) / stride + 1;
  a.Wo = (W + 2 * (ksize / 2) - ksize) / stride + 1;
  SST_REQUIRE(out_mode != OUT_UNSHUFFLE || ((a.Ho & 1) == 0 && (a.Wo & 1) == 0), "sst_conv_fwd: unshuffle needs even Ho,Wo");
  const int64_t mt = sst_conv_mtiles(B, a.Ho, a.Wo);
  SST_REQUIRE(mt < (1ll << 31), "sst_conv_fwd: too many tiles");
  dim3 grid((unsigned)mt, (Cout + 31) / 32);
  hipStream_t st = sst_stream(stream);
  if (ksize == 3 && stride == 1)
    conv_fwd_kernel<3, 1><<<grid, CONV_NT, 0, st>>>(a);
  else if (ksize == 3)
    conv_fwd_kernel<3, 2><<<grid, CONV_NT, 0, st>>>(a);
  else
    conv_fwd_kernel<9, 1><<<grid, CONV_NT, 0, st>>>(a);
  SST_LAUNCH_CHECK("conv_fwd_kernel");
  return SST_OK;
}

// 3x3 convolution, big-tile variant for the large layers (up-sampler, discriminator, VGG19): workgroup tile =
// 64 pixels (8x8) x 64 output channels, i.e. 2 x 2 v_mfma_f32_32x32x2_f32 accumulators per wave, so that ONE pair of
// LDS A-fragments and ONE pair of L2 B-fragments feeds 16 MFMAs (the 32x32 kernel of conv_fwd.hip feeds 4).
// Same GEMM view, same packed weights, same K split over the 4 waves, same epilogue (conv_epilogue.h, run once per
// 32x32 sub-tile).  4x fewer workgroups (dispatch cost), 2.4x less halo staging, half the weight traffic per FLOP.
// Chosen by sst_conv_fwd / sst_conv_dgrad_bwdstats / sst_conv_s2_dgrad when the layer has >= BIG_MIN_TILES such tiles.
#include "conv_epilogue.h"

namespace {

constexpr int T2 = 8;             // 8 x 8 output pixels

template <int S>
__global__ __launch_bounds__(CONV_NT, 3) void conv_fwd2_kernel(Conv3Args a) {   // <= 168 VGPRs: three workgroups per CU
  constexpr int KS = 3;
  constexpr int PW = (T2 - 1) * S + KS, PH = PW, NP = PW * PH;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // max(NP*LDSC, 4*32*33) floats
  __shared__ float sstat[4][3][32];
  const int KK = a.ksy * a.ksx;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tiles_x = (a.Wo + T2 - 1) / T2, tiles_y = (a.Ho + T2 - 1) / T2;
  const int mtb = blockIdx.x;
  const int b = mtb / (tiles_x * tiles_y), rt = mtb - b * tiles_x * tiles_y;
  const int ty = rt / tiles_x, tx = rt - ty * tiles_x;
  const int oy0 = ty * T2, ox0 = tx * T2;
  const int nf0 = blockIdx.y * 2, nfblocks = (a.Cout + 31) >> 5;
  const int iy0 = oy0 * S - a.pad_y, ix0 = ox0 * S - a.pad_x;
  const int ncb = (a.Cin + CB - 1) / CB;
  const int li = lane & 31, lh = lane >> 5;
  int a_base[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) a_base[mf] = (((mf * 4 + (li >> 3)) * S) * PW + (li & 7) * S) * LDSC + 4 * lh;
  const bool vec_ok = (a.Cin & 3) == 0;
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  for (int cb = 0; cb < ncb; ++cb) {
    const int c0 = cb * CB;
    const int cin_blk = min(CB, a.Cin - c0);
    const int nks = (cin_blk + 7) >> 3;
    const int nchunks = KK * nks;
    if (cb) __syncthreads();
    const int cbeg = (nchunks * wave) >> 2, cend = (nchunks * (wave + 1)) >> 2;
    const int nmine = cend - cbeg;
    const float* wzero = a.wp + (packed_floats_base(a.Cout, a.Cin, KK) - PACK_PAD) + lane * 4;
    const float* wblk0 = a.wp + ((size_t)(nf0 * ncb + cb) * KK * 8) * 256 + lane * 4;
    const float* wblk1 = (nf0 + 1 < nfblocks) ? a.wp + ((size_t)((nf0 + 1) * ncb + cb) * KK * 8) * 256 + lane * 4 : nullptr;
    int a_ks, a_dx, a_off, a_i = 0;
    int p_ks, p_off, p_i = 0;
    {
      const int tap = cbeg / nks;
      a_ks = cbeg - tap * nks;
      const int dy = tap / a.ksx;
      a_dx = tap - dy * a.ksx;
      a_off = (dy * PW + a_dx) * LDSC + a_ks * 8;
      p_ks = a_ks;
      p_off = (tap * 8 + a_ks) * 256;
    }
    struct BPair { f32x4 n0, n1; };
    auto pf_load = [&]() {
      const bool live = p_i < nmine;
      BPair v;
      v.n0 = *reinterpret_cast<const f32x4*>(live ? wblk0 + p_off : wzero);
      v.n1 = *reinterpret_cast<const f32x4*>((live && wblk1) ? wblk1 + p_off : wzero);
      const bool wrap = (p_ks + 1 == nks);
      p_ks = wrap ? 0 : p_ks + 1;
      p_off += wrap ? (9 - nks) * 256 : 256;
      ++p_i;
      return v;
    };
    struct APair { f32x4 m0, m1; };
    auto a_load = [&]() {
      APair v;
      v.m0 = *reinterpret_cast<const f32x4*>(&lds[a_base[0] + a_off]);
      v.m1 = *reinterpret_cast<const f32x4*>(&lds[a_base[1] + a_off]);
      const bool live = a_i + 1 < nmine;
      const bool wrap = (a_ks + 1 == nks);
      const bool wrapx = wrap && (a_dx + 1 == a.ksx);
      const int step = wrap ? (LDSC - 8 * (nks - 1)) + (wrapx ? (PW - a.ksx) * LDSC : 0) : 8;
      a_off += live ? step : 0;
      a_ks = wrap ? 0 : a_ks + 1;
      a_dx = wrapx ? 0 : (wrap ? a_dx + 1 : a_dx);
      ++a_i;
      return v;
    };
    BPair A0 = pf_load(), A1 = pf_load(), A2 = pf_load(), B0, B1, B2;

    // ---- stage the input patch (zero padding stays zero: transform only in-image pixels)
    {
      const int c4 = (tid & 15) * 4, c = c0 + c4;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (a.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < a.Cin) {
            sc[j] = a.in_scale[c + j];
            sh[j] = a.in_shift[c + j];
          }
      }
      for (int p = tid >> 4; p < NP; p += CONV_NT / 16) {
        const int py = p / PW, px = p - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
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
            for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
          }
          if (a.in_act == ACT_SLOPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
          }
          if (!vec_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (c + j >= a.Cin) v[j] = 0.f;
          }
        }
        *reinterpret_cast<f32x4*>(&lds[p * LDSC + c4]) = v;
      }
    }
    __syncthreads();

#define SST_CHUNK2(BUSE, BLOAD)                                                                                \
    {                                                                                                          \
      const APair av = a_load();                                                                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.m0[j], BUSE.n0[j], acc[0][0], 0, 0, 0);           \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.m0[j], BUSE.n1[j], acc[0][1], 0, 0, 0);           \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.m1[j], BUSE.n0[j], acc[1][0], 0, 0, 0);           \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.m1[j], BUSE.n1[j], acc[1][1], 0, 0, 0);           \
      }                                                                                                        \
      BLOAD = pf_load();                                                                                       \
    }
    if (nmine > 0) {
      for (int c = 0; c < nmine; c += 6) {
        SST_CHUNK2(A0, B0)
        SST_CHUNK2(A1, B1)
        SST_CHUNK2(A2, B2)
        SST_CHUNK2(B0, A0)
        SST_CHUNK2(B1, A1)
        SST_CHUNK2(B2, A2)
      }
    }
#undef SST_CHUNK2
  }

  // ---- epilogue: the four 32x32 sub-tiles one after the other (statistics tiles follow the 8x4 numbering)
  const int tiles_y4 = (a.Ho + THO - 1) / THO;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int oy = oy0 + mf * 4;
    const int ty4 = ty * 2 + mf;
    const bool tile_ok = ty4 < tiles_y4 && oy < a.Ho && ox0 < a.Wo;
    const int mt = (b * tiles_y4 + (tile_ok ? ty4 : 0)) * tiles_x + tx;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      if (nf0 + n < nfblocks)      // workgroup-uniform
        conv_tile_epilogue(a, lds, sstat, acc[mf][n], b, oy, ox0, nf0 + n, mt, tile_ok, tid, wave, lane);
    }
  }
}

size_t big_lds_bytes(int stride) {
  const int pw = (T2 - 1) * stride + 3, np = pw * pw;
  const int f = np * LDSC > 4 * 32 * 33 ? np * LDSC : 4 * 32 * 33;
  return (size_t)f * sizeof(float);
}

}  // namespace

static long g_big_launches = 0;
SST_API long sst_debug_big_tile_launches(void) { return g_big_launches; }   // test hook: how often the 64x64 kernel was chosen

// Launches the 64x64-tile kernel for the arguments prepared by conv_fwd.hip (KS = 3 only).
int sst_launch_conv_fwd2(const Conv3Args& a, int stride, hipStream_t st) {
  ++g_big_launches;
  const int tiles = a.B * ((a.Ho + T2 - 1) / T2) * ((a.Wo + T2 - 1) / T2);
  dim3 grid((unsigned)tiles, ((a.Cout + 31) / 32 + 1) / 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)big_lds_bytes(2));
    attr_set = true;
  }
  if (stride == 1)
    conv_fwd2_kernel<1><<<grid, CONV_NT, big_lds_bytes(1), st>>>(a);
  else
    conv_fwd2_kernel<2><<<grid, CONV_NT, big_lds_bytes(2), st>>>(a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SST_OK : sst_set_error(SST_ERR_HIP, "conv_fwd2_kernel: %s", hipGetErrorString(e));
}

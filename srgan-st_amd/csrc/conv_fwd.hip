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
#include <cstdlib>
#include "conv_epilogue.h"

// 64x64-tile variant (conv_fwd2.hip)
int sst_launch_conv_fwd2(const Conv3Args& a, int stride, hipStream_t st);
// band kernel for the 64-input-channel trunk shape (conv_band.hip)
int sst_launch_conv_band(const Conv3Args& a, int R, hipStream_t st, const BandAcc* acc);
int sst_conv_band_rows(int B, int H, int W, int Cin, int Cout, int ksize, int stride);

namespace {

// Use the 64 px x 64 ch tile kernel when the layer still yields enough workgroups to fill the chip with it.
constexpr int BIG_MIN_TILES = 2048;   // measured (tools/ablate_big.py): +9 % at >= 2304 tiles, slower below ~1500
inline bool use_big_tiles(const Conv3Args& a, int ksize) {
  if (ksize != 3 || a.Cout < 64 || a.in2 || a.side_out) return false;
  if (a.out_mode != OUT_NHWC && a.out_mode != OUT_SHUFFLE && a.out_mode != OUT_STRIDE2) return false;
  if (const char* e = sst_env("SST_CONV_BIG")) return atoi(e) != 0;       // dev override
  const long tiles = (long)a.B * ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8) * ((a.Cout + 63) / 64);
  return tiles >= BIG_MIN_TILES;
}

template <int KS, int S>
__device__ __forceinline__ void conv_fwd_body(const Conv3Args& a, const int bx, const int by) {
  const int KK = a.ksy * a.ksx;
  constexpr int PW = (TWO - 1) * S + KS, PH = (THO - 1) * S + KS, NP = PW * PH;
  constexpr int LDS_FLOATS = (NP * LDSC > 4 * 32 * 33) ? NP * LDSC : 4 * 32 * 33;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  __shared__ float sstat[4][3][32];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave id in an SGPR: everything derived from it stays scalar
  const int tiles_x = (a.Wo + TWO - 1) / TWO, tiles_y = (a.Ho + THO - 1) / THO;
  const int mt = bx;
  const int b = mt / (tiles_x * tiles_y), rt = mt - b * tiles_x * tiles_y;
  const int oy0 = (rt / tiles_x) * THO, ox0 = (rt % tiles_x) * TWO;
  const int nf = by;
  const int iy0 = oy0 * S - a.pad_y, ix0 = ox0 * S - a.pad_x;
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
    // ---- this wave's share of the K chunks (chunk = one tap x 8 input channels = 4 MFMAs).
    // The loop body is ONE basic block of 6 chunks with two register sets (ping-pong, prefetch distance 3
    // chunks) and no register moves, so hipcc keeps counted vmcnt waits.  A wave's chunk count is rounded
    // up to a multiple of 6; surplus chunks read their B fragment from the zero pad behind the packed
    // weights (pointer select, no branch) and so add nothing.
    const int cbeg = (nchunks * wave) >> 2, cend = (nchunks * (wave + 1)) >> 2;
    const int nmine = cend - cbeg;
    const float* wblk = a.wp + ((size_t)(nf * ncb + cb) * KK * 8) * 256 + lane * 4;
    const float* wzero = a.wp + (packed_floats_base(a.Cout, a.Cin, KK) - PACK_PAD) + lane * 4;
    // cursors (wave-uniform scalars): A = chunk being multiplied, P = chunk being prefetched
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
    auto pf_load = [&]() {
      const float* src = p_i < nmine ? wblk + p_off : wzero;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      const bool wrap = (p_ks + 1 == nks);
      p_ks = wrap ? 0 : p_ks + 1;
      p_off += wrap ? (9 - nks) * 256 : 256;
      ++p_i;
      return v;
    };
    auto a_load = [&]() {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[a_base + a_off]);
      const bool live = a_i + 1 < nmine;               // never walk the LDS cursor past the last real chunk
      const bool wrap = (a_ks + 1 == nks);
      const bool wrapx = wrap && (a_dx + 1 == a.ksx);
      const int step = wrap ? (LDSC - 8 * (nks - 1)) + (wrapx ? (PW - a.ksx) * LDSC : 0) : 8;
      a_off += live ? step : 0;
      a_ks = wrap ? 0 : a_ks + 1;
      a_dx = wrapx ? 0 : (wrap ? a_dx + 1 : a_dx);
      ++a_i;
      return v;
    };
    // first B fragments go out BEFORE the patch is staged: their L2 latency hides behind the staging
    f32x4 A0 = pf_load(), A1 = pf_load(), A2 = pf_load(), B0, B1, B2;

    // ---- stage the input patch for this channel block (zero padding stays zero: transform only in-image pixels).
    // A thread's channel quad is the same for all its patch pixels (CONV_NT % 16 == 0): scale/shift live in registers.
    {
      const int c4 = (tid & 15) * 4, c = c0 + c4;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      f32x4 kA = {1.f, 1.f, 1.f, 1.f}, kB = {0.f, 0.f, 0.f, 0.f}, kC = {0.f, 0.f, 0.f, 0.f};
      if (a.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < a.Cin) {
            sc[j] = a.in_scale[c + j];
            sh[j] = a.in_shift[c + j];
          }
      }
      if (a.in_cA) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < a.Cin) {
            kA[j] = a.in_cA[c + j];
            kB[j] = a.in_cB[c + j];
            kC[j] = a.in_cC[c + j];
          }
      }
      if (S == 2 && vec_ok && !a.in2) {
        // Stride 2 (10 patch pixels per thread, LDS-limited to 3 workgroups per CU): batches of 5 pixels with all global loads
        // of a batch issued before the first is consumed; branch-free (slots outside the image / past NP read the tensor's
        // first pixel and are zeroed) so that the vmcnt waits stay counted.
        constexpr int PPT = CONV_NT / 16;
        constexpr int NIT = (NP + PPT - 1) / PPT;
        constexpr int UNB = 5;
        const int nit = (a.dbg & 1) ? 0 : NIT;
        for (int u0 = 0; u0 < nit; u0 += UNB) {
          f32x4 v[UNB];
          bool ok[UNB];
#pragma unroll
          for (int u = 0; u < UNB; ++u) {
            const int p = (tid >> 4) + (u0 + u) * PPT;
            const int py = p / PW, px = p - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            ok[u] = p < NP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c < a.Cin;
            const size_t off = ok[u] ? (((size_t)b * a.H + iy) * a.W + ix) * a.Cin + c : (size_t)c4;
            v[u] = *reinterpret_cast<const f32x4*>(a.x + off);
          }
#pragma unroll
          for (int u = 0; u < UNB; ++u) {
            const int p = (tid >> 4) + (u0 + u) * PPT;
            if (p >= NP) continue;
            f32x4 t = v[u];
            if (a.in_scale) {
#pragma unroll
              for (int j = 0; j < 4; ++j) t[j] = fmaf(t[j], sc[j], sh[j]);
            }
            if (a.in_act == ACT_SLOPE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : t[j] * slope;
            }
            if (!ok[u]) t = f32x4{0.f, 0.f, 0.f, 0.f};          // zero padding stays exactly zero
            *reinterpret_cast<f32x4*>(&lds[p * LDSC + c4]) = t;
          }
        }
      } else
      for (int p = tid >> 4; p < ((a.dbg & 1) ? 0 : NP); p += CONV_NT / 16) {
        const int py = p / PW, px = p - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c < a.Cin) {
          const size_t off = (((size_t)b * a.H + iy) * a.W + ix) * a.Cin + c;
          const float* src = a.x + off;
          if (vec_ok) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (c + j < a.Cin) v[j] = src[j];
          }
          if (a.in2) {                       // fused BN-backward apply (vec_ok is required by the host wrapper)
            const f32x4 yv = *reinterpret_cast<const f32x4*>(a.in2 + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float gz = v[j];
              if (a.in_act == ACT_SLOPE) {
                const float z = fmaf(yv[j], sc[j], sh[j]);
                gz = z > 0.f ? gz : gz * slope;
              }
              v[j] = a.in_cA ? fmaf(kA[j], gz, fmaf(kB[j], yv[j], kC[j])) : gz;
            }
            // the tile's own pixels (S == 1: patch interior) are written once, by the nf == 0 workgroup
            if (a.side_out && nf == 0 && py >= a.pad_y && py < a.pad_y + THO && px >= a.pad_x && px < a.pad_x + TWO)
              *reinterpret_cast<f32x4*>(a.side_out + off) = v;
          } else {
            if (a.in_scale) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
            }
            if (a.in_act == ACT_SLOPE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
            }
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

#define SST_CHUNK(ACUR, ANEXT, BUSE, BLOAD)                                                            \
    {                                                                                                  \
      ANEXT = a_load();          /* next chunk's A fragment: its LDS latency hides behind these MFMAs */ \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ACUR[j], BUSE[j], acc, 0, 0, 0);                  \
      BLOAD = pf_load();                                                                               \
    }
    if (nmine > 0 && !(a.dbg & 2)) {
      f32x4 a0 = a_load(), a1;
      for (int c = 0; c < nmine; c += 6) {
        SST_CHUNK(a0, a1, A0, B0)
        SST_CHUNK(a1, a0, A1, B1)
        SST_CHUNK(a0, a1, A2, B2)
        SST_CHUNK(a1, a0, B0, A0)
        SST_CHUNK(a0, a1, B1, A1)
        SST_CHUNK(a1, a0, B2, A2)
      }
    }
#undef SST_CHUNK
  }

  if (a.dbg & 4) {
    if (acc[0] == 12345.f) a.y[0] = acc[0];
    return;
  }
  conv_tile_epilogue(a, lds, sstat, acc, b, oy0, ox0, nf, mt, true, tid, wave, lane);
}

template <int KS, int S>
__global__ __launch_bounds__(CONV_NT, (KS == 3 && S == 1 ? 5 : 3)) void conv_fwd_kernel(Conv3Args a) {   // 3x3 s1: <= 96 VGPRs, 5 workgroups per CU (measured +5..10 % on the dgrad shapes); the others are LDS-limited to 3
  conv_fwd_body<KS, S>(a, blockIdx.x, blockIdx.y);
}

// Data-gradient of a stride-2 conv: the 4 parity classes of the input-gradient pixels (1 / 2 / 2 / 4 taps) in ONE launch
// (blockIdx.z = class) instead of four short ones - the classes share nothing but run side by side and there is one launch
// tail instead of four.
struct S2Classes {
  long long wp_off[4];
  int nh[4], nw[4], tiles[4];
  int tile_base[4];          // first partial tile of the class (epi_partial is [sum of tiles][3][Cout])
};
__global__ __launch_bounds__(CONV_NT, 5) void conv_s2dgrad_kernel(Conv3Args a, S2Classes c) {
  const int cls = blockIdx.z;
  if ((int)blockIdx.x >= c.tiles[cls]) return;
  a.wp += c.wp_off[cls];
  a.ksy = 1 + (cls >> 1);
  a.ksx = 1 + (cls & 1);
  a.sub_y = cls >> 1;
  a.sub_x = cls & 1;
  a.Ho = c.nh[cls];
  a.Wo = c.nw[cls];
  if (cls) a.side_out = nullptr;                 // every class stages all dY pixels: class 0 writes the side output
  if (a.epi_partial) a.epi_partial += (size_t)c.tile_base[cls] * 3 * a.Cout;
  conv_fwd_body<3, 1>(a, blockIdx.x, blockIdx.y);
}

// Data-gradient of a stride-2 conv, all 4 parity classes of one 8x4 block of class pixels (= a 16x8 block of dX) in ONE
// workgroup: the classes read the same (8+1) x (4+1) patch of dY, so it is staged once instead of four times, and a
// workgroup runs 9 (class, tap) x 8 k-steps = 288 MFMAs per 64-channel block of dY instead of 32..128 (the per-class
// launch above spends most of its time staging).  Plain form only (no fused BatchNorm-backward input / epilogue sums), even
// H and W (all classes share one nh x nw grid), channels of dY a multiple of 4.
// K split: wave w takes k-steps {2w, 2w+1} of every (class, tap) - the accumulator of a class is a compile-time index.
// FUSED: the BatchNorm-backward stage around the data-gradient (sst_conv_s2_dgrad_fused): the staged value is
// dy = cA*gz + cB*y2 + cC with gz = g * act'(y2*in_scale + in_shift), written once to side_out; epilogue sums into epi_partial.
template <bool FUSED>
__global__ __launch_bounds__(CONV_NT, 3) void conv_s2dgrad4_kernel(Conv3Args a, S2Classes c) {
  constexpr int PW = TWO + 1, PH = THO + 1, NP = PW * PH;
  constexpr int LDS_FLOATS = (NP * LDSC > 4 * 32 * 33) ? NP * LDSC : 4 * 32 * 33;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  __shared__ float sstat[4][3][32];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int nh = c.nh[0], nw = c.nw[0];
  const int tiles_x = (nw + TWO - 1) / TWO, tiles_y = (nh + THO - 1) / THO;
  const int mt = blockIdx.x;
  const int b = mt / (tiles_x * tiles_y), rt = mt - b * tiles_x * tiles_y;
  const int oy0 = (rt / tiles_x) * THO, ox0 = (rt % tiles_x) * TWO;
  const int nf = blockIdx.y;
  const int ncb = (a.Cin + CB - 1) / CB;
  const int a_base = ((li >> 3) * PW + (li & 7)) * LDSC + 4 * lh;

  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // the 18 chunks of a wave per channel block: q -> (combo = q / 2, k-step 2*wave + q % 2); combo -> (class, tap)
  constexpr int CLS_OF[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
  constexpr int TAP_OF[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};
  for (int cb = 0; cb < ncb; ++cb) {
    const int c0 = cb * CB;
    const int nks = (min(CB, a.Cin - c0) + 7) >> 3;
    auto load_b = [&](const int q) {
      const int combo = q >> 1, cls = CLS_OF[combo], tap = TAP_OF[combo];
      const int KKc = (1 + (cls >> 1)) * (1 + (cls & 1));
      const int ks = 2 * wave + (q & 1);
      const float* src = a.wp + c.wp_off[cls] + (((size_t)(nf * ncb + cb) * KKc + tap) * 8 + (ks < nks ? ks : 0)) * 256 + lane * 4;
      return *reinterpret_cast<const f32x4*>(src);
    };
    f32x4 bq[2][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) bq[0][i] = load_b(i);             // in flight while the patch is staged
    if (cb) __syncthreads();
    {
      const int c4 = (tid & 15) * 4, ch = c0 + c4;
      f32x4 v[3], yv[3];
      size_t off[3];
      bool ok[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = (tid >> 4) + u * 16;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 + py, ix = ox0 + px;
        ok[u] = p < NP && iy < a.H && ix < a.W && ch < a.Cin;
        off[u] = ok[u] ? (((size_t)b * a.H + iy) * a.W + ix) * a.Cin + ch : (size_t)c4;
        v[u] = *reinterpret_cast<const f32x4*>(a.x + off[u]);
        if (FUSED) yv[u] = *reinterpret_cast<const f32x4*>(a.in2 + off[u]);
      }
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      f32x4 kA = {1.f, 1.f, 1.f, 1.f}, kB = {0.f, 0.f, 0.f, 0.f}, kC = {0.f, 0.f, 0.f, 0.f};
      float slope = 0.f;
      if (FUSED) {
        slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
        if (ch < a.Cin) {                 // Cin % 4 == 0 (checked by the entry point): the quad is in range as a whole
          if (a.in_scale) {
            sc = *reinterpret_cast<const f32x4*>(a.in_scale + ch);
            sh = *reinterpret_cast<const f32x4*>(a.in_shift + ch);
          }
          if (a.in_cA) {
            kA = *reinterpret_cast<const f32x4*>(a.in_cA + ch);
            kB = *reinterpret_cast<const f32x4*>(a.in_cB + ch);
            kC = *reinterpret_cast<const f32x4*>(a.in_cC + ch);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int p = (tid >> 4) + u * 16;
        if (p >= NP) continue;
        f32x4 t = v[u];
        if (FUSED) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float gz = t[j];
            if (a.in_act == ACT_SLOPE) {
              const float z = fmaf(yv[u][j], sc[j], sh[j]);
              gz = z > 0.f ? gz : gz * slope;
            }
            t[j] = a.in_cA ? fmaf(kA[j], gz, fmaf(kB[j], yv[u][j], kC[j])) : gz;
          }
          // the block's own 8x4 pixels (not the +1 halo row / column) are written once, by the nf == 0 workgroup
          const int py = p / PW, px = p - py * PW;
          if (ok[u] && a.side_out && nf == 0 && py < THO && px < TWO) *reinterpret_cast<f32x4*>(a.side_out + off[u]) = t;
        }
        *reinterpret_cast<f32x4*>(&lds[p * LDSC + c4]) = ok[u] ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (g + 1 < 3) {
#pragma unroll
        for (int i = 0; i < 6; ++i) bq[(g + 1) & 1][i] = load_b((g + 1) * 6 + i);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int q = g * 6 + i, combo = q >> 1, cls = CLS_OF[combo], tap = TAP_OF[combo];
        const int ntx = 1 + (cls & 1), jy = tap / ntx, jx = tap - jy * ntx;
        const int ks = 2 * wave + (q & 1);
        if (ks < nks) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(&lds[a_base + (jy * PW + jx) * LDSC + ks * 8]);
          const f32x4 bv = bq[g & 1][i];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[cls], 0, 0, 0);
        }
      }
    }
  }
  const bool tile_ok = oy0 < nh && ox0 < nw;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    Conv3Args t = a;
    t.sub_y = cls >> 1;
    t.sub_x = cls & 1;
    t.Ho = nh;
    t.Wo = nw;
    t.in2 = nullptr;                   // staging-side fields are not the epilogue's business
    t.side_out = nullptr;
    if (FUSED && a.epi_partial) t.epi_partial = a.epi_partial + (size_t)c.tile_base[cls] * 3 * a.Cout;
    if (!FUSED) t.epi_partial = nullptr;
    conv_tile_epilogue(t, lds, sstat, acc[cls], b, oy0, ox0, nf, mt, tile_ok, tid, wave, lane);
  }
}

// 3x3 / stride 1 / pad 1 conv with a 3-CHANNEL input and bias (Discriminator.features[0], model.py:32: the image enters the
// network).  K = 27: an MFMA tile would be 27/64 deep and the general kernel runs this shape at 7 TFLOP/s; the op is bound by
// writing its 64-channel output (37.7 MB at 96 px, B = 16), so: plain VALU, one workgroup per output row, the 3 input rows in
// LDS, a thread keeps the 27 x 4 weights of its 4 output channels in registers and walks the row's pixels.
// Weights are read from the ordinary packed buffer (packed_index with i < 3).
__global__ __launch_bounds__(CONV_NT) void conv3_c3in_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                             int Cout) {
  extern __shared__ __attribute__((aligned(16))) float xs[];            // [3][(W+2)*3 + 1]
  const int RS = (W + 2) * 3 + 1;
  const int b = blockIdx.x / H, oy = blockIdx.x - b * H;
  for (int i = threadIdx.x; i < 3 * RS; i += CONV_NT) {
    const int r = i / RS, o = i - r * RS;
    const int px = o / 3 - 1, iy = oy - 1 + r;
    float v = 0.f;
    if (o < (W + 2) * 3 && (unsigned)px < (unsigned)W && (unsigned)iy < (unsigned)H)
      v = x[(((size_t)b * H + iy) * W + px) * 3 + (o - (px + 1) * 3)];
    xs[i] = v;
  }
  const int nq = Cout >> 2;                       // output-channel quads; CONV_NT % nq == 0 (host check)
  const int cq = threadIdx.x % nq, pl = threadIdx.x / nq, npl = CONV_NT / nq;
  const int co = cq * 4;
  float w[27][4];                                 // [tap*3 + ci][j]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int j = 0; j < 4; ++j) w[t * 3 + ci][j] = wp[((((co + j) >> 5) * 9 + t) * 512 + ((co + j) & 31)) * 4 + ci];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + co);
  __syncthreads();
  for (int ox = pl; ox < W; ox += npl) {
    f32x4 acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = xs + ky * RS + ox * 3;   // 9 contiguous floats: (kx, ci)
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float v = row[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, w[ky * 9 + q][j], acc[j]);
      }
    }
    *reinterpret_cast<f32x4*>(y + (((size_t)b * H + oy) * W + ox) * Cout + co) = acc;
  }
}

// The 3-channel-input layer above on the matrix cores (round 2; W % 32 == 0, Cout == 64: the discriminator's first layer at 96 /
// 192 px).  M = 64 output channels (two MFMA halves), N = 32 pixels of one row segment, K = (ky, kx, ci) = 27 run as 3 rows x 10
// columns (the 10th column of a row has zero weights): lane (pixel, kk) reads the raw 3-channel patch in LDS at pixel*3 + kk plus
// an immediate per K step - no im2col.  Weights come from LDS (staged once per workgroup, MFMA A layout), the 32 x 64 result goes
// through LDS so that every store instruction writes 1 KB of contiguous output: the launch is bound by writing its output once
// (37.7 MB at B = 16 / 96 px; a plain fill of that size takes 7.5 us on this chip, tools/hbm_probe.py).
constexpr int C3F_ROW = 104;                  // floats per patch row in LDS (34 px x 3 ch = 102, + the 10th-column overrun)
constexpr int C3F_OST = 68;                   // floats per pixel in the output staging (64 channels + pad)
__global__ __launch_bounds__(CONV_NT) void conv3_c3in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                                  int W, int ntiles) {
  __shared__ __attribute__((aligned(16))) float ost[4][32 * C3F_OST];    // per wave: patch rows first, then the output tile
  __shared__ float wl[15 * 2 * 64];                                      // weights in A layout: [(ky, s)][h][lane]
  constexpr int Cout = 64;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  // weights: K step (ky, s) holds columns j = 2s + kk of row ky (j = kx*3 + ci; j = 9: zero) for channel 32h + li
  for (int i = tid; i < 15 * 2 * 64; i += CONV_NT) {
    const int l = i & 63, h = (i >> 6) & 1, ks = i >> 7;
    const int ky = ks / 5, j = 2 * (ks - ky * 5) + (l >> 5);
    wl[i] = j < 9 ? wp[((size_t)(h * 9 + ky * 3 + j / 3) * 512 + (l & 31)) * 4 + j % 3] : 0.f;
  }
  float* const ps = ost[wave];
  const int tpr = W >> 5;
  // persistent: wave w of workgroup g walks tiles 4g + w, + 4 * gridDim.x, ...; the next tile's patch values are loaded into
  // registers before the current tile's MFMAs and written to LDS behind its output stores
  float pv[5];
  auto load_patch = [&](int t) {
    const int b = t / (H * tpr), rem = t - b * (H * tpr);
    const int oy = rem / tpr, x0 = (rem - oy * tpr) << 5;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = lane + 64 * u;
      const int r = i / C3F_ROW, c = i - r * C3F_ROW;
      const int iy = oy - 1 + r, fx = (x0 - 1) * 3 + c;
      const bool ok = i < 3 * C3F_ROW + 8 && r < 3 && c < 102 && (unsigned)iy < (unsigned)H && (unsigned)fx < (unsigned)(W * 3);
      pv[u] = ok ? x[((size_t)b * H + iy) * W * 3 + fx] : 0.f;
    }
  };
  int t = blockIdx.x * 4 + wave;
  if (t < ntiles) load_patch(t);
  __syncthreads();                                         // weights staged
  const float* const bl = ps + li * 3 + lh;
  for (; t < ntiles; t += gridDim.x * 4) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = lane + 64 * u;
      if (i < 3 * C3F_ROW + 8) ps[i] = pv[u];
    }
    const int tn = t + gridDim.x * 4;
    if (tn < ntiles) load_patch(tn);
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int s2 = 0; s2 < 5; ++s2) {
        const float bv = bl[ky * C3F_ROW + 2 * s2];
        const int ks = ky * 5 + s2;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(ks * 2 + 0) * 64 + lane], bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(ks * 2 + 1) * 64 + lane], bv, acc[1], 0, 0, 0);
      }
    // acc[h][r]: channel 32h + (r & 3) + 8 (r >> 2) + 4 lh of pixel li  ->  LDS [pixel][channel] (the patch is dead: LDS ops of a
    // wave complete in order)  ->  rows of 256 B, 4 pixels per store instruction
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * h + 8 * q + 4 * lh;
        f32x4 v = {acc[h][4 * q], acc[h][4 * q + 1], acc[h][4 * q + 2], acc[h][4 * q + 3]};
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
        *reinterpret_cast<f32x4*>(ps + li * C3F_OST + c) = v;
      }
    const int b = t / (H * tpr), rem = t - b * (H * tpr);
    const int oy = rem / tpr, x0 = (rem - oy * tpr) << 5;
    float* const yp = y + (((size_t)b * H + oy) * W + x0) * Cout;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + 64 * i, px = idx >> 4, c4 = (idx & 15) * 4;
      *reinterpret_cast<f32x4*>(yp + (size_t)px * Cout + c4) = *reinterpret_cast<const f32x4*>(ps + px * C3F_OST + c4);
    }
  }
}

// 3x3 / stride 1 / pad 1 conv from 64 channels to 3 (round 2): the DATA-GRADIENT of the discriminator's first layer (model.py:32) -
// the last step of the generator's adversarial backward, d loss / d sr.  With 3 output channels the general kernel fills 3 of 32
// MFMA columns (8 TFLOP/s, 62 us at 96 px).  Here kx is folded into N: P[x'][(tx, ci)] = sum_{ty, co} dY[y-1+ty][x'][co] *
// w[(ci, co, ty, tx)] for every input column x' = -1 .. W (9 of 16 columns of v_mfma_f32_16x16x4_f32, K = 3 x 64), then
// dx[x][ci] = sum_tx P[x-1+tx][(tx, ci)] through LDS.  One wave = one output row: ceil((W+2)/16) M tiles x 48 MFMAs, the A operand
// (dY) loaded straight from global memory as 16-B quads in MFMA layout (next tile's 12 quads in flight under the current tile's
// MFMAs), the B operand (weights, 48 registers) read once from the ordinary mode-1 packed buffer.
typedef float f32x4c __attribute__((ext_vector_type(4)));
// nsp: a row is cut into nsp segments of Ws = W / nsp output columns, one wave each (nmt = tiles of a SEGMENT's Ws + 2 input columns):
// at 96 px two 48-column segments are 8 M tiles per row instead of 7, but twice the waves (3 per SIMD instead of 1.5) cover the load
// latency the one-tile-ahead prefetch leaves exposed (inside the step: 29.7 -> 28.8 us per launch at B = 16 - beside the other
// branch's kernels; SST_TO3_NO_SPLIT=1 for the one-wave-per-row form).
__global__ __launch_bounds__(CONV_NT) void conv3_to3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            float* __restrict__ y, int B, int H, int W, int nmt, int nsp) {
  extern __shared__ __attribute__((aligned(16))) float pl_all[];        // [wave][nmt * 16][12]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + wave;
  if (seg >= B * H * nsp) return;                          // no workgroup barrier below
  const int row = seg / nsp, Ws = W / nsp, xb = (seg - row * nsp) * Ws;
  const int b = row / H, oy = row - b * H;
  const int lm = lane & 15, lg = lane >> 4;
  float* const pl = pl_all + (size_t)wave * nmt * 16 * 12;
  // B operand: column n = lm = tx*3 + ci (n < 9), K index of MFMA (ty, q, t): co = 16q + 4 lg + t.  packed mode-1 layout
  // (conv_common.h: packed_index with o = ci < 3, i = co < 64): ((tap*8 + co/8)*256 + ((co&7)/4*32 + ci)*4 + (co&3)
  float bw[3][4][4];
  {
    const int tx = lm / 3, ci = lm - tx * 3;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int co = 16 * q + 4 * lg + t, tap = ty * 3 + tx;
          bw[ty][q][t] = lm < 9 ? wp[(size_t)(tap * 8 + (co >> 3)) * 256 + (((co & 7) >> 2) * 32 + ci) * 4 + (co & 3)] : 0.f;
        }
  }
  auto load_tile = [&](int mt, f32x4c (&a)[3][4]) {
    const int xp = xb + mt * 16 + lm - 1;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int iy = oy - 1 + ty;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)xp < (unsigned)W;
      const float* src = x + (((size_t)b * H + (ok ? iy : 0)) * W + (ok ? xp : 0)) * 64 + 4 * lg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4c v = *reinterpret_cast<const f32x4c*>(src + 16 * q);
        a[ty][q] = ok ? v : f32x4c{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  f32x4c ac[3][4], an[3][4];
  load_tile(0, ac);
  for (int mt = 0; mt < nmt; ++mt) {
    if (mt + 1 < nmt) load_tile(mt + 1, an);
    f32x4c acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[ty][q][t], bw[ty][q][t], acc, 0, 0, 0);
    // acc[r]: row x' (local) = mt*16 + 4 lg + r, column n = lm
    if (lm < 9) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pl[(mt * 16 + 4 * lg + r) * 12 + lm] = acc[r];
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q) ac[ty][q] = an[ty][q];
  }
  // horizontal fold (LDS ops of a wave complete in order): dx[x][ci] = sum_tx P[x + tx][(tx, ci)]  (local x' index = x' + 1)
  float* const out = y + (((size_t)b * H + oy) * W + xb) * 3;
  for (int i = lane; i < 3 * Ws; i += 64) {
    const int xo = i / 3, ci = i - xo * 3;
    out[i] = (pl[xo * 12 + ci] + pl[(xo + 1) * 12 + 3 + ci]) + pl[(xo + 2) * 12 + 6 + ci];
  }
}

// w [Cout][Cin][3][3] (reference layout) -> packed.  mode 0: forward.  mode 1: data-gradient of a stride-1
// conv (outputs = Cin, inputs = Cout, taps rotated 180 degrees).  Stride-2 data-gradients use
// pack_s2_dgrad_kernel below (one compact tap list per output-pixel parity class).
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KK,
                                 int mode, int64_t total) {
  const int O = mode ? Cin : Cout, I = mode ? Cout : Cin;
  const int ncb = (I + 63) / 64;
  const int64_t base = packed_floats_base(O, I, KK);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx >= base) {   // band-kernel section (conv_common.h: band_index)
      const int64_t q = idx - base;
      const int l = (q >> 2) & 63, tap = (q >> 10) % 9;
      const int o = (int)((q >> 10) / 9) * 16 + (l & 15), i = ((q >> 8) & 3) * 16 + (l >> 4) * 4 + (q & 3);
      wp[idx] = mode == 0 ? w[((size_t)o * Cin + i) * KK + tap] : w[((size_t)i * Cin + o) * KK + (KK - 1 - tap)];
      continue;
    }
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
    if (of < (O + 31) / 32 && o < O && i < I) {
      if (mode == 0)
        v = w[((size_t)o * Cin + i) * KK + tap];
      else
        v = w[((size_t)i * Cin + o) * KK + (KK - 1 - tap)];
    }
    wp[idx] = v;
  }
}

// Multi-tensor variant: ONE launch packs every conv weight of a network (job table in device memory).
struct PackJob {
  const float* w;
  float* wp;
  int Cout, Cin, KK, mode;
  long long total;
  long long block_begin;    // first workgroup of this job (each workgroup packs 1024 floats)
};
// The job of a workgroup's block is found by binary search over the jobs' first blocks - in an LDS copy of that column (one coalesced
// load per workgroup): searched in global memory it is 6-7 DEPENDENT loads in front of every block's gathers.
constexpr int PM_MAXJOBS = 512;
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ long long s_begin[PM_MAXJOBS];
  for (int j = threadIdx.x; j < njobs; j += 256) s_begin[j] = jobs[j].block_begin;
  __syncthreads();
  int lo = 0, hi = njobs - 1;                       // last job with block_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s_begin[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob jb = jobs[lo];
  if (jb.mode >= 2) {                               // 9x9 convs with a 3-channel side: 2 / 3 = c3 mode 0 / 1, 4 = to3
    const long long base = ((long long)blockIdx.x - jb.block_begin) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long idx = base + u * 256 + threadIdx.x;
      if (idx >= jb.total) break;
      // 5 / 6: the two one-line launches that used to stand next to a network's pack at the head of its forward ride along as jobs -
      // 5 clears `total` 32-bit words at wp (the trunk's fp64 statistics accumulators), 6 adds Cout to `total` int64 words at wp (the
      // BatchNorm layers' num_batches_tracked, one flat tensor per network)
      if (jb.mode == 5) { jb.wp[idx] = 0.f; continue; }
      if (jb.mode == 6) { reinterpret_cast<long long*>(jb.wp)[idx] += jb.Cout; continue; }
      jb.wp[idx] = jb.mode == 4 ? pack_to3_value(jb.w, idx, jb.Cin) : pack_c3_value(jb.w, idx, jb.Cout, jb.Cin, jb.mode - 2);
    }
    return;
  }
  const int O = jb.mode ? jb.Cin : jb.Cout, I = jb.mode ? jb.Cout : jb.Cin;
  const int ncb = (I + 63) / 64;
  const unsigned wl = (unsigned)((long long)blockIdx.x - jb.block_begin);          // this workgroup's 1,024-float block of the job
  const unsigned bbase = (unsigned)packed_floats_base(O, I, jb.KK), total = (unsigned)jb.total;     // (< 2^31: checked by the caller)
  if (jb.KK == 9) {
    // 3x3 weights: the packed element (o, i, tap) comes from w[o][i][tap] (mode 1: w[i][o][8 - tap]) - gathered element by element
    // that is one 64-B sector per 4 bytes (the generator's 5 M packed floats pulled 320 MB through L2: 33-40 us at the head of the
    // iteration).  Here the workgroups of a (32 o x 32 i) block's NINE taps are folded into the first of them: it reads the block's
    // 32 rows of 32 x 9 contiguous floats once (coalesced) into LDS and writes all nine 1,024-float packed blocks; the other eight
    // exit.  Same for the band section (16 o x 64 i x 9 taps per workgroup).
    __shared__ float tile[64 * 145];                      // [32 rows][288 + 1] / band section: [16][576 + 1] or [64][144 + 1]
    const unsigned nhalf = (unsigned)((O + 31) / 32) * ncb * 9 * 2;                // half-slab blocks of the first section
    const bool band = wl >= nhalf + PACK_PAD / 1024;
    if (!band && wl >= nhalf) {                           // the pad
#pragma unroll
      for (int u = 0; u < 4; ++u) jb.wp[wl * 1024 + u * 256 + threadIdx.x] = 0.f;
      return;
    }
    if (!band) {
      const unsigned ngrp = nhalf / 9;
      if (wl >= ngrp) return;
      const int h = wl & 1, cbk = (wl >> 1) % (unsigned)ncb, of = (wl >> 1) / (unsigned)ncb;
      const int o0 = of * 32, i0 = cbk * 64 + h * 32;
      // rows: mode 0 -> o (columns i_local * 9 + tap), mode 1 -> i (columns o_local * 9 + (8 - tap))
      float v[36];                                        // 32 x 288 / 256: every load issued before the first LDS store
#pragma unroll
      for (int k = 0; k < 36; ++k) {
        const int e = k * 256 + threadIdx.x;
        const int r = e / 288, c = e - r * 288, q = c / 9;
        v[k] = 0.f;
        if (jb.mode == 0) {
          if (o0 + r < O && i0 + q < I) v[k] = jb.w[((size_t)(o0 + r) * jb.Cin + i0) * 9 + c];
        } else {
          if (i0 + r < I && o0 + q < O) v[k] = jb.w[((size_t)(i0 + r) * jb.Cin + o0) * 9 + c];
        }
      }
#pragma unroll
      for (int k = 0; k < 36; ++k) {
        const int e = k * 256 + threadIdx.x;
        const int r = e / 288, c = e - r * 288;
        tile[r * 289 + c] = v[k];
      }
      __syncthreads();
      for (int tap = 0; tap < 9; ++tap) {
        float* dst = jb.wp + ((size_t)((of * ncb + cbk) * 9 + tap) * 2 + h) * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int x = u * 256 + threadIdx.x;
          const int j = x & 3, l = (x >> 2) & 63, ks = x >> 8;
          const int ol = l & 31, il = ks * 8 + (l >> 5) * 4 + j;
          dst[x] = jb.mode == 0 ? tile[ol * 289 + il * 9 + tap] : tile[il * 289 + ol * 9 + (8 - tap)];
        }
      }
      return;
    }
    // band section [o / 16][tap][i / 16][64 lanes][4] (conv_common.h: band_index; I = 64)
    const unsigned wb = wl - nhalf - PACK_PAD / 1024;
    if (wb >= (unsigned)(O / 16)) return;
    const int o0 = wb * 16;
    {
      float v[36];
      // mode 0: 16 rows (o) of 64 x 9 contiguous floats; mode 1: 64 rows (i) of 16 x 9
      const int RL = jb.mode == 0 ? 576 : 144;
#pragma unroll
      for (int k = 0; k < 36; ++k) {
        const int e = k * 256 + threadIdx.x;
        const int r = e / RL, c = e - r * RL;
        v[k] = jb.mode == 0 ? jb.w[((size_t)(o0 + r) * jb.Cin) * 9 + c] : jb.w[((size_t)r * jb.Cin + o0) * 9 + c];
      }
#pragma unroll
      for (int k = 0; k < 36; ++k) {
        const int e = k * 256 + threadIdx.x;
        const int r = e / RL, c = e - r * RL;
        tile[r * (RL + 1) + c] = v[k];
      }
    }
    __syncthreads();
    for (int tap = 0; tap < 9; ++tap) {
      float* dst = jb.wp + bbase + ((size_t)wb * 9 + tap) * 1024;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = u * 256 + threadIdx.x;
        const int l = (x >> 2) & 63;
        const int ol = l & 15, i = (x >> 8) * 16 + (l >> 4) * 4 + (x & 3);
        dst[x] = jb.mode == 0 ? tile[ol * 577 + i * 9 + tap] : tile[i * 145 + ol * 9 + (8 - tap)];
      }
    }
    return;
  }
  // other kernel sizes (9x9 through the general kernels): element-wise gather, 32-bit index arithmetic, the part of the index above
  // bit 10 decoded once per workgroup
  const unsigned base = wl * 1024;
  const unsigned top = base >> 11;                        // uniform (no band section for these sizes)
  int tap = top % (unsigned)jb.KK;
  const unsigned rr = top / (unsigned)jb.KK;
  int a2 = (int)(rr % (unsigned)ncb) * 64, a1 = (int)(rr / (unsigned)ncb) * 32;
  tap = __builtin_amdgcn_readfirstlane(tap); a1 = __builtin_amdgcn_readfirstlane(a1); a2 = __builtin_amdgcn_readfirstlane(a2);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned idx = base + u * 256 + threadIdx.x;
    if (idx >= total) break;
    const int j = idx & 3, l = (idx >> 2) & 63, ks = (idx >> 8) & 7;
    const int o = a1 + (l & 31), i = a2 + ks * 8 + (l >> 5) * 4 + j;
    float v = 0.f;
    if (a1 < ((O + 31) / 32) * 32 && o < O && i < I) {
      if (jb.mode == 0)
        v = jb.w[((size_t)o * jb.Cin + i) * jb.KK + tap];
      else
        v = jb.w[((size_t)i * jb.Cin + o) * jb.KK + (jb.KK - 1 - tap)];
    }
    jb.wp[idx] = v;
  }
}

// Data-gradient of a 3x3 / stride-2 / pad-1 conv, parity class (py,px) of the input-gradient pixel (iy,ix) =
// (2a+py, 2b+px):  dX[iy,ix,ci] = sum over taps (jy,jx) of the class, co:  dY[a+jy, b+jx, co] * W[co][ci][ky][kx]
// with ky = py ? (jy ? 0 : 2) : 1 (same for kx).  Class c = 2*py+px has (1+py)*(1+px) taps; the 4 classes are
// packed back to back (offsets: s2_class_offset).
__host__ __device__ inline int64_t s2_class_offset(int cls, int Cin, int Cout) {
  int64_t off = 0;
  for (int c = 0; c < cls; ++c) off += packed_floats(Cin, Cout, (1 + (c >> 1)) * (1 + (c & 1)));
  return off;
}
__global__ void pack_s2_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
  if (blockIdx.y == 4) {
    // 5th section, for the pipelined kernel (conv_pipe.hip, stride-2 data-gradient mode): the 9 (class, tap) pairs as the 9
    // "taps" of ONE forward-layout block list [Cin/32][Cout/64][9][8][256] - pair c = class {0,1,1,2,2,3,3,3,3}[c], tap
    // {0,0,1,0,1,0,1,2,3}[c] of that class
    const int O = Cin, I = Cout, ncb = (I + 63) / 64;
    float* dst = wp + s2_class_offset(4, Cin, Cout);
    const int64_t total = packed_floats_base(O, I, 9);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
      const int j = idx & 3, l = (idx >> 2) & 63, ks = (idx >> 8) & 7;
      int64_t rest = idx >> 11;
      const int combo = rest % 9;
      rest /= 9;
      const int cbk = rest % ncb;
      const int of = rest / ncb;
      const int o = of * 32 + (l & 31), i = cbk * 64 + ks * 8 + (l >> 5) * 4 + j;
      float v = 0.f;
      if (of < (O + 31) / 32 && o < O && i < I) {
        const int cls = combo == 0 ? 0 : (combo < 3 ? 1 : (combo < 5 ? 2 : 3));
        const int tap = combo == 0 ? 0 : (combo < 3 ? combo - 1 : (combo < 5 ? combo - 3 : combo - 5));
        const int py = cls >> 1, px = cls & 1, ntx = 1 + px;
        const int jy = tap / ntx, jx = tap - jy * ntx;
        const int ky = py ? (jy ? 0 : 2) : 1, kx = px ? (jx ? 0 : 2) : 1;
        v = w[(((size_t)i * Cin + o) * 3 + ky) * 3 + kx];
      }
      dst[idx] = v;
    }
    return;
  }
  const int cls = blockIdx.y, py = cls >> 1, px = cls & 1;
  const int nty = 1 + py, ntx = 1 + px, KKc = nty * ntx;
  const int O = Cin, I = Cout, ncb = (I + 63) / 64;
  float* dst = wp + s2_class_offset(cls, Cin, Cout);
  const int64_t total = packed_floats(O, I, KKc);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx & 3, l = (idx >> 2) & 63, ks = (idx >> 8) & 7;
    int64_t rest = idx >> 11;
    const int tap = rest % KKc;
    rest /= KKc;
    const int cbk = rest % ncb;
    const int of = rest / ncb;
    const int o = of * 32 + (l & 31), i = cbk * 64 + ks * 8 + (l >> 5) * 4 + j;
    float v = 0.f;
    if (of < (O + 31) / 32 && o < O && i < I) {
      const int jy = tap / ntx, jx = tap - jy * ntx;
      const int ky = py ? (jy ? 0 : 2) : 1, kx = px ? (jx ? 0 : 2) : 1;
      v = w[(((size_t)i * Cin + o) * 3 + ky) * 3 + kx];
    }
    dst[idx] = v;
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

// jobs: device array of {w, wp, Cout, Cin, KK, mode, total, block_begin} (8 x 8 bytes each, see srganst/ops.py);
// mode 0 / 1: forward / data-gradient layout of a k x k conv; 2 / 3: sst_conv9_c3_pack mode 0 / 1; 4: sst_conv9_to3_pack;
// 5: clear `total` 32-bit words at wp (w unused); 6: add Cout to `total` int64 words at wp (w unused)
SST_API int sst_conv_pack_multi(const void* jobs, int njobs, int total_blocks, void* stream) {
  SST_REQUIRE(jobs && njobs > 0 && njobs <= PM_MAXJOBS && total_blocks > 0, "sst_conv_pack_multi: bad argument (at most %d jobs)", PM_MAXJOBS);
  static_assert(sizeof(PackJob) == 48, "PackJob layout");
  pack_multi_kernel<<<total_blocks, 256, 0, sst_stream(stream)>>>(reinterpret_cast<const PackJob*>(jobs), njobs);
  SST_LAUNCH_CHECK("pack_multi_kernel");
  return SST_OK;
}

SST_API int sst_conv_mtiles(int B, int Ho, int Wo) { return B * ((Ho + THO - 1) / THO) * ((Wo + TWO - 1) / TWO); }

// Number of statistics tiles (first dimension of stats / stats_cnt / epi_partial) the NHWC-store conv of this shape
// writes: one per band when the band kernel (conv_band.hip) takes the shape, else sst_conv_mtiles of the output.
SST_API int sst_conv_stat_tiles(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  const int R = sst_conv_band_rows(B, H, W, Cin, Cout, ksize, stride);
  if (R) return B * (H / R);
  const int p = ksize / 2;
  return sst_conv_mtiles(B, (H + 2 * p - ksize) / stride + 1, (W + 2 * p - ksize) / stride + 1);
}

// y = conv(act(x*in_scale+in_shift), w) (+bias) (+residual), stored per out_mode; optional BN partial stats.
static int conv_fwd_impl(const float* x, const float* wp, float* y, float* y_pre, const float* bias, const float* in_scale,
                         const float* in_shift, const float* in_slope, float in_slope_const, int in_act,
                         const float* residual, float* stats, float* stats_cnt, int out_mode, int B, int H, int W, int Cin,
                         int Cout, int ksize, int stride, const float* epi_y, const float* epi_scale, const float* epi_shift,
                         const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, void* stream,
                         const float* in2 = nullptr, const float* in_cA = nullptr, const float* in_cB = nullptr,
                         const float* in_cC = nullptr, float* side_out = nullptr, const BandAcc* band_acc = nullptr) {
  SST_REQUIRE(x && wp && y, "sst_conv_fwd: null pointer");
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) && (ksize == 3 || ksize == 9),
              "sst_conv_fwd: bad shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d stride=%d", B, H, W, Cin, Cout, ksize, stride);
  SST_REQUIRE(!(ksize == 9 && stride == 2), "sst_conv_fwd: 9x9 stride 2 not built");
  SST_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sst_conv_fwd: in_scale/in_shift must come together");
  SST_REQUIRE(!stats || stats_cnt, "sst_conv_fwd: stats needs stats_cnt");
  const int dbg_bits = out_mode >> 8;   // undocumented ablation bits (tools/), 0 in production
  out_mode &= 0xff;
  SST_REQUIRE(out_mode >= 0 && out_mode <= 3, "sst_conv_fwd: bad out_mode");
  SST_REQUIRE(out_mode != OUT_SHUFFLE || (Cout & 3) == 0, "sst_conv_fwd: shuffle store needs Cout %% 4 == 0");
  SST_REQUIRE(out_mode == OUT_NHWC || (!residual && !stats && !epi_partial), "sst_conv_fwd: residual/stats only with NHWC store");
  SST_REQUIRE(!epi_partial || (epi_y && !stats && ((epi_scale == nullptr) == (epi_shift == nullptr))),
              "sst_conv_fwd: backward partials need epi_y and exclude forward stats");
  Conv3Args a;
  a.x = x; a.wp = wp; a.y = y; a.y_pre = y_pre; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift;
  a.in_slope = in_slope; a.in_slope_const = in_slope_const; a.in_act = in_act; a.residual = residual; a.stats = stats;
  a.stats_cnt = stats_cnt; a.out_mode = out_mode; a.dbg = dbg_bits;
  a.epi_y = epi_y; a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_slope = epi_slope;
  a.epi_slope_const = epi_slope_const; a.epi_act = epi_act; a.epi_partial = epi_partial;
  a.in2 = in2; a.in_cA = in_cA; a.in_cB = in_cB; a.in_cC = in_cC; a.side_out = side_out;
  SST_REQUIRE(!in2 || ((Cin & 3) == 0 && stride == 1), "sst_conv_fwd: fused BN-backward input needs Cin %% 4 == 0, stride 1");
  SST_REQUIRE(!in_cA || (in2 && in_cB && in_cC), "sst_conv_fwd: cA/cB/cC need in2");
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ksy = a.ksx = ksize; a.pad_y = a.pad_x = ksize / 2; a.sub_y = a.sub_x = 0; a.Hy = a.Wy = 0;
  a.Ho = (H + 2 * (ksize / 2) - ksize) / stride + 1;
  a.Wo = (W + 2 * (ksize / 2) - ksize) / stride + 1;
  SST_REQUIRE(out_mode != OUT_UNSHUFFLE || ((a.Ho & 1) == 0 && (a.Wo & 1) == 0), "sst_conv_fwd: unshuffle needs even Ho,Wo");
  const int64_t mt = sst_conv_mtiles(B, a.Ho, a.Wo);
  SST_REQUIRE(mt < (1ll << 31), "sst_conv_fwd: too many tiles");
  dim3 grid((unsigned)mt, (Cout + 31) / 32);
  hipStream_t st = sst_stream(stream);
  const size_t extra_lds = (size_t)(dbg_bits >> 4) * 1024;   // dev knob: pad LDS to cap workgroups per CU
  a.dbg = dbg_bits & 15;
  if (ksize == 3 && stride == 1 && Cin == 64 && Cout == 3 && out_mode == OUT_NHWC && !bias && !in_scale && in_act == ACT_NONE && !residual &&
      !stats && !y_pre && !epi_partial && !in2 && dbg_bits == 0 && !sst_env("SST_NO_TO3")) {
    const int nsp = (W % 32 == 0 && W >= 64 && !sst_env("SST_TO3_NO_SPLIT")) ? 2 : 1;
    const int nmt = (W / nsp + 2 + 15) / 16;
    const size_t lds = (size_t)4 * nmt * 16 * 12 * sizeof(float);
    if (lds <= 60 * 1024) {
      conv3_to3_kernel<<<(unsigned)((B * H * nsp + 3) / 4), CONV_NT, lds, st>>>(x, wp, y, B, H, W, nmt, nsp);
      SST_LAUNCH_CHECK("conv3_to3_kernel");
      return SST_OK;
    }
  }
  if (ksize == 3 && stride == 1 && Cin == 3 && (Cout & 3) == 0 && CONV_NT % (Cout >> 2) == 0 && out_mode == OUT_NHWC && !in_scale &&
      in_act == ACT_NONE && !residual && !stats && !y_pre && !epi_partial && !in2 && dbg_bits == 0 && !sst_env("SST_NO_C3IN")) {
    if ((W & 31) == 0 && Cout == 64 && !sst_env("SST_NO_C3IN_MFMA")) {
      const int ntiles = B * H * (W >> 5);
      const int wgs = (ntiles + 3) / 4;
      conv3_c3in_mfma_kernel<<<(unsigned)(wgs < 768 ? wgs : 768), CONV_NT, 0, st>>>(x, wp, bias, y, B, H, W, ntiles);
      SST_LAUNCH_CHECK("conv3_c3in_mfma_kernel");
      return SST_OK;
    }
    const size_t lds = (size_t)3 * ((W + 2) * 3 + 1) * sizeof(float);
    if (lds <= 48 * 1024) {
      conv3_c3in_kernel<<<(unsigned)(B * H), CONV_NT, lds, st>>>(x, wp, bias, y, B, H, W, Cout);
      SST_LAUNCH_CHECK("conv3_c3in_kernel");
      return SST_OK;
    }
  }
  if (out_mode == OUT_NHWC && !(dbg_bits & 8)) {
    const int R = sst_conv_band_rows(B, H, W, Cin, Cout, ksize, stride);
    if (R) return sst_launch_conv_band(a, R, st, band_acc);
  }
  if (dbg_bits == 0 && use_big_tiles(a, ksize)) return sst_launch_conv_fwd2(a, stride, st);
  if (ksize == 3 && stride == 1)
    conv_fwd_kernel<3, 1><<<grid, CONV_NT, extra_lds, st>>>(a);
  else if (ksize == 3)
    conv_fwd_kernel<3, 2><<<grid, CONV_NT, 0, st>>>(a);
  else
    conv_fwd_kernel<9, 1><<<grid, CONV_NT, 0, st>>>(a);
  SST_LAUNCH_CHECK("conv_fwd_kernel");
  return SST_OK;
}

// Name of the kernel sst_conv_fwd / sst_conv_dgrad_* dispatch to for this shape (as rocprofv3 prints it, without the
// argument list) - bench.py labels its roofline rows with it so that they can be matched against profiles/.
SST_API const char* sst_conv_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int out_mode, int fused_in) {
  const int p = ksize / 2;
  Conv3Args a{};
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.out_mode = out_mode;
  a.Ho = (H + 2 * p - ksize) / stride + 1;
  a.Wo = (W + 2 * p - ksize) / stride + 1;
  a.in2 = fused_in ? reinterpret_cast<const float*>(1) : nullptr;
  if (ksize == 3 && stride == 1 && Cin == 64 && Cout == 3 && out_mode == OUT_NHWC && !fused_in && !sst_env("SST_NO_TO3") &&
      (size_t)4 * ((W + 2 + 15) / 16) * 16 * 12 * sizeof(float) <= 60 * 1024)
    return "conv3_to3_kernel";       // (when called plain, as the data-gradient of a 3-channel-input layer is)
  if (ksize == 3 && stride == 1 && Cin == 3 && (Cout & 3) == 0 && CONV_NT % (Cout >> 2) == 0 && out_mode == OUT_NHWC && !fused_in)
    return ((W & 31) == 0 && Cout == 64 && !sst_env("SST_NO_C3IN_MFMA")) ? "conv3_c3in_mfma_kernel" : "conv3_c3in_kernel";   // (when called without input affine / activation / residual / statistics)
  if (out_mode == OUT_NHWC) {
    const int R = sst_conv_band_rows(B, H, W, Cin, Cout, ksize, stride);
    if (R) return R * W / 16 == 9 ? "conv_band_kernel<9>" : "conv_band_kernel<3>";
  }
  if (use_big_tiles(a, ksize)) return stride == 1 ? "conv_fwd2_kernel<1>" : "conv_fwd2_kernel<2>";
  if (ksize == 3) return stride == 1 ? "conv_fwd_kernel<3, 1>" : "conv_fwd_kernel<3, 2>";
  return "conv_fwd_kernel<9, 1>";
}

SST_API int sst_conv_fwd(const float* x, const float* wp, float* y, float* y_pre, const float* bias, const float* in_scale,
                         const float* in_shift, const float* in_slope, float in_slope_const, int in_act,
                         const float* residual, float* stats, float* stats_cnt, int out_mode, int B, int H, int W, int Cin,
                         int Cout, int ksize, int stride, void* stream) {
  return conv_fwd_impl(x, wp, y, y_pre, bias, in_scale, in_shift, in_slope, in_slope_const, in_act, residual, stats, stats_cnt,
                       out_mode, B, H, W, Cin, Cout, ksize, stride, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, stream);
}

// Data-gradient conv (NHWC store, optional residual) that also emits the BatchNorm/activation backward partials of its
// result g against the saved conv output epi_y:  epi_partial [sst_conv_mtiles][3][Cout]  (same layout as sst_bwd_reduce).
SST_API int sst_conv_dgrad_bwdstats(const float* x, const float* wp, float* y, const float* residual, const float* epi_y,
                                    const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                    float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                    int Cout, int ksize, void* stream) {
  SST_REQUIRE(epi_y && epi_partial, "sst_conv_dgrad_bwdstats: epi_y / epi_partial");
  return conv_fwd_impl(x, wp, y, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, ACT_NONE, residual, nullptr, nullptr, OUT_NHWC,
                       B, H, W, Cin, Cout, ksize, 1, epi_y, epi_scale, epi_shift, epi_slope, epi_slope_const, epi_act, epi_partial,
                       stream);
}

// The whole BatchNorm-backward stage of a stride-1 conv in one launch:
//   dy   = cA*gz + cB*y2 + cC, gz = in_act ? (y2*in_scale+in_shift > 0 ? g : g*slope) : g     (BN+activation backward apply,
//                                                                           computed while the input tile is staged)
//   dy_out[b,y,x,:] = dy                                                     (side output for the weight-gradient kernel)
//   out  = conv(dy, wp) (+ residual)                                         (data-gradient, mode-1 weights)
//   epi_partial = BN/activation backward partials of `out` against epi_y     (optional, as sst_conv_dgrad_bwdstats)
SST_API int sst_conv_dgrad_fused(const float* g, const float* y2, const float* cA, const float* cB, const float* cC,
                                 const float* in_scale, const float* in_shift, const float* in_slope, float in_slope_const,
                                 int in_act, float* dy_out, const float* wp, float* out, const float* residual,
                                 const float* epi_y, const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                 float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin, int Cout,
                                 int ksize, void* stream) {
  SST_REQUIRE(g && y2 && dy_out, "sst_conv_dgrad_fused: g / y2 / dy_out");
  SST_REQUIRE((epi_partial == nullptr) == (epi_y == nullptr), "sst_conv_dgrad_fused: epi_y and epi_partial come together");
  return conv_fwd_impl(g, wp, out, nullptr, nullptr, in_scale, in_shift, in_slope, in_slope_const, in_act, residual, nullptr,
                       nullptr, OUT_NHWC, B, H, W, Cin, Cout, ksize, 1, epi_y, epi_scale, epi_shift, epi_slope, epi_slope_const,
                       epi_act, epi_partial, stream, y2, cA, cB, cC, dy_out);
}

// Forward conv whose input is a residual sum that has not been materialised yet (reference model.py:180-186:
// out = x + rcb(x) feeding the next block's first conv):
//   h = x + y2*bn_scale + bn_shift   (computed while the input tile is staged; also written to h_out for later consumers)
//   y = conv(h, wp) (+ bias), optional BatchNorm partial statistics of y          - replaces one bn_residual launch.
SST_API int sst_conv_fwd_resin(const float* x, const float* y2, const float* ones, const float* bn_scale, const float* bn_shift,
                               float* h_out, const float* wp, float* y, const float* bias, float* stats, float* stats_cnt, int B,
                               int H, int W, int Cin, int Cout, int ksize, void* stream) {
  SST_REQUIRE(x && y2 && ones && bn_scale && bn_shift && h_out, "sst_conv_fwd_resin: null pointer");
  return conv_fwd_impl(x, wp, y, nullptr, bias, nullptr, nullptr, nullptr, 0.f, ACT_NONE, nullptr, stats, stats_cnt, OUT_NHWC, B, H, W,
                       Cin, Cout, ksize, 1, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, stream, y2, ones, bn_scale, bn_shift,
                       h_out);
}

// ---- accumulator mode of the BatchNorm statistics (band kernel only, see BandAcc in conv_epilogue.h)
SST_API int sst_conv_acc_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  return Cin == 64 && sst_conv_band_rows(B, H, W, Cin, Cout, ksize, stride) != 0;
}

// Forward conv in accumulator mode.  Input forms:
//   in2 == null : staged = act(x*scale + shift)      (scale, shift) from in_acc when given, else no affine
//   in2 != null : staged = x + in2*scale + shift     (residual sum of the previous block), also written to side_out
// in_acc / st_acc: fp64 accumulators [nrep][64][2] / [nrep][Cout][2] (st_acc must be zero before the launch); o_* and run_*
// may be null.  Fails unless sst_conv_acc_supported().
SST_API int sst_conv_fwd_acc(const float* x, const float* in2, float* side_out, const float* ones, const float* wp, float* y,
                             const float* bias, const float* in_slope, float in_slope_const, int in_act, const double* in_acc,
                             const float* in_gamma, const float* in_beta, float in_n, float eps, float momentum, float* o_mean,
                             float* o_rstd, float* o_scale, float* o_shift, float* run_mean, float* run_var, double* st_acc,
                             int nrep, int B, int H, int W, int Cin, int Cout, int ksize, void* stream) {
  SST_REQUIRE(sst_conv_acc_supported(B, H, W, Cin, Cout, ksize, 1), "sst_conv_fwd_acc: shape not covered by the band kernel");
  SST_REQUIRE(nrep > 0 && nrep <= 16 && (in_acc || st_acc), "sst_conv_fwd_acc: no accumulator given / nrep > 16");
  SST_REQUIRE(!in_acc || (in_gamma && in_beta && in_n > 0.f), "sst_conv_fwd_acc: in_acc needs gamma / beta / n");
  SST_REQUIRE(!in2 || (side_out && ones), "sst_conv_fwd_acc: residual form needs side_out and the ones vector");
  BandAcc ba{};
  ba.st_acc = st_acc; ba.in_acc = in_acc; ba.nrep = nrep; ba.in_target = in2 ? 1 : 0;
  ba.in_gamma = in_gamma; ba.in_beta = in_beta; ba.in_n = in_n; ba.in_eps = eps; ba.momentum = momentum;
  ba.o_mean = o_mean; ba.o_rstd = o_rstd; ba.o_scale = o_scale; ba.o_shift = o_shift; ba.run_mean = run_mean; ba.run_var = run_var;
  // the residual form rides on the fused-input path (kA = ones, kB / kC come from the accumulators inside the kernel)
  return conv_fwd_impl(x, wp, y, nullptr, bias, nullptr, nullptr, in_slope, in_slope_const, in_act, nullptr, nullptr, nullptr, OUT_NHWC,
                       B, H, W, Cin, Cout, ksize, 1, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, stream, in2, in2 ? ones : nullptr,
                       in2 ? ones : nullptr, in2 ? ones : nullptr, side_out, &ba);
}

// One BatchNorm-backward stage in accumulator mode (see sst_conv_dgrad_fused for the arithmetic): the coefficients of
//   dy = cA*gz + cB*y2 + cC   come from bw_in_acc [nrep][64][4] (sums added by the PREVIOUS stage's epilogue) + mean / rstd /
// gamma, and this stage's epilogue adds the next stage's sums into bw_st_acc [nrep][Cout][4] (zero before the launch).
// y2 == null: plain data-gradient of g (no BatchNorm-backward apply on the input; bw_in_acc must be null).
// dgamma / dbeta [64] (and dslope [1] when non-null) of the applied BatchNorm are written once.
SST_API int sst_conv_dgrad_fused_acc(const float* g, const float* y2, const float* in_scale, const float* in_shift, const float* in_slope,
                                     float in_slope_const, int in_act, float* dy_out, const float* wp, float* out,
                                     const float* residual, const float* epi_y, const float* epi_scale, const float* epi_shift,
                                     const float* epi_slope, float epi_slope_const, int epi_act, const double* bw_in_acc,
                                     const float* mean, const float* rstd, const float* gamma, float n, float* dgamma, float* dbeta,
                                     float* dslope, double* bw_st_acc, int nrep, int B, int H, int W, int Cin, int Cout, int ksize,
                                     void* stream) {
  SST_REQUIRE(sst_conv_acc_supported(B, H, W, Cin, Cout, ksize, 1), "sst_conv_dgrad_fused_acc: shape not covered by the band kernel");
  SST_REQUIRE(nrep > 0 && nrep <= 16 && (bw_in_acc || bw_st_acc), "sst_conv_dgrad_fused_acc: no accumulator given / nrep > 16");
  SST_REQUIRE(!bw_in_acc || (y2 && dy_out && mean && rstd && gamma && dgamma && dbeta && n > 0.f),
              "sst_conv_dgrad_fused_acc: bw_in_acc needs y2 / dy_out / mean / rstd / gamma / dgamma / dbeta / n");
  SST_REQUIRE(!bw_st_acc || epi_y, "sst_conv_dgrad_fused_acc: bw_st_acc needs epi_y");
  SST_REQUIRE(!y2 || dy_out, "sst_conv_dgrad_fused_acc: y2 needs dy_out");
  BandAcc ba{};
  ba.nrep = nrep;
  ba.bw_st_acc = bw_st_acc; ba.bw_in_acc = bw_in_acc; ba.bw_mean = mean; ba.bw_rstd = rstd; ba.bw_gamma = gamma; ba.bw_n = n;
  ba.o_dgamma = dgamma; ba.o_dbeta = dbeta; ba.o_dslope = dslope;
  return conv_fwd_impl(g, wp, out, nullptr, nullptr, in_scale, in_shift, in_slope, in_slope_const, in_act, residual, nullptr, nullptr,
                       OUT_NHWC, B, H, W, Cin, Cout, ksize, 1, epi_y, epi_scale, epi_shift, epi_slope, epi_slope_const, epi_act, nullptr,
                       stream, y2, nullptr, nullptr, nullptr, y2 ? dy_out : nullptr, &ba);
}

// ---- data-gradient of a 3x3 stride-2 pad-1 convolution (Discriminator.features, model.py:35,42,49,56)
// 4 per-class sections + the unified 9-pair section of the pipelined kernel
SST_API int64_t sst_conv_s2_dgrad_packed_floats(int Cout, int Cin) { return s2_class_offset(4, Cin, Cout) + packed_floats_base(Cin, Cout, 9); }
int64_t sst_conv_s2_dgrad_pipe_section(int Cout, int Cin) { return s2_class_offset(4, Cin, Cout); }     // for conv_pipe.hip

SST_API int sst_conv_s2_dgrad_pack(const float* w, float* wp, int Cout, int Cin, void* stream) {
  SST_REQUIRE(w && wp && Cout > 0 && Cin > 0, "sst_conv_s2_dgrad_pack: bad argument");
  pack_s2_dgrad_kernel<<<dim3(256, 5), 256, 0, sst_stream(stream)>>>(w, wp, Cout, Cin);
  SST_LAUNCH_CHECK("pack_s2_dgrad_kernel");
  return SST_OK;
}

// dx [B,H,W,Cin] = conv_transpose(dy [B,Ho,Wo,Cout]) for y = conv3x3(x, stride 2, pad 1): the 4 parity classes of dx in one
// launch (four when a class is empty or large enough for the 64x64-tile kernel).
struct S2Fused {             // optional fused BatchNorm-backward stage around the data-gradient (see sst_conv_dgrad_fused)
  const float* in2; const float* cA; const float* cB; const float* cC; const float* in_scale; const float* in_shift;
  const float* in_slope; float in_slope_const; int in_act; float* dy_out;
  const float* epi_y; const float* epi_scale; const float* epi_shift; const float* epi_slope; float epi_slope_const; int epi_act;
  float* epi_partial;
};
SST_API int sst_conv_s2_dgrad_tiles(int B, int H, int W) {
  int t = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int nh = (H - (cls >> 1) + 1) / 2, nw = (W - (cls & 1) + 1) / 2;
    if (nh > 0 && nw > 0) t += sst_conv_mtiles(B, nh, nw);
  }
  return t;
}
static int conv_s2_dgrad_impl(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cin, int Cout, void* stream,
                              const S2Fused* f) {
  SST_REQUIRE(dy && wp && dx && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "sst_conv_s2_dgrad: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  {
    // merged launch (all classes non-empty and none of them large enough for the 64x64-tile kernel)
    Conv3Args a;
    a.x = dy; a.wp = wp; a.y = dx; a.y_pre = nullptr; a.bias = nullptr;
    a.in_scale = a.in_shift = a.in_slope = nullptr; a.in_slope_const = 0.f; a.in_act = ACT_NONE;
    a.residual = nullptr; a.stats = nullptr; a.stats_cnt = nullptr; a.out_mode = OUT_STRIDE2; a.dbg = 0;
    a.epi_y = a.epi_scale = a.epi_shift = a.epi_slope = nullptr; a.epi_slope_const = 0.f; a.epi_act = 0; a.epi_partial = nullptr;
    a.in2 = a.in_cA = a.in_cB = a.in_cC = nullptr; a.side_out = nullptr;
    if (f) {
      a.in2 = f->in2; a.in_cA = f->cA; a.in_cB = f->cB; a.in_cC = f->cC; a.side_out = f->dy_out;
      a.in_scale = f->in_scale; a.in_shift = f->in_shift; a.in_slope = f->in_slope; a.in_slope_const = f->in_slope_const;
      a.in_act = f->in_act;
      a.epi_y = f->epi_y; a.epi_scale = f->epi_scale; a.epi_shift = f->epi_shift; a.epi_slope = f->epi_slope;
      a.epi_slope_const = f->epi_slope_const; a.epi_act = f->epi_act; a.epi_partial = f->epi_partial;
    }
    a.B = B; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin;
    a.ksy = a.ksx = 1; a.pad_y = a.pad_x = 0; a.sub_y = a.sub_x = 0; a.Ho = a.Wo = 0; a.Hy = H; a.Wy = W;
    S2Classes c;
    bool ok = !sst_env("SST_S2_SPLIT") || f;
    int max_tiles = 0, tbase = 0;
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1;
      c.nh[cls] = (H - py + 1) / 2;
      c.nw[cls] = (W - px + 1) / 2;
      c.wp_off[cls] = s2_class_offset(cls, Cin, Cout);
      c.tiles[cls] = (c.nh[cls] > 0 && c.nw[cls] > 0) ? sst_conv_mtiles(B, c.nh[cls], c.nw[cls]) : 0;
      c.tile_base[cls] = tbase;
      tbase += c.tiles[cls];
      if (c.tiles[cls] == 0 && !f) ok = false;
      Conv3Args t = a;
      t.Ho = c.nh[cls]; t.Wo = c.nw[cls];
      if (!f && c.tiles[cls] && use_big_tiles(t, 3)) ok = false;
      max_tiles = c.tiles[cls] > max_tiles ? c.tiles[cls] : max_tiles;
    }
    static const bool s2d4 = !(sst_env("SST_S2DGRAD4") && atoi(sst_env("SST_S2DGRAD4")) == 0);
    if (ok && s2d4 && !(H & 1) && !(W & 1) && !(Cout & 3) && !(Cin & 3)) {
      a.Ho = c.nh[0]; a.Wo = c.nw[0];
      dim3 grid((unsigned)c.tiles[0], (Cin + 31) / 32);
      if (f)
        conv_s2dgrad4_kernel<true><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, c);
      else
        conv_s2dgrad4_kernel<false><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, c);
      SST_LAUNCH_CHECK("conv_s2dgrad4_kernel");
      return SST_OK;
    }
    if (ok) {
      dim3 grid((unsigned)max_tiles, (Cin + 31) / 32, 4);
      conv_s2dgrad_kernel<<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, c);
      SST_LAUNCH_CHECK("conv_s2dgrad_kernel");
      return SST_OK;
    }
    SST_REQUIRE(!f, "sst_conv_s2_dgrad_fused: merged launch not possible");
  }
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    const int nh = (H - py + 1) / 2, nw = (W - px + 1) / 2;   // pixels of this parity class
    if (nh <= 0 || nw <= 0) continue;
    Conv3Args a;
    a.x = dy; a.wp = wp + s2_class_offset(cls, Cin, Cout); a.y = dx; a.y_pre = nullptr; a.bias = nullptr;
    a.in_scale = a.in_shift = a.in_slope = nullptr; a.in_slope_const = 0.f; a.in_act = ACT_NONE;
    a.residual = nullptr; a.stats = nullptr; a.stats_cnt = nullptr; a.out_mode = OUT_STRIDE2; a.dbg = 0;
    a.epi_y = a.epi_scale = a.epi_shift = a.epi_slope = nullptr; a.epi_slope_const = 0.f; a.epi_act = 0; a.epi_partial = nullptr;
    a.in2 = a.in_cA = a.in_cB = a.in_cC = nullptr; a.side_out = nullptr;
    a.B = B; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = Cin;     // roles swap: the "input" of this conv is dy
    a.ksy = 1 + py; a.ksx = 1 + px; a.pad_y = a.pad_x = 0; a.sub_y = py; a.sub_x = px;
    a.Ho = nh; a.Wo = nw; a.Hy = H; a.Wy = W;
    if (use_big_tiles(a, 3)) {
      const int rc = sst_launch_conv_fwd2(a, 1, sst_stream(stream));
      if (rc != SST_OK) return rc;
      continue;
    }
    dim3 grid((unsigned)sst_conv_mtiles(B, nh, nw), (Cin + 31) / 32);
    conv_fwd_kernel<3, 1><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a);
    SST_LAUNCH_CHECK("conv_fwd_kernel<3,1> (s2 dgrad)");
  }
  return SST_OK;
}

// Name of the kernel sst_conv_s2_dgrad (fused = 0) / sst_conv_s2_dgrad_fused (fused = 1) launch for this shape (rocprofv3
// spelling): the merged-classes kernel for even H, W with channel counts that are multiples of 4, else the per-class launch.
SST_API const char* sst_conv_s2_dgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int fused) {
  (void)B;
  const bool off = sst_env("SST_S2DGRAD4") && atoi(sst_env("SST_S2DGRAD4")) == 0;
  if (!off && !(H & 1) && !(W & 1) && !(Cout & 3) && !(Cin & 3)) return fused ? "conv_s2dgrad4_kernel<true>" : "conv_s2dgrad4_kernel<false>";
  return "conv_s2dgrad_kernel";
}

SST_API int sst_conv_s2_dgrad(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cin, int Cout,
                              void* stream) {
  return conv_s2_dgrad_impl(dy, wp, dx, B, H, W, Cin, Cout, stream, nullptr);
}

// One BatchNorm-backward stage around a STRIDE-2 data-gradient (same contract as sst_conv_dgrad_fused; g / y2 / dy_out are
// [B,Ho,Wo,Cout] of the conv's output side, dx / epi_y [B,H,W,Cin]); epi_partial: [sst_conv_s2_dgrad_tiles(B,H,W)][3][Cin].
SST_API int sst_conv_s2_dgrad_fused(const float* g, const float* y2, const float* cA, const float* cB, const float* cC,
                                    const float* in_scale, const float* in_shift, const float* in_slope, float in_slope_const,
                                    int in_act, float* dy_out, const float* wp, float* dx, const float* epi_y,
                                    const float* epi_scale, const float* epi_shift, const float* epi_slope, float epi_slope_const,
                                    int epi_act, float* epi_partial, int B, int H, int W, int Cin, int Cout, void* stream) {
  SST_REQUIRE(g && y2 && dy_out && (Cout & 3) == 0, "sst_conv_s2_dgrad_fused: g / y2 / dy_out, Cout %% 4 == 0");
  SST_REQUIRE((epi_partial == nullptr) == (epi_y == nullptr), "sst_conv_s2_dgrad_fused: epi_y and epi_partial come together");
  SST_REQUIRE(!cA || (cB && cC), "sst_conv_s2_dgrad_fused: cA / cB / cC come together");
  const S2Fused f{y2, cA, cB, cC, in_scale, in_shift, in_slope, in_slope_const, in_act, dy_out,
                  epi_y, epi_scale, epi_shift, epi_slope, epi_slope_const, epi_act, epi_partial};
  return conv_s2_dgrad_impl(g, wp, dx, B, H, W, Cin, Cout, stream, &f);
}

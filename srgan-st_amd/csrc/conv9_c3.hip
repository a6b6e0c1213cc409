// 9x9 convolutions with a 3-channel side (Generator.conv1 3->C and conv3 C->3, reference model.py:101,127),
// formulated so that the fp32 MFMA tiles are (nearly) full instead of 3/32 occupied.
//
// Trick: fold the horizontal tap index into the GEMM N dimension together with the 3 channels:
//   n = (kx, ch3) -> 27 of 32 columns used (84 %), and the K / M dimension runs over (ky, C) or pixels.
// In an NHWC tensor with 3 channels the 27 values (kx, ch3) of one row are CONTIGUOUS in memory,
// so the "im2col" of the 3-channel side is just a sliding window over a row held in LDS.
//
//   wgrad_c3  : dW of conv1 / conv3     M = C channels, N = (kx,ch3), K = pixels, one pass per ky
//               conv3: dW3[co][ci][ky][kx] = sum P[y,x,ci] * g[y-ky+4, x-kx+4, co]
//               conv1: dW1[co][ci][ky][kx] = sum x3[y+ky-4, x+kx-4, ci] * dz[y,x,co]
// Per-workgroup partial results go to a slab [wg][9][C][32]; c3_reduce sums them in fixed order into the
// reference weight-gradient layout (no atomics: reproducible).
#include "conv_common.h"

namespace {

constexpr int TR = 6;        // rows per workgroup
constexpr int TXMAX = 96;    // pixels per row segment
constexpr int MAXU = 9;      // (ky, channel-fragment) units per wave  -> C <= 128

struct WgC3Args {
  const float* big;        // [B,H,W,C]  (conv3: P = act(u) ; conv1: dz)
  const float* small;      // [B,H,W,3]  (conv3: g      ; conv1: x3)
  float* slab;             // [nwg][9][C][32]
  const float* in_slope;   // activation on `big` (conv3 only): device scalar or null
  float in_slope_const;
  int in_act;
  int kind;                // 0 = conv3 (C -> 3), 1 = conv1 (3 -> C)
  int B, H, W, C;
  int nrc, nxc;            // row chunks, x chunks per image
};

__global__ __launch_bounds__(CONV_NT) void wgrad_c3_kernel(WgC3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int nfrag = (a.C + 31) >> 5;
  const int CL = nfrag * 32 + 4;                 // LDS row stride of the big operand
  const int wg = blockIdx.x;
  const int b = wg / (a.nrc * a.nxc), r1 = wg - b * a.nrc * a.nxc;
  const int rc = r1 / a.nxc, xc = r1 - rc * a.nxc;
  const int y0 = rc * TR, x0 = xc * TXMAX;
  const int tw = min(TXMAX, a.W - x0);           // pixels in this segment
  const int RS = (TXMAX + 8) * 3 + 16;           // band row stride (floats), 8 floats of slack on each side
  float* sB = lds;                               // [TXMAX][CL]
  float* sS = lds + TXMAX * CL;                  // [TR+8][RS]   rows y0-4 .. y0+TR+3, pixels x0-4 .. x0+tw+3
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const int nunits = 9 * nfrag;

  // ---- the small-operand band (zero outside the image)
  for (int i = tid; i < (TR + 8) * RS; i += CONV_NT) {
    const int r = i / RS, o = i - r * RS - 8;    // o: float offset inside the padded row, pixel = o/3 - 4 + x0
    float v = 0.f;
    const int y = y0 - 4 + r;
    if (o >= 0 && o < (tw + 8) * 3 && (unsigned)y < (unsigned)a.H) {
      const int px = o / 3, ch = o - px * 3, x = x0 - 4 + px;
      if ((unsigned)x < (unsigned)a.W) v = a.small[(((size_t)b * a.H + y) * a.W + x) * 3 + ch];
    }
    sS[i] = v;
  }

  f32x16 acc[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

  // per-unit constants (wave-uniform): channel-fragment offset and band-row offset
  int u_fr[MAXU], u_br[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int unit = wave + 4 * u;
    const int ky = unit / nfrag, fr = unit - ky * nfrag;
    u_fr[u] = fr * 32 + li;
    u_br[u] = (4 + (a.kind ? ky - 4 : 4 - ky)) * RS;
  }
  const int c4n = (nfrag * 32) >> 2;             // float4 columns per pixel (zero-filled beyond C)
  const int rows = min(TR, a.H - y0);
  for (int ry = 0; ry < rows; ++ry) {
    const int y = y0 + ry;
    __syncthreads();                             // previous row's MFMAs are done with sB (and the band is written)
    for (int i = tid; i < tw * c4n; i += CONV_NT) {
      const int px = i / c4n, c = (i - px * c4n) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < a.C) {
        v = *reinterpret_cast<const f32x4*>(a.big + (((size_t)b * a.H + y) * a.W + x0 + px) * a.C + c);
        if (a.in_act == ACT_SLOPE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
        }
      }
      *reinterpret_cast<f32x4*>(&sB[px * CL + c]) = v;
    }
    __syncthreads();
    const int npair = (tw + 1) >> 1;
    for (int kk = 0; kk < npair; ++kk) {
      const int px = 2 * kk + lh;
      const bool pv = px < tw;
      // window offset of this lane's column q = li in the band row: conv1: 3*px + q ; conv3: 3*px + 26 - q
      const int so = 8 + 3 * px + (a.kind ? li : 26 - li);
#pragma unroll
      for (int u = 0; u < MAXU; ++u) {
        const int unit = wave + 4 * u;
        if (unit < nunits) {                     // wave-uniform
          const float av = pv ? sB[px * CL + u_fr[u]] : 0.f;
          const float bv = pv ? sS[ry * RS + u_br[u] + so] : 0.f;   // band row 0 is image row y0-4
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial slab: [wg][ky][C][32]
  float* out = a.slab + (size_t)wg * 9 * a.C * 32;
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int unit = wave + 4 * u;
    if (unit < nunits) {
      const int ky = unit / nfrag, fr = unit - ky * nfrag;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = fr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ch < a.C) out[((size_t)ky * a.C + ch) * 32 + li] = acc[u][r];
      }
    }
  }
}

// conv3: dW[co][ci][ky][kx] = sum_wg slab[wg][ky][ci][3kx - co + 2]        (Cout = 3, Cin = C)
// conv1: dW[co][ci][ky][kx] = sum_wg slab[wg][ky][co][3kx + ci]            (Cout = C, Cin = 3)
__global__ __launch_bounds__(256) void c3_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nwg, int C,
                                                        int kind, int accumulate) {
  const int total = 9 * C * 27;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int q = i % 27, t = i / 27, ch = t % C, ky = t / C;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += slab[((size_t)(w * 9 + ky) * C + ch) * 32 + q];
    int co, ci, kx;
    if (kind == 0) {            // q = 3kx - co + 2
      kx = (q + 0) / 3;         // q+.. : co = 3kx + 2 - q in {0,1,2}
      co = 3 * kx + 2 - q;
      if (co > 2) { kx -= 1; co -= 3; }
      if (co < 0) { kx += 1; co += 3; }
      ci = ch;
    } else {                    // q = 3kx + ci
      kx = q / 3;
      ci = q - 3 * kx;
      co = ch;
    }
    const int Cin = kind == 0 ? C : 3;
    float* d = dw + (((size_t)co * Cin + ci) * 9 + ky) * 9 + kx;
    *d = accumulate ? *d + s : s;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_wgrad_c3_supported(int C, int ksize) { return ksize == 9 && C >= 4 && (C & 3) == 0 && C <= 32 * 4 * MAXU / 9; }

SST_API int64_t sst_wgrad_c3_slab_floats(int B, int H, int W, int C) {
  const int nrc = (H + TR - 1) / TR, nxc = (W + TXMAX - 1) / TXMAX;
  return (int64_t)B * nrc * nxc * 9 * C * 32;
}

// kind 0: conv3 (C -> 3): big = conv input [B,H,W,C] (activation applied on load), small = dY [B,H,W,3], dw [3][C][9][9]
// kind 1: conv1 (3 -> C): big = dY [B,H,W,C],                                       small = x  [B,H,W,3], dw [C][3][9][9]
SST_API int sst_wgrad_c3(const float* big, const float* small, float* slab, float* dw, const float* in_slope,
                         float in_slope_const, int in_act, int kind, int B, int H, int W, int C, int accumulate,
                         void* stream) {
  SST_REQUIRE(big && small && slab && dw && B > 0 && H > 0 && W > 0, "sst_wgrad_c3: bad argument");
  SST_REQUIRE(sst_wgrad_c3_supported(C, 9), "sst_wgrad_c3: C=%d not supported", C);
  SST_REQUIRE(kind == 0 || kind == 1, "sst_wgrad_c3: kind");
  WgC3Args a;
  a.big = big; a.small = small; a.slab = slab; a.in_slope = in_slope; a.in_slope_const = in_slope_const; a.in_act = in_act;
  a.kind = kind; a.B = B; a.H = H; a.W = W; a.C = C;
  a.nrc = (H + TR - 1) / TR;
  a.nxc = (W + TXMAX - 1) / TXMAX;
  const int nfrag = (C + 31) / 32;
  const int nwg = B * a.nrc * a.nxc;
  const size_t smem = ((size_t)TXMAX * (nfrag * 32 + 4) + (size_t)(TR + 8) * ((TXMAX + 8) * 3 + 16)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  wgrad_c3_kernel<<<nwg, CONV_NT, smem, sst_stream(stream)>>>(a);
  SST_LAUNCH_CHECK("wgrad_c3_kernel");
  c3_reduce_kernel<<<(9 * C * 27 + 255) / 256, 256, 0, sst_stream(stream)>>>(slab, dw, nwg, C, kind, accumulate);
  SST_LAUNCH_CHECK("c3_reduce_kernel");
  return SST_OK;
}

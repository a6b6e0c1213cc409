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

constexpr int TR = 3;        // rows per workgroup (2 co-resident workgroups per CU at 96x96, B=16)
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

// NU = (ky, channel-fragment) units per wave, compile-time so that the unit loops unroll without branches
// (only the LAST unit of a wave can be missing: 9*nfrag units are dealt round-robin over 4 waves).
template <int NU>
__global__ __launch_bounds__(CONV_NT, (NU <= 5 ? 2 : 1)) void wgrad_c3_kernel(WgC3Args a) {   // <= 256 VGPRs: 2 workgroups per CU
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave id in an SGPR: everything derived from it stays scalar
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
  const bool last_ok = wave + 4 * (NU - 1) < nunits;

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

  f32x16 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

  // per-unit constants (wave-uniform): channel-fragment offset and band-row offset
  int u_fr[NU], u_br[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
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
    if (tw == TXMAX) {
      // Full-width segment (the 96-px case): every LDS address is a per-unit base (computed once per row) plus a compile-time
      // offset, so the unrolled pair loop issues no address arithmetic at all - with ~40 VALU instructions per 5 MFMAs the
      // generic loop below is issue-bound (PMC: MFMA pipe 35 % busy, waves 28 % of their time issuing non-MFMA work).
      const float* pa[NU];
      const float* pb[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        pa[u] = sB + lh * CL + u_fr[u];
        pb[u] = sS + ry * RS + u_br[u] + 8 + 3 * lh + (a.kind ? li : 26 - li);
      }
      float av[NU], bv[NU], an[NU], bn[NU];
#define SST_C3_LOAD(KK, A_, B_)                                                         \
      _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                   \
        if (u < NU - 1 || last_ok) {                                                     \
          A_[u] = pa[u][(KK) * 2 * CL];                                                  \
          B_[u] = pb[u][(KK) * 6];                                                       \
        }                                                                                \
      }
#define SST_C3_MMA(A_, B_)                                                               \
      _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                   \
        if (u < NU - 1 || last_ok) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u], B_[u], acc[u], 0, 0, 0); \
      }
      // CL is a runtime value (C-dependent), so the A offsets are (KK * 2) * CL: one scalar multiply-add per load at most,
      // the B offsets are immediates
      // 6 blocks of 8 pairs: inside a block the offsets are compile-time, between blocks the 2*NU base pointers advance
      // (full unrolling lets the scheduler hoist every load of the row: 490 VGPRs, one wave per SIMD)
      constexpr int BLK = 4;
      static_assert((TXMAX / 2) % BLK == 0, "pair blocks");
      SST_C3_LOAD(0, av, bv)
      for (int k0 = 0; k0 < TXMAX / 2; k0 += BLK) {
#pragma unroll
        for (int kk = 0; kk < BLK; kk += 2) {
          SST_C3_LOAD(kk + 1, an, bn)
          SST_C3_MMA(av, bv)
          if (kk + 2 < BLK) { SST_C3_LOAD(kk + 2, av, bv) }
          else if (k0 + BLK < TXMAX / 2) { SST_C3_LOAD(BLK, av, bv) }      // first pair of the next block
          SST_C3_MMA(an, bn)
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          pa[u] += BLK * 2 * CL;
          pb[u] += BLK * 6;
        }
      }
#undef SST_C3_LOAD
#undef SST_C3_MMA
      continue;
    }
    // operands of pair kk+1 are read from LDS while the (<= MAXU) MFMAs of pair kk run
    float av[NU], bv[NU];
    auto load_pair = [&](int kk, float (&A)[NU], float (&Bv)[NU]) {
      const int px = 2 * kk + lh;
      const bool pv = px < tw;
      const int so = 8 + 3 * px + (a.kind ? li : 26 - li);   // window offset of column q = li: conv1 3*px+q ; conv3 3*px+26-q
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u < NU - 1 || last_ok) {                         // wave-uniform; compile-time true except for the last unit
          A[u] = pv ? sB[px * CL + u_fr[u]] : 0.f;
          Bv[u] = pv ? sS[ry * RS + u_br[u] + so] : 0.f;    // band row 0 is image row y0-4
        }
      }
    };
    // two register sets in ping-pong (no register moves between iterations: the compiler then keeps counted LDS waits
    // instead of draining lgkmcnt before every copy); the overshooting prefetch re-reads the last pair
    float an[NU], bn[NU];
    load_pair(0, av, bv);
    for (int kk = 0; kk < npair; kk += 2) {
      load_pair(kk + 1 < npair ? kk + 1 : npair - 1, an, bn);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u < NU - 1 || last_ok) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u], 0, 0, 0);
      }
      load_pair(kk + 2 < npair ? kk + 2 : npair - 1, av, bv);
      if (kk + 1 < npair) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u < NU - 1 || last_ok) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[u], bn[u], acc[u], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial slab: [wg][ky][C][32]
  float* out = a.slab + (size_t)wg * 9 * a.C * 32;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
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
  // Workgroup = 16 float4 items (256 B of one slab row, fully used cache lines) x 16 slab groups: group sg sums slabs
  // sg, sg+16, ... with independent 16-B loads, the 16 group sums are combined in fixed order through LDS (reproducible).
  // (The first version gave 16 lanes ONE float of 16 different slabs each: 4 used floats per 64-B line, 2.3 TB/s.)
  __shared__ f32x4 part[16][16];
  const int item = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int total4 = 9 * C * 8;                               // float4 items of one slab ([9][C][32] floats)
  const int i4 = blockIdx.x * 16 + item;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slab) + i4;
#pragma unroll 8
    for (int w = sg; w < nwg; w += 16) t += src[(size_t)w * total4];
  }
  part[sg][item] = t;
  __syncthreads();
  if (sg == 0 && i4 < total4) {
#pragma unroll
    for (int g = 1; g < 16; ++g) t += part[g][item];
    const int q0 = (i4 & 7) * 4, ch = (i4 >> 3) % C, ky = (i4 >> 3) / C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = q0 + e;
      if (q >= 27) continue;
      int co, ci;
      const int kx = q / 3;
      if (kind == 0) {            // q = 3kx - co + 2  ->  co = 3kx + 2 - q in {0,1,2}
        co = 3 * kx + 2 - q;
        ci = ch;
      } else {                    // q = 3kx + ci
        ci = q - 3 * kx;
        co = ch;
      }
      const int Cin = kind == 0 ? C : 3;
      float* d = dw + (((size_t)co * Cin + ci) * 9 + ky) * 9 + kx;
      *d = accumulate ? *d + t[e] : t[e];
    }
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
  const int nu = (9 * nfrag + 3) / 4;     // units per wave (3, 5, 7 or 9)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c3_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c3_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c3_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipStream_t st = sst_stream(stream);
  if (nu <= 3) wgrad_c3_kernel<3><<<nwg, CONV_NT, smem, st>>>(a);
  else if (nu <= 5) wgrad_c3_kernel<5><<<nwg, CONV_NT, smem, st>>>(a);
  else if (nu <= 7) wgrad_c3_kernel<7><<<nwg, CONV_NT, smem, st>>>(a);
  else wgrad_c3_kernel<9><<<nwg, CONV_NT, smem, st>>>(a);
  SST_LAUNCH_CHECK("wgrad_c3_kernel");
  c3_reduce_kernel<<<(9 * C * 8 + 15) / 16, 256, 0, sst_stream(stream)>>>(slab, dw, nwg, C, kind, accumulate);
  SST_LAUNCH_CHECK("c3_reduce_kernel");
  return SST_OK;
}

// =================================================================================================
// conv9_c3_fwd: 9x9 convolution FROM a 3-channel NHWC tensor TO Cout channels (Generator.conv1 forward,
// model.py:101; and the data-gradient of conv3, model.py:127, with transposed+rotated weights).
//   GEMM: M = pixels, N = Cout, K = (ky, j) with j = 3*kx + ch3 in [0,27) padded to 32  ->  36 chunks of 8
//   (the generic kernel spends 81 chunks on the same work because a tap only carries 3 of 8 k-lanes).
// Workgroup = 4 rows x 32 pixels, one wave per row, one 32x32 accumulator per wave (no K split, no LDS
// reduction).  The 3-channel input band (4+8 rows) sits in LDS; the A fragment is a sliding window over it.
namespace {

constexpr int F_TW = 32, F_TH = 4;
constexpr int F_RS = (F_TW + 8) * 3 + 24;     // band row stride: window reads reach 3*31 + 31 = 124 < F_RS

struct C3FwdArgs {
  const float* x;      // [B,H,W,3]
  const float* wp;     // packed: [(Cout+31)/32][9][4][64 lanes][4]
  float* y;            // [B,H,W,Cout]
  const float* bias;   // [Cout] or null
  int B, H, W, Cout;
};

__global__ __launch_bounds__(CONV_NT) void conv9_c3_fwd_kernel(C3FwdArgs a) {
  __shared__ float sS[(F_TH + 8) * F_RS];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave id in an SGPR: everything derived from it stays scalar
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_x = (a.W + F_TW - 1) / F_TW, tiles_y = (a.H + F_TH - 1) / F_TH;
  const int mt = blockIdx.x;
  const int b = mt / (tiles_x * tiles_y), rt = mt - b * tiles_x * tiles_y;
  const int y0 = (rt / tiles_x) * F_TH, x0 = (rt % tiles_x) * F_TW;
  const int nf = blockIdx.y;
  const float* wblk = a.wp + (size_t)nf * 36 * 256 + lane * 4;

  // B fragments: 36 chunks, 3-deep ping-pong like the generic kernel (36 = 6 x 6: no tail)
  int p_i = 0;
  auto pf_load = [&]() {
    const f32x4 v = *reinterpret_cast<const f32x4*>(wblk + (size_t)(p_i < 36 ? p_i : 35) * 256);
    ++p_i;
    return v;
  };
  f32x4 A0 = pf_load(), A1 = pf_load(), A2 = pf_load(), B0, B1, B2;

  for (int i = tid; i < (F_TH + 8) * F_RS; i += CONV_NT) {
    const int r = i / F_RS, o = i - r * F_RS;
    float v = 0.f;
    const int y = y0 - 4 + r;
    if (o < (F_TW + 8) * 3 && (unsigned)y < (unsigned)a.H) {
      const int px = o / 3, ch = o - px * 3, x = x0 - 4 + px;
      if ((unsigned)x < (unsigned)a.W) v = a.x[(((size_t)b * a.H + y) * a.W + x) * 3 + ch];
    }
    sS[i] = v;
  }
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // window of output pixel li for (ky, c4): band row (wave + ky), floats [3*li + 8*c4 + 4*lh, +4)
  int a_off = wave * F_RS + 3 * li + 4 * lh, a_c4 = 0;
  auto a_load = [&]() {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sS[a_off + j];
    const bool wrap = (a_c4 == 3);
    a_off += wrap ? F_RS - 24 : 8;
    a_c4 = wrap ? 0 : a_c4 + 1;
    return v;
  };
#define SST_C3_CHUNK(BUSE, BLOAD)                                                                      \
  {                                                                                                    \
    const f32x4 av = a_load();                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], BUSE[j], acc, 0, 0, 0);                      \
    BLOAD = pf_load();                                                                                 \
  }
  for (int c = 0; c < 36; c += 6) {
    SST_C3_CHUNK(A0, B0)
    SST_C3_CHUNK(A1, B1)
    SST_C3_CHUNK(A2, B2)
    SST_C3_CHUNK(B0, A0)
    SST_C3_CHUNK(B1, A1)
    SST_C3_CHUNK(B2, A2)
  }
#undef SST_C3_CHUNK
  // ---- store: acc[r] = out[pixel (r&3)+8(r>>2)+4lh][cout li]
  const int oy = y0 + wave, co = nf * 32 + li;
  if (oy < a.H && co < a.Cout) {
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ox < a.W) a.y[(((size_t)b * a.H + oy) * a.W + ox) * a.Cout + co] = acc[r] + bv;
    }
  }
}

// w [Cout][Cin][9][9] -> packed for conv9_c3_fwd.  mode 0: Cin must be 3 (conv1 forward).
// mode 1: Cout must be 3 (data-gradient of conv3): outputs = Cin, inputs = Cout, taps rotated 180 degrees.
__global__ void pack_c3_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int mode) {
  const int total = (int)c3_packed_floats(mode ? Cin : Cout);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
    wp[idx] = pack_c3_value(w, idx, Cout, Cin, mode);
}

}  // namespace

SST_API int64_t sst_conv9_c3_packed_floats(int Cout_eff) { return c3_packed_floats(Cout_eff); }

SST_API int sst_conv9_c3_pack(const float* w, float* wp, int Cout, int Cin, int mode, void* stream) {
  SST_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (mode == 0 ? Cin == 3 : Cout == 3), "sst_conv9_c3_pack: the 3-channel side is missing");
  const int O = mode ? Cin : Cout;
  const int total = ((O + 31) / 32) * 36 * 256;
  pack_c3_fwd_kernel<<<(total + 255) / 256, 256, 0, sst_stream(stream)>>>(w, wp, Cout, Cin, mode);
  SST_LAUNCH_CHECK("pack_c3_fwd_kernel");
  return SST_OK;
}

// y [B,H,W,Cout] = conv9x9(x [B,H,W,3]) (+ bias), pad 4, stride 1
SST_API int sst_conv9_c3_fwd(const float* x, const float* wp, float* y, const float* bias, int B, int H, int W, int Cout,
                             void* stream) {
  SST_REQUIRE(x && wp && y && B > 0 && H > 0 && W > 0 && Cout > 0, "sst_conv9_c3_fwd: bad argument");
  C3FwdArgs a;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias; a.B = B; a.H = H; a.W = W; a.Cout = Cout;
  const int64_t mt = (int64_t)B * ((H + F_TH - 1) / F_TH) * ((W + F_TW - 1) / F_TW);
  SST_REQUIRE(mt < (1ll << 31), "sst_conv9_c3_fwd: too many tiles");
  conv9_c3_fwd_kernel<<<dim3((unsigned)mt, (Cout + 31) / 32), CONV_NT, 0, sst_stream(stream)>>>(a);
  SST_LAUNCH_CHECK("conv9_c3_fwd_kernel");
  return SST_OK;
}

// =================================================================================================
// conv9_to3_fwd: 9x9 convolution from C channels TO 3 channels (Generator.conv3 forward + clamp, model.py:127,148-150).
//   Step 1 (MFMA): T[y][x'][q] = sum_{ky,ci} P[y+ky-4][x'][ci] * W[co][ci][ky][kx],  q = 3*kx + co  (27 of 32 columns)
//   Step 2 (LDS) : out[y][x][co] = bias[co] + sum_kx T[y][x+kx-4][3*kx+co]
// Workgroup = 512 threads = 8 rows x 24 output pixels (32-pixel M fragment incl. the +-4 halo), one wave per row;
// the 16-row x 32-pixel input patch is staged 32 channels at a time in LDS (74 KB) and reused by all 9 ky.
namespace {

constexpr int T3_TH = 8, T3_TW = 24, T3_NT = 512;     // 8 waves, one output row each (4 waves x 2 rows measured slower: 103 vs 88 us)
constexpr int T3_PW = 32, T3_PH = T3_TH + 8;
// input channels staged per pass: 32 (two passes per 64-block) keeps the patch at 74 KB, so TWO workgroups fit a CU and one's
// staging / fold overlaps the other's MFMAs (with 64 channels = 139 KB there is one workgroup per CU and nothing overlaps)
constexpr int T3_CB = 32, T3_LDSC = T3_CB + 4;

struct To3Args {
  const float* x;        // [B,H,W,C]
  const float* wp;       // packed [ncb][9][8][64 lanes][4] (+ PACK_PAD zeros)
  float* y;              // [B,3,H,W]  clamp(out,0,1)
  float* y_pre;          // [B,3,H,W]  out (saved for backward) or null
  const float* bias;     // [3] or null
  const float* in_slope; // device scalar or null
  float in_slope_const;
  int in_act;
  int B, H, W, C;
};

__global__ __launch_bounds__(T3_NT) void conv9_to3_fwd_kernel(To3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [T3_PH][T3_PW][LDSC]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave id in an SGPR: everything derived from it stays scalar
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_x = (a.W + T3_TW - 1) / T3_TW, tiles_y = (a.H + T3_TH - 1) / T3_TH;
  const int mt = blockIdx.x;
  const int b = mt / (tiles_x * tiles_y), rt = mt - b * tiles_x * tiles_y;
  const int y0 = (rt / tiles_x) * T3_TH, x0 = (rt % tiles_x) * T3_TW;
  const int ncb = (a.C + CB - 1) / CB;
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const float* wzero = a.wp + (size_t)ncb * 9 * 8 * 256 + lane * 4;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int pass = 0; pass < 2 * ncb; ++pass) {
    const int cb = pass >> 1, half = pass & 1;
    const int c0 = cb * CB + half * T3_CB;
    if (c0 >= a.C) break;
    const int nks = (min(T3_CB, a.C - c0) + 7) >> 3;
    const int nchunks = 9 * nks;
    const float* wblk = a.wp + (size_t)cb * 9 * 8 * 256 + lane * 4;     // packed per 64-block: chunk (ky, ks) at (ky*8 + ks)*256
    int p_ks = 0, p_off = half * 4 * 256, p_i = 0;
    auto pf_load = [&]() {
      const float* src = p_i < nchunks ? wblk + p_off : wzero;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      const bool wrap = (p_ks + 1 == nks);
      p_ks = wrap ? 0 : p_ks + 1;
      p_off += wrap ? (9 - nks) * 256 : 256;
      ++p_i;
      return v;
    };
    f32x4 A0 = pf_load(), A1 = pf_load(), A2 = pf_load(), B0, B1, B2;
    if (pass) __syncthreads();
    {
      const int c4 = (tid & 7) * 4, c = c0 + c4;
      for (int p = tid >> 3; p < T3_PH * T3_PW; p += T3_NT / 8) {
        const int py = p / T3_PW, px = p - py * T3_PW;
        const int iy = y0 - 4 + py, ix = x0 - 4 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c < a.C) {
          v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)b * a.H + iy) * a.W + ix) * a.C + c);
          if (a.in_act == ACT_SLOPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
          }
        }
        *reinterpret_cast<f32x4*>(&lds[p * T3_LDSC + c4]) = v;
      }
    }
    __syncthreads();
    // A cursor: patch row (wave + ky), pixel li, channels ks*8 + 4*lh
    int a_off = (wave * T3_PW + li) * T3_LDSC + 4 * lh, a_ks = 0, a_i = 0;
    auto a_load = [&]() {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[a_off]);
      const bool live = a_i + 1 < nchunks;
      const bool wrap = (a_ks + 1 == nks);
      a_off += live ? (wrap ? T3_PW * T3_LDSC - 8 * (nks - 1) : 8) : 0;
      a_ks = wrap ? 0 : a_ks + 1;
      ++a_i;
      return v;
    };
#define SST_T3_CHUNK(BUSE, BLOAD)                                                                      \
    {                                                                                                  \
      const f32x4 av = a_load();                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], BUSE[j], acc, 0, 0, 0);                    \
      BLOAD = pf_load();                                                                               \
    }
    for (int c = 0; c < nchunks; c += 6) {
      SST_T3_CHUNK(A0, B0)
      SST_T3_CHUNK(A1, B1)
      SST_T3_CHUNK(A2, B2)
      SST_T3_CHUNK(B0, A0)
      SST_T3_CHUNK(B1, A1)
      SST_T3_CHUNK(B2, A2)
    }
#undef SST_T3_CHUNK
  }
  // ---- T -> LDS (overlays the patch), then the horizontal fold
  __syncthreads();
  float* sT = lds;   // [T3_TH][32][33]
#pragma unroll
  for (int r = 0; r < 16; ++r) sT[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
  __syncthreads();
  for (int i = tid; i < T3_TH * 3 * T3_TW; i += T3_NT) {
    const int x = i % T3_TW, t = i / T3_TW, co = t % 3, r = t / 3;
    const int oy = y0 + r, ox = x0 + x;
    if (oy < a.H && ox < a.W) {
      float s = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int kx = 0; kx < 9; ++kx) s += sT[(r * 32 + x + kx) * 33 + 3 * kx + co];
      const size_t o = (((size_t)b * 3 + co) * a.H + oy) * a.W + ox;
      if (a.y_pre) a.y_pre[o] = s;
      a.y[o] = fminf(fmaxf(s, 0.f), 1.f);
    }
  }
}

// w [3][C][9][9] -> packed [ncb][ky][ks 0..7][lane][4]: B[k=(ky,ci)][n=3kx+co] (+ PACK_PAD zeros at the end)
__global__ void pack_to3_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int64_t total) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    wp[idx] = pack_to3_value(w, idx, C);
}

}  // namespace

SST_API int64_t sst_conv9_to3_packed_floats(int C) { return to3_packed_floats(C); }

SST_API int sst_conv9_to3_pack(const float* w, float* wp, int C, void* stream) {
  SST_REQUIRE(w && wp && C > 0, "sst_conv9_to3_pack: bad argument");
  const int64_t total = sst_conv9_to3_packed_floats(C);
  pack_to3_kernel<<<(int)((total + 255) / 256), 256, 0, sst_stream(stream)>>>(w, wp, C, total);
  SST_LAUNCH_CHECK("pack_to3_kernel");
  return SST_OK;
}

// y [B,3,H,W] = clamp(conv9x9(act(x [B,H,W,C])) + bias, 0, 1);  y_pre = pre-clamp copy (or null)
SST_API int sst_conv9_to3_fwd(const float* x, const float* wp, float* y, float* y_pre, const float* bias, const float* in_slope,
                              float in_slope_const, int in_act, int B, int H, int W, int C, void* stream) {
  SST_REQUIRE(x && wp && y && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "sst_conv9_to3_fwd: bad argument");
  To3Args a;
  a.x = x; a.wp = wp; a.y = y; a.y_pre = y_pre; a.bias = bias; a.in_slope = in_slope; a.in_slope_const = in_slope_const;
  a.in_act = in_act; a.B = B; a.H = H; a.W = W; a.C = C;
  const int64_t mt = (int64_t)B * ((H + T3_TH - 1) / T3_TH) * ((W + T3_TW - 1) / T3_TW);
  SST_REQUIRE(mt < (1ll << 31), "sst_conv9_to3_fwd: too many tiles");
  const size_t smem = (size_t)T3_PH * T3_PW * T3_LDSC * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv9_to3_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_set = true;
  }
  conv9_to3_fwd_kernel<<<(unsigned)mt, T3_NT, smem, sst_stream(stream)>>>(a);
  SST_LAUNCH_CHECK("conv9_to3_fwd_kernel");
  return SST_OK;
}

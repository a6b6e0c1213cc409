// Weight gradient of the 3x3 (or any KSxKS, pad KS/2) NHWC convolution: fp32 MFMA, split-K over pixel chunks.
//
// Replaces autograd's cuDNN/oneDNN wgrad for every nn.Conv2d of the hot path (reference model.py:32-56,
// 113,159,173,176):     dW[co][ci][ky][kx] = sum_{b,oy,ox} X'[b, oy*S+ky-P, ox*S+kx-P, ci] * dY[b,oy,ox,co]
// where X' = act(x*in_scale+in_shift) is the conv's (virtual, re-computed while staging) input.
//
// GEMM view per tap: M = Cout, N = Cin, K = pixels.  Workgroup = (pixel chunk, tap, 64x64 block of
// (co,ci)); wave w owns one 32x32 v_mfma_f32_32x32x2_f32 accumulator.  Pixel sub-tiles of 64 are
// staged in LDS ([px][channel], so both MFMA operands are conflict-free ds_read_b32 columns).
// Each workgroup writes its partial 64x64 tile to a slab [chunk][tap][Cout][Cin]; wgrad_reduce sums the
// chunks in fixed order into the reference layout [Cout][Cin][KS][KS]  (no atomics: reproducible).
#include "conv_common.h"
#include <cstring>
#include <cstdlib>

namespace {

constexpr int SUB = 64;          // pixels per staged sub-tile
constexpr int WLD = 64 + 4;      // LDS row stride (floats)

// Grouped launch: many layers of identical shape in ONE launch (their per-layer pointers live in a device table).
struct WgJob {
  const float* x; const float* dy; float* slab; float* dw;
  const float* in_scale; const float* in_shift; const float* in_slope;
  float in_slope_const; int in_act;
};

// Job table of a grouped launch, passed BY VALUE as a kernel argument (2.5 KB of the 4 KB kernarg segment): no device-side
// table, hence no table-writing launch in front of every grouped weight gradient (it was a 4.5 us kernel per step).
constexpr int WG_TAB_MAX = 40;
struct WgJobTab { WgJob j[WG_TAB_MAX]; };

struct WgradArgs {
  int grouped;             // 0: single layer (fields below); else blockIdx.x = job * nchunk + chunk, jobs in the WgJobTab argument
  int nchunk;
  const float* x;          // [B,H,W,Cin]
  const float* dy;         // [B,Ho,Wo,Cout]
  float* slab;             // [nchunk][KK][Cout][Cin]
  const float* in_scale;   // [Cin] or null
  const float* in_shift;
  const float* in_slope;   // device scalar or null
  float in_slope_const;
  int in_act;
  int B, H, W, Cin, Cout, Ho, Wo, stride, KS, pad;
  int chunk_px;            // pixels per chunk (multiple of SUB)
  int R, bpc, nbands;      // band kernel: rows per band, bands per chunk, bands in the tensor (B*H/R)
  int dbg;                 // dev ablation bits from $SST_WGRAD_DBG (0 in production): 1 no global loads, 2 no MFMA, 4 no slab store
};

// VEC: Cin % 4 == 0 and Cout % 4 == 0 (every hot-path layer) - the scalar-tail code is compiled out.
template <bool VEC>
__global__ __launch_bounds__(CONV_NT, 4) void conv_wgrad_kernel(WgradArgs a, WgJobTab tab) {   // <= 128 registers incl. the 16 accumulator AGPRs: four workgroups per CU (it sat at 138 = three)
  int chunk = blockIdx.x;
  if (a.grouped) {                    // grouped: fetch this workgroup's layer (workgroup-uniform scalar loads)
    const int job = blockIdx.x / a.nchunk;
    chunk = blockIdx.x - job * a.nchunk;
    const WgJob jb = tab.j[job];
    a.x = jb.x; a.dy = jb.dy; a.slab = jb.slab; a.in_scale = jb.in_scale; a.in_shift = jb.in_shift;
    a.in_slope = jb.in_slope; a.in_slope_const = jb.in_slope_const; a.in_act = jb.in_act;
  }
  __shared__ __attribute__((aligned(16))) float sX[SUB * WLD];
  __shared__ __attribute__((aligned(16))) float sD[SUB * WLD];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave id in an SGPR: everything derived from it stays scalar
  const int li = lane & 31, lh = lane >> 5;
  const int tap = blockIdx.y, ky = tap / a.KS, kx = tap - ky * a.KS;
  const int nci = (a.Cin + 63) / 64;
  const int cob = blockIdx.z / nci, cib = blockIdx.z - cob * nci;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int64_t M = (int64_t)a.B * a.Ho * a.Wo;
  const int64_t m_begin = (int64_t)chunk * a.chunk_px;
  const int64_t m_end = min(M, m_begin + a.chunk_px);
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const bool xvec = VEC || (a.Cin & 3) == 0, dvec = VEC || (a.Cout & 3) == 0;
  const int wco = (wave & 1) * 32, wci = (wave >> 1) * 32;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // Register-staged software pipeline: the global loads of sub-tile t+1 are in flight while sub-tile t's
  // 32 MFMAs run out of LDS (one register set, written to LDS after the barrier that retires the reads of t).
  f32x4 rx[4], rd[4];
  unsigned rvalid = 0;     // bit u: slot u holds an in-image pixel (padding must stay exactly zero)
  // Per-slot pixel cursors (b, oy, ox) are kept incrementally: one 32-bit division per slot up front, then
  // "advance by SUB pixels with carries" per sub-tile - the div/mod chains used to cost as many VALU cycles as
  // the MFMAs of the sub-tile.
  const int HW = a.Ho * a.Wo;
  int sb[4], soy[4], sox[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned m = (unsigned)m_begin + ((tid + u * CONV_NT) >> 4);
    sb[u] = (int)(m / (unsigned)HW);
    const unsigned rem = m - (unsigned)sb[u] * HW;
    soy[u] = (int)(rem / (unsigned)a.Wo);
    sox[u] = (int)(rem - (unsigned)soy[u] * a.Wo);
  }
  const int adv_y = SUB / a.Wo, adv_x = SUB - adv_y * a.Wo;      // SUB pixels = adv_y rows + adv_x columns
  const int c4 = (tid & 15) * 4;
  auto stage_load = [&](int64_t m0) {
    rvalid = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = (tid + u * CONV_NT) >> 4;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
      if ((int64_t)m0 + p < m_end && !(a.dbg & 1)) {
        const int b = sb[u], oy = soy[u], ox = sox[u];
        const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        const int c = ci0 + c4;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && c < a.Cin) {
          const float* src = a.x + (((size_t)b * a.H + iy) * a.W + ix) * a.Cin + c;
          if (VEC || xvec) {
            xv = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (c + j < a.Cin) xv[j] = src[j];
          }
          rvalid |= 1u << u;
        }
        const int co = co0 + c4;
        if (co < a.Cout) {
          const float* src = a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
          if (VEC || dvec) {
            dv = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (co + j < a.Cout) dv[j] = src[j];
          }
        }
      }
      rx[u] = xv;
      rd[u] = dv;
      // advance this slot's cursor by SUB pixels
      int ox = sox[u] + adv_x, oy = soy[u] + adv_y;
      if (ox >= a.Wo) { ox -= a.Wo; ++oy; }
      if (oy >= a.Ho) { oy -= a.Ho; ++sb[u]; if (oy >= a.Ho) { sb[u] += oy / a.Ho; oy %= a.Ho; } }
      sox[u] = ox;
      soy[u] = oy;
    }
  };
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};   // a thread's channel quad is fixed (CONV_NT % 16 == 0)
  {
    const int c = ci0 + (tid & 15) * 4;
    if (a.in_scale) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < a.Cin) {
          sc4[j] = a.in_scale[c + j];
          sh4[j] = a.in_shift[c + j];
        }
    }
  }
  auto stage_store = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = tid + u * CONV_NT;
      const int p = q >> 4, c4 = (q & 15) * 4;
      f32x4 xv = rx[u];
      if ((rvalid >> u) & 1u) {
        const int c = ci0 + c4;
        if (a.in_scale) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[j] = fmaf(xv[j], sc4[j], sh4[j]);
        }
        if (a.in_act == ACT_SLOPE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[j] = xv[j] > 0.f ? xv[j] : xv[j] * slope;
        }
        if (!VEC) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j >= a.Cin) xv[j] = 0.f;
        }
      } else {
        xv = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(&sX[p * WLD + c4]) = xv;
      *reinterpret_cast<f32x4*>(&sD[p * WLD + c4]) = rd[u];
    }
  };
  if (m_begin < m_end) stage_load(m_begin);
  for (int64_t m0 = m_begin; m0 < m_end; m0 += SUB) {
    __syncthreads();                 // MFMAs of the previous sub-tile are done reading LDS
    stage_store();
    __syncthreads();
    if (m0 + SUB < m_end) stage_load(m0 + SUB);
    // ---- 32 MFMAs: K = 64 pixels, 2 per instruction (lane half lh picks the pixel of the pair)
    if (!(a.dbg & 2)) {
      // fragments of step kk+1 are read from LDS while the MFMA of step kk runs (explicit 1-deep software pipeline)
      float av = sD[lh * WLD + wco + li], bv = sX[lh * WLD + wci + li];
#pragma unroll
      for (int kk = 0; kk < SUB / 2; ++kk) {
        const int kn = (kk + 1 < SUB / 2) ? kk + 1 : kk;
        const float an = sD[(2 * kn + lh) * WLD + wco + li];
        const float bn = sX[(2 * kn + lh) * WLD + wci + li];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        av = an;
        bv = bn;
      }
    }
  }
  // ---- store the partial tile: rows = co, cols = ci (lanes contiguous along ci)
  if (a.dbg & 4) {
    if (acc[0] == 12345.f) a.slab[0] = 1.f;
    return;
  }
  float* out = a.slab + ((size_t)chunk * gridDim.y + tap) * a.Cout * a.Cin;
  const int ci = ci0 + wci + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (co < a.Cout && ci < a.Cin) out[(size_t)co * a.Cin + ci] = acc[r];
  }
}

// ---- "all taps" variant for 3x3 / stride 1 / pad 1 with Cin, Cout multiples of 64 (every trunk / up-sampler conv of the
// generator, the stride-1 convs of the discriminator).  A workgroup owns a 64x64 (co,ci) block for ALL 9 taps: wave w keeps
// 9 accumulators (one 32x32 block per tap, 144 AGPRs), so the dY tile and the X patch of a band of R image rows are staged
// ONCE and feed 9 MFMAs per pixel pair - the kernel above stages both once per tap.  The workgroup walks `bpc` consecutive
// bands, prefetching the next band into registers while the MFMAs of the current one run, and writes one slab tile per tap
// at the end (same slab layout, same reduce kernel).
constexpr int WB_XS = 10, WB_DS = 3;      // max register slots (16 B each) per thread for the X patch / dY tile of one band

// (XS = 10 under the 256-VGPR cap of two waves per SIMD spilled 11 registers to scratch memory; a kernel with a private segment is
// not safe inside the two-branch hipGraph of the iteration - see DESIGN.md section 5, tests/test_abi_symbols.py - so that instance gets the
// whole register file: one workgroup per CU, which its 80 KB of LDS nearly forced anyway)
template <int XS>
__global__ __launch_bounds__(CONV_NT, (XS > 7 ? 1 : 2)) void conv_wgrad_band_kernel(WgradArgs a, WgJobTab tab) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  int chunk = blockIdx.x;
  if (a.grouped) {
    const int job = blockIdx.x / a.nchunk;
    chunk = blockIdx.x - job * a.nchunk;
    const WgJob jb = tab.j[job];
    a.x = jb.x; a.dy = jb.dy; a.slab = jb.slab; a.in_scale = jb.in_scale; a.in_shift = jb.in_shift;
    a.in_slope = jb.in_slope; a.in_slope_const = jb.in_slope_const; a.in_act = jb.in_act;
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int R = a.R, W = a.W, PW = W + 2, H = a.H;
  const int npx_x = (R + 2) * PW, npx_d = R * W;
  float* sX = wlds;
  float* sD = wlds + npx_x * WLD;
  const int nci = a.Cin >> 6;
  const int cob = blockIdx.y / nci, cib = blockIdx.y - cob * nci;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int wco = (wave & 1) * 32, wci = (wave >> 1) * 32;
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const int bands_per_img = H / R;
  const int band_begin = chunk * a.bpc, band_end = min(a.nbands, band_begin + a.bpc);

  // staging slots: q = tid + u*256 -> patch pixel q/16, channel quad q%16 (a thread's channel quad is fixed)
  const int c4 = (tid & 15) * 4;
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (a.in_scale) {
    sc4 = *reinterpret_cast<const f32x4*>(a.in_scale + ci0 + c4);
    sh4 = *reinterpret_cast<const f32x4*>(a.in_shift + ci0 + c4);
  }
  // Slot validity: the halo columns are never valid; patch row 0 / R+1 fall outside the image for the first / last band of
  // an image.  Three static bit masks over the slots replace per-slot row bookkeeping.
  int xoff[XS];             // element offset of the slot's pixel relative to the band origin pixel (b, y0, 0)
  unsigned m_col = 0, m_top = 0, m_bot = 0;
#pragma unroll
  for (int u = 0; u < XS; ++u) {
    const int p = (tid + u * CONV_NT) >> 4;
    const int py = p / PW, px = p - py * PW;
    if (p < npx_x && px >= 1 && px <= W) m_col |= 1u << u;
    if (py == 0) m_top |= 1u << u;
    if (py == R + 1) m_bot |= 1u << u;
    xoff[u] = ((py - 1) * W + (px - 1)) * a.Cin;
  }
  f32x4 rx[XS], rd[WB_DS];
  unsigned rvalid = 0;
  auto stage_load = [&](int band) {
    const int b = band / bands_per_img, y0 = (band - b * bands_per_img) * R;
    const float* xb = a.x + ((size_t)(b * H + y0) * W) * a.Cin + ci0 + c4;
    const float* db = a.dy + ((size_t)(b * H + y0) * W) * a.Cout + co0 + c4;
    rvalid = m_col & ~(y0 == 0 ? m_top : 0u) & ~(y0 + R == H ? m_bot : 0u);
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const bool ok = (rvalid >> u) & 1u;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (ok ? xoff[u] : 0));   // invalid slots read the band origin (in bounds)
      rx[u] = v;
    }
#pragma unroll
    for (int u = 0; u < WB_DS; ++u) {
      const int p = (tid + u * CONV_NT) >> 4;
      rd[u] = *reinterpret_cast<const f32x4*>(db + (size_t)(p < npx_d ? p : 0) * a.Cout);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const int p = (tid + u * CONV_NT) >> 4;
      if (p >= npx_x) break;
      f32x4 xv = rx[u];
      if (a.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = fmaf(xv[j], sc4[j], sh4[j]);
      }
      if (a.in_act == ACT_SLOPE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xv[j] > 0.f ? xv[j] : xv[j] * slope;
      }
      if (!((rvalid >> u) & 1u)) xv = f32x4{0.f, 0.f, 0.f, 0.f};      // padding stays exactly zero
      *reinterpret_cast<f32x4*>(&sX[p * WLD + c4]) = xv;
    }
#pragma unroll
    for (int u = 0; u < WB_DS; ++u) {
      const int p = (tid + u * CONV_NT) >> 4;
      if (p < npx_d) *reinterpret_cast<f32x4*>(&sD[p * WLD + c4]) = rd[u];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-lane LDS bases (floats) and wave-uniform tap offsets
  const int la = lh * WLD + wco + li, lb = lh * WLD + wci + li;
  int tofs[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tofs[t] = ((t / 3) * PW + (t % 3)) * WLD;
  const int halfw = W >> 1, npairs = R * halfw;

  if (band_begin < band_end) stage_load(band_begin);
  for (int band = band_begin; band < band_end; ++band) {
    __syncthreads();                 // the MFMAs of the previous band are done reading LDS
    stage_store();
    __syncthreads();
    if (band + 1 < band_end && !(a.dbg & 1)) stage_load(band + 1);
    if (a.dbg & 2) continue;
    // ---- pixel pairs (r, 2*x2 + lh): A = dY, B_t = X shifted by tap t.  Fragments of pair i+1 are read while pair i's MFMAs run.
    int pa = 0, pb = 0, x2 = 0;      // wave-uniform cursors (floats): dY pair base, X pair base (tap 0,0), pair column
    auto advance = [&]() {
      pa += 2 * WLD;
      pb += 2 * WLD;
      if (++x2 == halfw) { x2 = 0; pb += 2 * WLD; }
    };
    float av0 = sD[pa + la], av1;
    float bv0[9], bv1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) bv0[t] = sX[pb + tofs[t] + lb];
    for (int i = 0; i < npairs; i += 2) {
      advance();
      av1 = sD[pa + la];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv1[t] = sX[pb + tofs[t] + lb];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0[t], acc[t], 0, 0, 0);
      if (i + 2 < npairs) advance();          // the prefetch after the last pair re-reads the last pair (stays in bounds)
      av0 = sD[pa + la];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv0[t] = sX[pb + tofs[t] + lb];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1[t], acc[t], 0, 0, 0);
    }
  }

  // ---- one slab tile per tap: rows = co, cols = ci (lanes contiguous along ci)
  if (a.dbg & 4) {
    if (acc[0][0] == 12345.f) a.slab[0] = 1.f;
    return;
  }
  const int ci = ci0 + wci + li;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* out = a.slab + ((size_t)chunk * 9 + t) * a.Cout * a.Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[(size_t)co * a.Cin + ci] = acc[t][r];
    }
  }
}

// ---- 3x3 / stride 1 weight gradient of a layer with a 3-CHANNEL input (Discriminator.features[0], model.py:32): MFMA tiles
// would be 3/64 occupied (the general kernel runs this shape at 2 TFLOP/s), so this one is plain VALU: a workgroup owns a
// band of R image rows, the 3-channel input patch sits in LDS, thread (co = tid % 64, pixel lane = tid / 64) keeps the 27
// accumulators dW[co][ci][ky][kx] and walks its pixels: one coalesced dY load + 27 FMAs on LDS broadcasts per pixel.
// Partials go to the usual slab [chunk = band][tap][Cout][3]; wgrad_reduce_kernel sums them.
constexpr int C3_ROWS = 8;
__global__ __launch_bounds__(CONV_NT) void wgrad_k3c3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ slab, int B, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float xs[];        // [(R+2)][(W+2)*3 + 3]  zero-padded patch
  __shared__ float red[3][64][28];
  const int bands = (H + C3_ROWS - 1) / C3_ROWS;
  const int b = blockIdx.x / bands, y0 = (blockIdx.x - b * bands) * C3_ROWS;
  const int rows = min(C3_ROWS, H - y0);
  const int RS = (W + 2) * 3 + 3;
  for (int i = threadIdx.x; i < (C3_ROWS + 2) * RS; i += CONV_NT) {
    const int r = i / RS, o = i - r * RS;
    const int px = o / 3 - 1, y = y0 - 1 + r;
    float v = 0.f;
    if (o < (W + 2) * 3 && (unsigned)px < (unsigned)W && (unsigned)y < (unsigned)H)
      v = x[(((size_t)b * H + y) * W + px) * 3 + (o - (px + 1) * 3)];
    xs[i] = v;
  }
  __syncthreads();
  const int co = threadIdx.x & 63, pl = threadIdx.x >> 6;
  for (int cob = blockIdx.y * 64; cob < Cout; cob += gridDim.y * 64) {
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    if (cob + co < Cout) {
      // pixels p = pl + 4*i of the band; 8 dY loads in flight per thread (a one-load-per-iteration loop is pure latency)
      constexpr int UNR = 8;
      const int npx = rows * W;
      const float* dyb = dy + ((size_t)b * H + y0) * W * Cout + cob + co;
      for (int p0 = pl; p0 < npx; p0 += UNR * (CONV_NT / 64)) {
        float g[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int p = p0 + u * (CONV_NT / 64);
          g[u] = p < npx ? dyb[(size_t)p * Cout] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int p = min(p0 + u * (CONV_NT / 64), npx - 1);      // clamped: g is 0 for the overshoot
          const int r = p / W, c = p - r * W;
          const float* w0 = xs + r * RS + c * 3;          // window top-left = patch (r, c): 3 rows x 9 contiguous floats
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[ky * 9 + q] = fmaf(g[u], w0[ky * RS + q], acc[ky * 9 + q]);     // q = kx*3 + ci
        }
      }
    }
    // combine the 4 pixel lanes in fixed order, then store [tap][co][ci]
    __syncthreads();
    if (pl > 0) {
#pragma unroll
      for (int k = 0; k < 27; ++k) red[pl - 1][co][k] = acc[k];
    }
    __syncthreads();
    if (pl == 0 && cob + co < Cout) {
      float* out = slab + (size_t)blockIdx.x * 9 * Cout * 3;
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const float t = ((acc[k] + red[0][co][k]) + red[1][co][k]) + red[2][co][k];
        const int ky = k / 9, kx = (k % 9) / 3, ci = k % 3;
        out[((size_t)(ky * 3 + kx) * Cout + cob + co) * 3 + ci] = t;
      }
    }
  }
}
// ---- the same layer on the matrix cores (round 2): M = 32 output channels x 2 halves, N = (ky, kx, ci) = 27 of 32 columns, K = pixels.
// No im2col: lane n of the B operand always wants the SAME (ky, kx, ci) of the pixel's 3x3x3 window, i.e. a fixed offset into the
// raw 3-channel patch [3 rows][34 px x 3 ch] in LDS (+ 3 floats for the second pixel of the K pair); lanes 27..31 read a zero
// region.  The A operand (dY) is loaded straight from global memory in MFMA layout (lane = (channel, pixel of the pair): two 128-B
// segments per load).  One wave = one 32-pixel row segment (16 K steps x 2 MFMAs), 4 waves per workgroup summed through LDS, one
// 64 x 27 partial per workgroup; plenty of resident waves instead of software pipelining - the launch is bound by streaming dY
// (37.7 MB at B = 16 / 96 px) once.  Needs W % 32 == 0 and Cout == 64 (Discriminator.features[0] at 96 and 192 px); other shapes
// keep the VALU kernel above.
constexpr int C3M_ROW = 104;                 // floats per patch row in LDS (34 px x 3 ch = 102, padded)
constexpr int C3M_WAVE = 3 * C3M_ROW + 96;   // + the zero region the idle columns read (16 K steps x 6 floats)
constexpr int C3M_TPW = 4;                   // row segments a wave accumulates before the workgroup writes its partial
__global__ __launch_bounds__(CONV_NT) void wgrad_k3c3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ slab, int B, int H, int W, int ntiles) {
  __shared__ float patch[4 * C3M_WAVE];
  __shared__ float scr[4 * 32 * 33];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int Cout = 64;
  float* const ps = patch + wave * C3M_WAVE;
  const int tpr = W >> 5;                                 // tiles per image row
  // this lane's column n = li = ky*9 + kx*3 + ci  ->  fixed offset into the patch (+ 3 floats for the pair's second pixel)
  const int ky = li / 9, kr = li - ky * 9;
  const float* const bl = ps + (li < 27 ? ky * C3M_ROW + kr + 3 * lh : 3 * C3M_ROW);
  f32x16 acc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
  for (int i = lane; i < 96; i += 64) ps[3 * C3M_ROW + i] = 0.f;      // the zero region the idle columns read
  // a wave accumulates C3M_TPW row segments (tiles blockIdx.x * 4 * C3M_TPW + wave + 4 i) before the workgroup writes its partial:
  // one 64 x 27 partial per 4 * C3M_TPW tiles - the slab and its reduce (15.9 MB / 43 us with one partial per 4 tiles at B = 32)
  // shrink by C3M_TPW
  // Software pipeline over the wave's tiles: the dY fragments (32 registers) and the patch values (5) of tile it + 1 are loaded before the
  // 32 MFMAs of tile it are issued - with 2-3 waves per SIMD and load -> wait -> multiply in sequence the launch ran at 1.9 TB/s
  // (42.4 -> 36.6 us inside the step at B = 32).
  float av[2][16][2], xv[2][5];
  auto issue = [&](int it, float (&a_)[16][2], float (&x_)[5]) {
    const int t = (blockIdx.x * C3M_TPW + it) * 4 + wave;
    const bool live = it < C3M_TPW && t < ntiles;
    const int b = t / (H * tpr), rem = t - b * (H * tpr);
    const int y = rem / tpr, x0 = (rem - y * tpr) << 5;
    // dY in MFMA layout: K step ks, half h -> channel 32h + li of pixel x0 + 2ks + lh
    const float* dp = dy + (live ? (((size_t)b * H + y) * W + x0 + lh) * Cout + li : 0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      a_[ks][0] = live ? dp[(size_t)(2 * ks) * Cout] : 0.f;
      a_[ks][1] = live ? dp[(size_t)(2 * ks) * Cout + 32] : 0.f;
    }
    // X patch rows y-1 .. y+1, columns x0-1 .. x0+32 (as floats: (x0-1)*3 .. +102), zero outside the image
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = lane + 64 * u;
      const int r = i / 102, c = i - r * 102;
      const int iy = y - 1 + r, fx = (x0 - 1) * 3 + c;
      const bool ok = live && i < 306 && (unsigned)iy < (unsigned)H && (unsigned)fx < (unsigned)(W * 3);
      x_[u] = ok ? x[((size_t)b * H + iy) * W * 3 + fx] : 0.f;
    }
  };
  issue(0, av[0], xv[0]);
#pragma unroll
  for (int it = 0; it < C3M_TPW; ++it) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = lane + 64 * u;
      if (i < 306) ps[(i / 102) * C3M_ROW + i % 102] = xv[it & 1][u];      // (the previous tile's LDS reads were issued before: in order)
    }
    if (it + 1 < C3M_TPW) issue(it + 1, av[(it + 1) & 1], xv[(it + 1) & 1]);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float bv = bl[6 * ks];                        // LDS ops of a wave complete in order: the stores above have landed
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[it & 1][ks][0], bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[it & 1][ks][1], bv, acc[1], 0, 0, 0);
    }
  }   // tiles of this wave
  // 4 waves -> one 64 x 27 partial (fixed order), slab [chunk][tap][Cout][3]
  float* const out = slab + (size_t)blockIdx.x * 9 * Cout * 3;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[h][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * wave + 4 * lh + j;
      const float v = ((scr[row * 33 + li] + scr[(32 + row) * 33 + li]) + scr[(64 + row) * 33 + li]) + scr[(96 + row) * 33 + li];
      if (li < 27) out[((size_t)(li / 3) * Cout + 32 * h + row) * 3 + li % 3] = v;
    }
  }
}
// dW[co][ci][tap] (+)= sum_chunk slab[chunk][tap][co][ci] for the kernel above (64 x 27 = 1,728 outputs, 288-576 chunks): a workgroup
// takes 16 outputs x 64 chunk groups (64-B runs per chunk row; 108 workgroups instead of the 27 of the 64-output form, whose 16 groups
// walked 36 dependent loads each: 26.6 us for 4 MB at B = 32 inside the step, 22.2 now), <= 9 loads per thread in two batches, combined in
// fixed order through LDS.
constexpr int C3R_OUT = 16, C3R_GRP = 1024 / C3R_OUT;
__global__ __launch_bounds__(1024) void c3m_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nchunk,
                                                          int accumulate) {
  __shared__ float part[C3R_GRP][C3R_OUT + 1];
  constexpr int TOTAL = 9 * 64 * 3;
  const int o = threadIdx.x & (C3R_OUT - 1), gq = threadIdx.x / C3R_OUT;
  const int i = blockIdx.x * C3R_OUT + o;
  float t = 0.f;
  int c = gq;
  for (; c + 7 * C3R_GRP < nchunk; c += 8 * C3R_GRP) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(c + C3R_GRP * u) * TOTAL + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c + C3R_GRP * u < nchunk ? slab[(size_t)(c + C3R_GRP * u) * TOTAL + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  part[gq][o] = t;
  __syncthreads();
  if (threadIdx.x < 64) {                                  // 4 lanes per output, 16 groups each, then two shuffles: fixed order
    const int oo = threadIdx.x & (C3R_OUT - 1), q = threadIdx.x / C3R_OUT;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < C3R_GRP / 4; ++k) t += part[q * (C3R_GRP / 4) + k][oo];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (q == 0) {
      const int ii = blockIdx.x * C3R_OUT + oo;
      const int tap = ii / 192, oc = ii - tap * 192;      // slab order [tap][co][ci] -> dW[co][ci][tap]
      float* d = dw + (size_t)oc * 9 + tap;
      *d = accumulate ? *d + t : t;
    }
  }
}
inline bool k3c3_mfma_applies(int W, int Cout) { return (W & 31) == 0 && Cout == 64 && !sst_env("SST_WGRAD_NO_K3C3_MFMA"); }
inline int k3c3_mfma_chunks(int B, int H, int W) { return (B * H * (W >> 5) + 4 * C3M_TPW - 1) / (4 * C3M_TPW); }

inline bool k3c3_applies(int Cin, int ksize, int stride, const float* in_scale, int in_act) {
  return Cin == 3 && ksize == 3 && stride == 1 && !in_scale && in_act == ACT_NONE;
}

// Plan of the band variant: rows per band R, bands per chunk, number of chunks - or R = 0 when the shape is not covered.
struct WgBandPlan { int R, bpc, nchunk, nbands; size_t lds; };
inline WgBandPlan wgrad_band_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int njobs) {
  WgBandPlan best{0, 0, 0, 0, 0};
  if (ksize != 3 || stride != 1 || (Cin & 63) || (Cout & 63) || (W & 3) || W < 4) return best;
  if (const char* e = sst_env("SST_WGRAD_BAND")) {
    if (atoi(e) == 0) return best;
  }
  // small single layers stay with the per-tap kernel: a band workgroup's fixed cost (9 slab tiles) needs several bands of
  // MFMA work behind it (measured: equal at 2.7 GFLOP, 1.3x faster at 10.9 GFLOP, 2.3x slower at 0.68 GFLOP)
  if (2.0 * B * H * W * Cin * Cout * 9 * njobs < 2.5e9 && !sst_env("SST_WGRAD_BAND")) return best;
  const int nblk = (Cout >> 6) * (Cin >> 6);
  long slots = 512;                        // 2 resident workgroups per CU
  int rmax = 2;
  if (const char* e = sst_env("SST_WGRAD_BAND_SLOTS")) slots = atoi(e);      // dev overrides
  if (const char* e = sst_env("SST_WGRAD_BAND_R")) rmax = atoi(e);
  double best_cost = 1e30;
  for (int R = rmax; R >= 1; --R) {
    if (H % R) continue;
    const int npx_x = (R + 2) * (W + 2), npx_d = R * W;
    if ((npx_x * 16 + CONV_NT - 1) / CONV_NT > WB_XS || (npx_d * 16 + CONV_NT - 1) / CONV_NT > WB_DS) continue;
    const size_t lds = (size_t)(npx_x + npx_d) * WLD * sizeof(float);
    if (lds > 78 * 1024) continue;
    const long nbands = (long)B * (H / R);
    const long per = (long)nblk * njobs;
    long bpc = (nbands * per + slots - 1) / slots;
    if (bpc < 1) bpc = 1;
    const long nchunk = (nbands + bpc - 1) / bpc;
    const long rounds = (nchunk * per + slots - 1) / slots;
    // time ~ rounds * (bands of MFMA work per workgroup + fixed cost of the 9 slab tiles ~ 0.7 band of R = 2)
    const double cost = (double)rounds * ((double)bpc * R + 1.4);
    if (cost < best_cost) {
      best_cost = cost;
      best = WgBandPlan{R, (int)bpc, (int)nchunk, (int)nbands, lds};
    }
  }
  return best;
}

// dW[co][ci][tap] (+)= sum_chunk slab[chunk][tap][co][ci]     (fixed chunk order: reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nchunk,
                                                           int KK, int Cout, int Cin, int accumulate, int grouped, WgJobTab tab) {
  if (grouped) {                       // grouped: blockIdx.y = layer
    slab = tab.j[blockIdx.y].slab;
    dw = tab.j[blockIdx.y].dw;
  }
  const int64_t per_tap = (int64_t)Cout * Cin, total = per_tap * KK;
  const bool split = accumulate & 2;      // flag bit 1: split variant
  accumulate &= 1;
  if (split) {
    // split variant (many chunks): 64 float4 items per workgroup x 4 chunk groups; group cg sums chunks cg, cg+4, ...
    // (fixed order), the 4 partial sums are combined in order through LDS - same result on every run
    __shared__ f32x4 part[4][64];
    const int item = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t i4 = blockIdx.x * 64ll + item, total4 = total >> 2;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (i4 < total4) {
#pragma unroll 4
      for (int s = cg; s < nchunk; s += 4) t += *reinterpret_cast<const f32x4*>(slab + (size_t)s * total + (i4 << 2));
    }
    part[cg][item] = t;
    __syncthreads();
    if (cg == 0 && i4 < total4) {
      t = part[0][item] + part[1][item];
      t += part[2][item];
      t += part[3][item];
      const int64_t i = i4 << 2;
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = dw + (oc + j) * KK + tap;
        *d = accumulate ? *d + t[j] : t[j];
      }
    }
    return;
  }
  if ((Cin & 3) == 0) {
    const int64_t total4 = total >> 2;
    for (int64_t i4 = blockIdx.x * 256ll + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
      const int64_t i = i4 << 2;
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int s = 0; s < nchunk; ++s) t += *reinterpret_cast<const f32x4*>(slab + (size_t)s * total + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = dw + (oc + j) * KK + tap;
        *d = accumulate ? *d + t[j] : t[j];
      }
    }
  } else {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
      float t = 0.f;
#pragma unroll 8
      for (int s = 0; s < nchunk; ++s) t += slab[(size_t)s * total + i];
      float* d = dw + oc * KK + tap;
      *d = accumulate ? *d + t : t;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// Number of pixel chunks the wgrad kernel will use for this shape (slab floats = chunks*KS*KS*Cout*Cin).
SST_API int sst_conv_wgrad_chunks(int B, int Ho, int Wo, int Cin, int Cout, int ksize) {
  const int64_t M = (int64_t)B * Ho * Wo;
  const int nblk = ((Cout + 63) / 64) * ((Cin + 63) / 64);
  // Measured on MI355X (tools/ablate_wgrad.py): ~4 co-resident workgroups per CU hide the staging latency best, as long
  // as a workgroup still gets >= ~160 pixels of K (shorter chunks are all fixed cost + slab traffic).
  int64_t want = (1024 + ksize * ksize * nblk - 1) / (ksize * ksize * nblk);
  int64_t maxc = (M + 159) / 160;
  if (want > maxc) want = maxc;
  if (want < 1) want = 1;
  int64_t chunk_px = ((M + want - 1) / want + SUB - 1) / SUB * SUB;
  return (int)((M + chunk_px - 1) / chunk_px);
}

// slab reduce launch: the split variant (flag bit 1 of `accumulate`) when there are many chunks to sum
static void launch_wgrad_reduce(const float* slab, float* dw, int nchunk, int KK, int Cout, int Cin, int accumulate,
                                const WgJobTab* tab, int njobs, hipStream_t st) {
  static const WgJobTab no_tab{};
  const WgJobTab& t = tab ? *tab : no_tab;
  const int grouped = tab != nullptr;
  const int64_t total = (int64_t)KK * Cout * Cin;
  if ((Cin & 3) == 0 && nchunk >= 8) {
    const int blocks = (int)((total / 4 + 63) / 64);
    wgrad_reduce_kernel<<<dim3(blocks, njobs), 256, 0, st>>>(slab, dw, nchunk, KK, Cout, Cin, (accumulate & 1) | 2, grouped, t);
    return;
  }
  const int64_t items = (Cin & 3) == 0 ? total / 4 : total;
  const int rb = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
  wgrad_reduce_kernel<<<dim3(rb, njobs), 256, 0, st>>>(slab, dw, nchunk, KK, Cout, Cin, accumulate, grouped, t);
}

// ---- slab reduces of several layers (any shapes) in ONE launch: the weight gradients of a backward pass are leaves of its chain, so
// their reduces can wait for the end of the pass (sst_conv_wgrad_grp with accumulate bit 2 leaves the slab alone) and go out together -
// one launch that fills the chip instead of 5-10 us of a few workgroups behind every layer's weight gradient.  Per job the arithmetic
// is launch_wgrad_reduce's (same variant by the same rule, same chunk order): bit-identical to the per-layer reduce.
namespace {
struct WrJob { const float* slab; float* dw; int nchunk, KK, Cout, Cin, accumulate, blocks; };
constexpr int WR_TAB_MAX = 24;
struct WrJobTab { WrJob j[WR_TAB_MAX]; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WrJobTab tab) {
  const WrJob jb = tab.j[blockIdx.y];
  if ((int)blockIdx.x >= jb.blocks) return;
  const float* __restrict__ slab = jb.slab;
  float* __restrict__ dw = jb.dw;
  const int nchunk = jb.nchunk, KK = jb.KK, accumulate = jb.accumulate & 1;
  const int64_t per_tap = (int64_t)jb.Cout * jb.Cin, total = per_tap * KK;
  if ((jb.Cin & 3) == 0 && nchunk >= 8) {                  // split variant (wgrad_reduce_kernel)
    __shared__ f32x4 part[4][64];
    const int item = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t i4 = blockIdx.x * 64ll + item, total4 = total >> 2;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (i4 < total4) {
#pragma unroll 4
      for (int s = cg; s < nchunk; s += 4) t += *reinterpret_cast<const f32x4*>(slab + (size_t)s * total + (i4 << 2));
    }
    part[cg][item] = t;
    __syncthreads();
    if (cg == 0 && i4 < total4) {
      t = ((part[0][item] + part[1][item]) + part[2][item]) + part[3][item];
      const int64_t i = i4 << 2;
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = dw + (oc + j) * KK + tap;
        *d = accumulate ? *d + t[j] : t[j];
      }
    }
    return;
  }
  if ((jb.Cin & 3) == 0) {
    const int64_t total4 = total >> 2;
    for (int64_t i4 = blockIdx.x * 256ll + threadIdx.x; i4 < total4; i4 += (int64_t)jb.blocks * 256) {
      const int64_t i = i4 << 2;
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int s = 0; s < nchunk; ++s) t += *reinterpret_cast<const f32x4*>(slab + (size_t)s * total + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = dw + (oc + j) * KK + tap;
        *d = accumulate ? *d + t[j] : t[j];
      }
    }
  } else {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)jb.blocks * 256) {
      const int tap = (int)(i / per_tap);
      const int64_t oc = i - (int64_t)tap * per_tap;
      float t = 0.f;
#pragma unroll 8
      for (int s = 0; s < nchunk; ++s) t += slab[(size_t)s * total + i];
      float* d = dw + oc * KK + tap;
      *d = accumulate ? *d + t : t;
    }
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 / STRIDE-2 / pad-1 weight gradient, ALL 9 TAPS per wave (discriminator layers 2 / 4 / 6 / 8, reference model.py:34-59).
//   dW[co][ci][ky][kx] = sum_{b,oy,ox} dY[b,oy,ox,co] * act(X)[b, 2oy+ky-1, 2ox+kx-1, ci]
// GEMM view per tap: M = co, N = ci, K = output pixels.  The per-tap kernel (conv_wgrad_kernel) re-stages X for each of the 9
// taps and ran these layers at 30 % of the MFMA peak; the stride-1 all-taps form (conv_wgrad_band_kernel: a 64 x 64 block x 9
// taps per WORKGROUP, K split over workgroups) cannot take them either - a stride-2 X patch is 4.8x the dY tile, a band of whole
// rows does not fit the LDS at 96 / 48 px, and 512 workgroups x 147 KB of partials are 75 MB of slab per launch whatever the layer.
// Here the split is turned round:
//   * a WORKGROUP owns ONE 32 (co) x 32 (ci) block for all 9 taps and a contiguous range of pixel tiles; its 8 WAVES split K:
//     wave w takes tiles w, w+8, ... of the range, each with 9 accumulators (144 registers) = the whole block;
//   * a tile is TH x TW output pixels (2x8 or 2x6 - never across an image), its X patch (2TH+1) x (2TW+1) pixels x 32
//     channels and its dY tile live in a WAVE-PRIVATE LDS region: no workgroup barrier in the main loop.  The next tile's global
//     loads are issued before the current tile's MFMA loop and consumed (BatchNorm affine + LeakyReLU of the producer applied,
//     padding zeroed) behind it; the MFMA loop is fully unrolled, every LDS offset an immediate, the operands of K step ks+1 are
//     read before the 9 MFMAs of step ks are issued;
//   * the waves' partial blocks are summed through LDS at the end (tap by tap, double-buffered: one barrier per tap), so a
//     workgroup emits ONE 32 x 32 x 9 partial: (Cout/32)(Cin/32) x nchunk x 36.9 KB of slab per launch - 9.4 MB with ~256
//     workgroups instead of 75 MB, and no K split over workgroups at all for the 512-channel layer.  wgrad_reduce_kernel sums the
//     chunks in fixed order.
// Measured (tools/time_wgrad_s2.py, B = 16, incl. the slab reduce): 96 px 46 us (per-tap kernel 51), 48 px 43 (54), 24 px 44 (57),
// 12 px 52 (66).  Where a launch goes (SST_WGRAD_S2_DBG ablation, 48-px layer): launch + slab reduce 6, first loads 3.5, MFMA loop
// 21 (= the MFMA rate), staging 4.5, exchange 5-7 - strictly additive: the two waves of a SIMD do not hide each other's phases
// (de-phasing them with a head start changed nothing), 4 waves x 4x8 tiles with 512 registers each measured 49-65 us, fewer VALU
// instructions in the staging (interior / edge instantiations, LeakyReLU as max) and LDS operand prefetch changed nothing.
struct WgS2Args {
  const float* x; const float* dy; float* slab;
  const float* in_scale; const float* in_shift; const float* in_slope; float in_slope_const; int in_act;
  int B, H, W, Cin, Cout, Ho, Wo;
  int tiles_x, tiles_img, ntiles, tpc;          // tiles per row of tiles / per image / in total / per chunk
  int dbg;                                      // ablation bits (SST_WGRAD_S2_DBG, dev): 1 no staging stores, 2 no global loads, 4 no MFMA loop, 8 no exchange
  float* dw; int accumulate;                    // one chunk only (no K split over workgroups): dW[co][ci][tap] written (or added to) directly, no slab
  int gB;                                       // images per coefficient group (in_scale / in_shift are [B / gB][Cin]: passes batched as one tensor); 0 = one group
};

// S2_NW waves per workgroup: 8 = two per SIMD, one stages while the other multiplies (256 registers each); 4 = one per SIMD with
// 512 registers (room for the scheduler to read LDS operands ahead)
template <int S, int TH, int TW, int S2_NW>
struct WgS2Geom {
  static constexpr int NPIX = TH * TW, PH = S * (TH - 1) + 3, PW = S * (TW - 1) + 3, NPX = PH * PW;
  static constexpr int XQ = (NPX * 8 + 63) / 64;           // 16-B quads per lane: X patch (8 quads = 32 channels per pixel)
  static constexpr int DQ = (NPIX * 8 + 63) / 64;          //                       dY tile
  static constexpr int WAVE_FLOATS = (XQ * 8 + DQ * 8) * 32;    // one wave's private region: [patch pixel slot][32 ci] then [tile pixel slot][32 co]
  static constexpr size_t LDS_BYTES = (size_t)S2_NW * WAVE_FLOATS * sizeof(float);
  static_assert(NPIX % 2 == 0 && TW % 2 == 0, "pixel pairs of one tile row");
  static_assert((size_t)2 * S2_NW * 32 * 33 * sizeof(float) <= LDS_BYTES, "exchange scratch (double-buffered) fits the staging area");
};

template <int S, int TH, int TW, int S2_NW>
__global__ __launch_bounds__(S2_NW * 64, 1) void conv_wgrad_tile_kernel(WgS2Args a) {
  using G = WgS2Geom<S, TH, TW, S2_NW>;
  constexpr int NPIX = G::NPIX, PW = G::PW, NPX = G::NPX, XQ = G::XQ, DQ = G::DQ;
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  float* const xs = wlds + wave * G::WAVE_FLOATS;        // X patch
  float* const ds = xs + XQ * 8 * 32;                    // dY tile
  const int nci = a.Cin >> 5;
  const int co0 = (blockIdx.y / nci) * 32, ci0 = (blockIdx.y % nci) * 32;
  const int t0 = blockIdx.x * a.tpc, t1 = min(a.ntiles, t0 + a.tpc);
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const int q4 = (lane & 7) * 4;                         // this lane's channel quad inside the 32-channel slice

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Staging comes in two instantiations chosen per tile (wave-uniform): INTERIOR tiles (no padding pixel in the patch: 4 of 5) need
  // no address selects and no zeroing; EDGE tiles (first tile row / column of an image) mask the top row / left column.  VALU work
  // matters here beyond its own time: on this chip it does not run under another wave's MFMAs (the SQ counters of every conv
  // kernel show VALU-active + MFMA-busy < 100 %, and this kernel's phases measured strictly additive).
  f32x4 xr[XQ], dr[DQ];
  unsigned okm = 0;
  bool edge = false;                                     // of the tile whose loads are in xr
  int goff = 0;                                          // ... and the float offset of its coefficient group in in_scale / in_shift
  const bool lmax = a.in_act == ACT_SLOPE && slope >= 0.f && slope <= 1.f;       // LeakyReLU as max(v, slope * v)
  auto issue = [&](int t, auto edge_) {
    constexpr bool EDGE = decltype(edge_)::value;
    const int b = t / a.tiles_img, rem = t - b * a.tiles_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    // origin of the patch (EDGE: may lie one row / one column outside the image: those quads are masked), < 2^31 floats (host check)
    const int xo = ((b * a.H + S * oy0 - 1) * a.W + S * ox0 - 1) * a.Cin + ci0 + q4;
    const bool last_y = oy0 + TH == a.Ho, last_x = ox0 + TW == a.Wo;
    const float* db = a.dy + ((size_t)(b * a.Ho + oy0) * a.Wo + ox0) * a.Cout + co0 + q4;
    okm = 0;
#pragma unroll
    for (int u = 0; u < XQ; ++u) {
      const int p = (lane >> 3) + 8 * u;
      const int pr = p / PW, pc = p - pr * PW;
      bool ok = 8 * u + 7 < NPX || p < NPX;              // slots past the patch exist in the last quad only (compile-time for the others)
      if (EDGE) ok = ok && (pr > 0 || oy0 > 0) && (pc > 0 || ox0 > 0);           // stride 2: bottom / right never leave the image (H, W even)
      if (EDGE && S == 1) ok = ok && (pr < G::PH - 1 || !last_y) && (pc < PW - 1 || !last_x);
      const int off = xo + (pr * a.W + pc) * a.Cin;
      xr[u] = *reinterpret_cast<const f32x4*>(a.x + ((EDGE || 8 * u + 7 >= NPX) ? (ok ? off : ci0 + q4) : off));
      if (EDGE) okm |= ok ? (1u << u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < DQ; ++u) {
      const int p = (lane >> 3) + 8 * u;
      const int py = p / TW, px = p - py * TW;
      dr[u] = *reinterpret_cast<const f32x4*>(db + ((8 * u + 7 < NPIX || p < NPIX) ? (py * a.Wo + px) * a.Cout : 0));
    }
  };
  auto is_edge = [&](int t) {
    const int rem = t % a.tiles_img, tx = rem % a.tiles_x;
    bool e = rem < a.tiles_x || tx == 0;
    if (S == 1) e = e || rem >= a.tiles_img - a.tiles_x || tx == a.tiles_x - 1;
    return e;
  };
  auto issue_any = [&](int t) {
    goff = a.gB ? (t / a.tiles_img / a.gB) * a.Cin : 0;
    edge = is_edge(t);
    if (edge) issue(t, std::true_type{}); else issue(t, std::false_type{});
  };
  // registers -> this wave's LDS region.  The patch region has XQ * 8 >= NPX pixel slots and the dY region DQ * 8 >= NPIX, so
  // every lane stores unconditionally
  auto store = [&](auto edge_) {
    constexpr bool EDGE = decltype(edge_)::value;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};      // re-read per tile (cache hits): 8 registers not held across the K loop
    if (a.in_scale) {
      sc = *reinterpret_cast<const f32x4*>(a.in_scale + goff + ci0 + q4);
      sh = *reinterpret_cast<const f32x4*>(a.in_shift + goff + ci0 + q4);
    }
#pragma unroll
    for (int u = 0; u < XQ; ++u) {
      const int p = (lane >> 3) + 8 * u;
      f32x4 v = xr[u];
      if (a.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
      }
      if (lmax) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], v[j] * slope);
      } else if (a.in_act == ACT_SLOPE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
      }
      if (EDGE && !((okm >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // padding stays exactly zero
      *reinterpret_cast<f32x4*>(xs + p * 32 + q4) = v;
    }
#pragma unroll
    for (int u = 0; u < DQ; ++u) *reinterpret_cast<f32x4*>(ds + ((lane >> 3) + 8 * u) * 32 + q4) = dr[u];
  };

  int t = t0 + wave;
  if (t < t1 && !(a.dbg & 2)) issue_any(t);
  // per-lane fragment bases: lane (li = m or n, lh = k): dY pixel 2*ks + lh, X patch pixel const(ks, tap) + S*lh
  const float* const dl = ds + lh * 32 + li;
  const float* const xl = xs + S * lh * 32 + li;
  for (; t < t1; t += S2_NW) {
    // (the previous tile's MFMA loop has issued all its LDS reads: LDS ops of a wave complete in order)
    if (!(a.dbg & 1)) {
      if (edge) store(std::true_type{}); else store(std::false_type{});
    }
    if (t + S2_NW < t1 && !(a.dbg & 2)) issue_any(t + S2_NW);            // in flight during the MFMA loop below
    if (a.dbg & 4) continue;
    // ---- K loop: pixel pairs (2ks, 2ks+1) of the tile (TW even: same tile row), 9 taps each
    // Operands of step ks+1 are read from LDS BEFORE the 9 MFMAs of step ks are issued (scheduling barriers keep that order: left
    // alone, hipcc reads each operand 1-3 MFMAs ahead of its use and waits lgkmcnt(0) every other MFMA - measured 42 us per layer)
    float av, bv[9];
    auto fetch = [&](int ks, float& a_, float (&b_)[9]) {
      const int py = (2 * ks) / TW, px = (2 * ks) % TW;
      a_ = dl[2 * ks * 32];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) b_[tap] = xl[((S * py + tap / 3) * PW + S * px + tap % 3) * 32];
    };
    fetch(0, av, bv);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NPIX / 2; ++ks) {
      float an = 0.f, bn[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (ks + 1 < NPIX / 2) fetch(ks + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[tap], acc[tap], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      av = an;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) bv[tap] = bn[tap];
    }
  }

  // ---- the waves' K-partials -> one 32 x 32 block per tap (through LDS, fixed order); wave w then owns rows 4w .. 4w+3
  if (a.dbg & 8) {
    if (acc[0][0] == 12345.f) a.slab[0] = 1.f;
    return;
  }
  // [2][wave][row 32][33] at the start of the staging area, double-buffered: round r writes buffer r & 1, so ONE barrier per round
  // is enough (a wave reaches the barrier of round r + 1 only after its reads of round r, and buffer r & 1 is next written in round r + 2)
  float* const scr = wlds;
  float* const out = a.slab + (size_t)blockIdx.x * 9 * a.Cout * a.Cin;
  __syncthreads();                                       // every wave has left its main loop (the scratch overlays the staging area)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float* const sb = scr + (tap & 1) * (S2_NW * 32 * 33);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      sb[(wave * 32 + row) * 33 + li] = acc[tap][r];
    }
    __syncthreads();
    constexpr int RPW = 32 / S2_NW;                      // rows of the block per wave
#pragma unroll
    for (int j = 0; j < RPW / 2; ++j) {
      const int row = RPW * wave + (RPW / 2) * lh + j;
      float v = sb[row * 33 + li];
#pragma unroll
      for (int w = 1; w < S2_NW; ++w) v += sb[(w * 32 + row) * 33 + li];
      if (a.dw) {
        float* d = a.dw + ((size_t)(co0 + row) * a.Cin + ci0 + li) * 9 + tap;
        *d = a.accumulate ? *d + v : v;
      } else {
        out[((size_t)tap * a.Cout + co0 + row) * a.Cin + ci0 + li] = v;
      }
    }
  }
}

// Plan of the stride-2 all-taps kernel: tile shape (th = 0: shape not taken) and number of pixel chunks.
struct WgS2Plan { int th, tw, nw, nchunk, tpc, ntiles; };
inline WgS2Plan wgrad_s2_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  WgS2Plan pl{0, 0, 0, 0, 0, 0};
  if (ksize != 3 || (stride != 1 && stride != 2) || (Cin & 31) || (Cout & 31) || B <= 0) return pl;
  if (stride == 2 && ((H | W) & 1)) return pl;
  bool force = false;                                   // SST_WGRAD_S2 / SST_WGRAD_S1T: 0 = off, 1 = also below the work threshold (tests)
  if (const char* e = sst_env(stride == 2 ? "SST_WGRAD_S2" : "SST_WGRAD_S1T")) {
    if (atoi(e) == 0) return pl;
    force = true;
  }
  const int Ho = H / stride, Wo = W / stride;
  constexpr int NW = 8;
  int th = 0, tw = 0;
  if (stride == 2) {
    if (Wo % 8 == 0 && Ho % 2 == 0) { th = 2; tw = 8; }
    else if (Wo % 6 == 0 && Ho % 2 == 0) { th = 2; tw = 6; }
  } else {
    // stride 1 (same kernel, 4-row tiles: a 6 x 10 patch per 32 pixels): single layers with enough work only - the trunk's grouped
    // launches and small layers stay where they are (conv_wgrad_band_kernel / per-tap kernel)
    if (2.0 * B * H * W * Cin * Cout * 9 < 2.5e9 && !force) return pl;
    if (Wo % 8 == 0 && Ho % 4 == 0) { th = 4; tw = 8; }
    else if (Wo % 6 == 0 && Ho % 4 == 0) { th = 4; tw = 6; }
  }
  if (!th || (int64_t)B * H * W * Cin >= (1ll << 31)) return pl;
  const int ntiles = B * (Ho / th) * (Wo / tw);
  const int nblk = (Cout >> 5) * (Cin >> 5);
  // ~one workgroup (8 waves) per CU; a chunk is a multiple of 8 tiles (the waves split it), never less than 8
  int nchunk = 256 / nblk;
  if (nchunk < 1) nchunk = 1;
  int tpc = ((ntiles + nchunk - 1) / nchunk + NW - 1) / NW * NW;
  if (tpc < NW) tpc = NW;
  nchunk = (ntiles + tpc - 1) / tpc;
  pl = WgS2Plan{th, tw, NW, nchunk, tpc, ntiles};
  return pl;
}

template <int S, int TH, int TW, int S2_NW>
static int launch_wgrad_s2_t(const WgS2Args& a, const WgS2Plan& pl, hipStream_t st) {
  using G = WgS2Geom<S, TH, TW, S2_NW>;
  static bool lds_ok = false;
  if (!lds_ok) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tile_kernel<S, TH, TW, S2_NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess)
      return sst_set_error(SST_ERR_HIP, "conv_wgrad_tile: cannot raise the LDS limit");
    lds_ok = true;
  }
  conv_wgrad_tile_kernel<S, TH, TW, S2_NW><<<dim3(pl.nchunk, (a.Cout >> 5) * (a.Cin >> 5)), S2_NW * 64, G::LDS_BYTES, st>>>(a);
  return SST_OK;
}

static long g_wgrad_band_launches = 0;
SST_API long sst_debug_wgrad_band_launches(void) { return g_wgrad_band_launches; }   // test hook

static int launch_wgrad_band(WgradArgs& a, const WgBandPlan& pl, int njobs, hipStream_t st, const WgJobTab& tab) {
  ++g_wgrad_band_launches;
  static bool big_lds_enabled = false;
  if (!big_lds_enabled) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_band_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_band_kernel<WB_XS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return sst_set_error(SST_ERR_HIP, "conv_wgrad_band: cannot raise the LDS limit");
    big_lds_enabled = true;
  }
  a.R = pl.R; a.bpc = pl.bpc; a.nbands = pl.nbands; a.nchunk = pl.nchunk;
  dim3 grid((unsigned)pl.nchunk * njobs, (a.Cout >> 6) * (a.Cin >> 6));
  const int xslots = ((pl.R + 2) * (a.W + 2) * 16 + CONV_NT - 1) / CONV_NT;
  if (xslots <= 7)
    conv_wgrad_band_kernel<7><<<grid, CONV_NT, pl.lds, st>>>(a, tab);
  else
    conv_wgrad_band_kernel<WB_XS><<<grid, CONV_NT, pl.lds, st>>>(a, tab);
  SST_LAUNCH_CHECK("conv_wgrad_band_kernel");
  return SST_OK;
}

// Chunk count (slab floats per layer = chunks*k*k*Cout*Cin) that sst_conv_wgrad (njobs = 1) / sst_conv_wgrad_grouped use
// for this shape; H, W are the INPUT size.
SST_API int sst_conv_wgrad_chunks2(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int njobs) {
  if (njobs == 1) {
    const WgS2Plan p2 = wgrad_s2_plan(B, H, W, Cin, Cout, ksize, stride);
    if (p2.th) return p2.nchunk;
  }
  const WgBandPlan pl = wgrad_band_plan(B, H, W, Cin, Cout, ksize, stride, njobs);
  if (pl.R) return pl.nchunk;
  const int pad = ksize / 2;
  if (Cin == 3 && ksize == 3 && stride == 1) {         // 3-channel-input kernels (one chunk per workgroup) or the general one
    int a = B * ((H + C3_ROWS - 1) / C3_ROWS);
    const int g = sst_conv_wgrad_chunks(B, H, W, Cin, Cout, ksize);
    if (k3c3_mfma_applies(W, Cout) && k3c3_mfma_chunks(B, H, W) > a) a = k3c3_mfma_chunks(B, H, W);
    return a > g ? a : g;
  }
  return sst_conv_wgrad_chunks(B, (H + 2 * pad - ksize) / stride + 1, (W + 2 * pad - ksize) / stride + 1, Cin, Cout, ksize);
}

// Name of the main kernel sst_conv_wgrad (njobs = 1) / sst_conv_wgrad_grouped launch for this shape (rocprofv3 spelling).
SST_API const char* sst_conv_wgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int njobs) {
  if (njobs == 1) {
    const WgS2Plan p2 = wgrad_s2_plan(B, H, W, Cin, Cout, ksize, stride);
    if (p2.th) {
      static thread_local char nm[56];
      snprintf(nm, sizeof(nm), "conv_wgrad_tile_kernel<%d, %d, %d, %d>", stride, p2.th, p2.tw, p2.nw);
      return nm;
    }
  }
  const WgBandPlan pl = wgrad_band_plan(B, H, W, Cin, Cout, ksize, stride, njobs);
  if (pl.R) return ((pl.R + 2) * (W + 2) * 16 + CONV_NT - 1) / CONV_NT <= 7 ? "conv_wgrad_band_kernel<7>" : "conv_wgrad_band_kernel<10>";
  if (Cin == 3 && ksize == 3 && stride == 1) return k3c3_mfma_applies(W, Cout) ? "wgrad_k3c3_mfma_kernel" : "wgrad_k3c3_kernel";
  return ((Cin & 3) == 0 && (Cout & 3) == 0) ? "conv_wgrad_kernel<true>" : "conv_wgrad_kernel<false>";
}

// Does sst_conv_wgrad_grp take this shape with coefficient groups of grp_images images (the all-taps tile kernel does; a layer whose
// input has no BatchNorm affine needs no groups at all)?
SST_API int sst_conv_wgrad_groups_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images) {
  return grp_images > 0 && B % grp_images == 0 && wgrad_s2_plan(B, H, W, Cin, Cout, ksize, stride).th != 0;
}

SST_API int sst_conv_wgrad_grp(const float* x, const float* dy, float* slab, float* dw, const float* in_scale,
                               const float* in_shift, const float* in_slope, float in_slope_const, int in_act, int B, int H,
                               int W, int Cin, int Cout, int stride, int ksize, int accumulate, int grp_images, void* stream);
SST_API int sst_conv_wgrad(const float* x, const float* dy, float* slab, float* dw, const float* in_scale,
                           const float* in_shift, const float* in_slope, float in_slope_const, int in_act, int B, int H,
                           int W, int Cin, int Cout, int stride, int ksize, int accumulate, void* stream) {
  return sst_conv_wgrad_grp(x, dy, slab, dw, in_scale, in_shift, in_slope, in_slope_const, in_act, B, H, W, Cin, Cout, stride, ksize,
                            accumulate, 0, stream);
}

// grp_images > 0: in_scale / in_shift are [B / grp_images][Cin] - images b*grp_images .. use row b (several passes of the network
// batched as one tensor, each with the BatchNorm coefficients of its own batch statistics; dW sums over all of them).
SST_API int sst_conv_wgrad_grp(const float* x, const float* dy, float* slab, float* dw, const float* in_scale,
                               const float* in_shift, const float* in_slope, float in_slope_const, int in_act, int B, int H,
                               int W, int Cin, int Cout, int stride, int ksize, int accumulate, int grp_images, void* stream) {
  SST_REQUIRE(x && dy && slab && dw, "sst_conv_wgrad: null pointer");
  const bool grouped = grp_images > 0 && grp_images < B && in_scale;
  SST_REQUIRE(!grouped || sst_conv_wgrad_groups_ok(B, H, W, Cin, Cout, ksize, stride, grp_images),
              "sst_conv_wgrad: coefficient groups are only taken by the all-taps tile kernel (B=%d H=%d W=%d Cin=%d Cout=%d stride=%d)", B, H, W,
              Cin, Cout, stride);
  SST_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) && (ksize & 1) && ksize <= 9,
              "sst_conv_wgrad: bad shape");
  WgradArgs a;
  a.grouped = 0; a.nchunk = 0;
  static const WgJobTab no_tab{};
  a.x = x; a.dy = dy; a.slab = slab; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.in_slope_const = in_slope_const; a.in_act = in_act;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.KS = ksize; a.pad = ksize / 2;
  a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
  a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
  const int64_t M = (int64_t)B * a.Ho * a.Wo;
  SST_REQUIRE(M < (1ll << 31), "sst_conv_wgrad: too many pixels");
  int nchunk = sst_conv_wgrad_chunks(B, a.Ho, a.Wo, Cin, Cout, ksize);
  a.chunk_px = (int)(((M + nchunk - 1) / nchunk + SUB - 1) / SUB * SUB);
  {
    const char* e = sst_env("SST_WGRAD_DBG");
    a.dbg = e ? atoi(e) : 0;
  }
  const int KK = ksize * ksize;
  dim3 grid(nchunk, KK, ((Cout + 63) / 64) * ((Cin + 63) / 64));
  const WgS2Plan p2 = wgrad_s2_plan(B, H, W, Cin, Cout, ksize, stride);
  const WgBandPlan pl = p2.th ? WgBandPlan{0, 0, 0, 0, 0} : wgrad_band_plan(B, H, W, Cin, Cout, ksize, stride, 1);
  if (pl.R) {
    const int rc = launch_wgrad_band(a, pl, 1, sst_stream(stream), no_tab);
    if (rc != SST_OK) return rc;
    nchunk = pl.nchunk;
  } else if (p2.th) {
    WgS2Args s2;
    s2.x = x; s2.dy = dy; s2.slab = slab; s2.in_scale = in_scale; s2.in_shift = in_shift; s2.in_slope = in_slope;
    s2.in_slope_const = in_slope_const; s2.in_act = in_act;
    s2.B = B; s2.H = H; s2.W = W; s2.Cin = Cin; s2.Cout = Cout; s2.Ho = a.Ho; s2.Wo = a.Wo;
    s2.dbg = sst_env("SST_WGRAD_S2_DBG") ? atoi(sst_env("SST_WGRAD_S2_DBG")) : 0;
    const bool direct = p2.nchunk == 1 && !sst_env("SST_WGRAD_TILE_NO_DIRECT");
    s2.dw = direct ? dw : nullptr;
    s2.accumulate = accumulate & 1;
    s2.gB = grouped ? grp_images : 0;
    s2.tiles_x = a.Wo / p2.tw; s2.tiles_img = (a.Ho / p2.th) * s2.tiles_x; s2.ntiles = p2.ntiles; s2.tpc = p2.tpc;
    int rc;
    if (stride == 2) rc = p2.tw == 8 ? launch_wgrad_s2_t<2, 2, 8, 8>(s2, p2, sst_stream(stream)) : launch_wgrad_s2_t<2, 2, 6, 8>(s2, p2, sst_stream(stream));
    else rc = p2.tw == 8 ? launch_wgrad_s2_t<1, 4, 8, 8>(s2, p2, sst_stream(stream)) : launch_wgrad_s2_t<1, 4, 6, 8>(s2, p2, sst_stream(stream));
    if (rc != SST_OK) return rc;
    if (direct) {
      SST_LAUNCH_CHECK("conv_wgrad_tile_kernel");
      return SST_OK;                            // the single chunk went straight into dW
    }
    nchunk = p2.nchunk;
  } else if (k3c3_applies(Cin, ksize, stride, in_scale, in_act) && k3c3_mfma_applies(W, Cout)) {
    nchunk = k3c3_mfma_chunks(B, H, W);
    wgrad_k3c3_mfma_kernel<<<nchunk, CONV_NT, 0, sst_stream(stream)>>>(x, dy, slab, B, H, W, B * H * (W >> 5));
    SST_LAUNCH_CHECK("wgrad_k3c3_mfma_kernel");
    c3m_reduce_kernel<<<9 * 64 * 3 / C3R_OUT, 1024, 0, sst_stream(stream)>>>(slab, dw, nchunk, accumulate);
    SST_LAUNCH_CHECK("c3m_reduce_kernel");
    return SST_OK;
  } else if (k3c3_applies(Cin, ksize, stride, in_scale, in_act) && !sst_env("SST_WGRAD_NO_K3C3")) {
    nchunk = B * ((H + C3_ROWS - 1) / C3_ROWS);
    const size_t lds = (size_t)(C3_ROWS + 2) * ((W + 2) * 3 + 3) * sizeof(float);
    SST_REQUIRE(lds <= 48 * 1024, "sst_conv_wgrad: image too wide for the 3-channel-input kernel (W=%d)", W);
    wgrad_k3c3_kernel<<<dim3(nchunk, 1), CONV_NT, lds, sst_stream(stream)>>>(x, dy, slab, B, H, W, Cout);
  } else if ((Cin & 3) == 0 && (Cout & 3) == 0)
    conv_wgrad_kernel<true><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, no_tab);
  else
    conv_wgrad_kernel<false><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, no_tab);
  SST_LAUNCH_CHECK("conv_wgrad_kernel");
  if (accumulate & 4) return SST_OK;             // the caller reduces the slab later (sst_wgrad_reduce_multi)
  launch_wgrad_reduce(slab, dw, nchunk, KK, Cout, Cin, accumulate, nullptr, 1, sst_stream(stream));
  SST_LAUNCH_CHECK("wgrad_reduce_kernel");
  return SST_OK;
}

// Chunks of slab that sst_conv_wgrad_grp leaves for sst_wgrad_reduce_multi when called with accumulate bit 2 (value 4) for this shape;
// 0: the launch writes dW itself (single-chunk tile kernel, 3-channel MFMA kernel with its own reduce) and the bit changes nothing.
SST_API int sst_conv_wgrad_pending_reduce(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int has_in_scale, int in_act) {
  const WgS2Plan p2 = wgrad_s2_plan(B, H, W, Cin, Cout, ksize, stride);
  if (p2.th) return (p2.nchunk == 1 && !sst_env("SST_WGRAD_TILE_NO_DIRECT")) ? 0 : p2.nchunk;
  const WgBandPlan pl = wgrad_band_plan(B, H, W, Cin, Cout, ksize, stride, 1);
  if (pl.R) return pl.nchunk;
  static const float one = 0.f;
  const bool c3 = k3c3_applies(Cin, ksize, stride, has_in_scale ? &one : nullptr, in_act);
  if (c3 && k3c3_mfma_applies(W, Cout)) return 0;
  if (c3 && !sst_env("SST_WGRAD_NO_K3C3")) return B * ((H + C3_ROWS - 1) / C3_ROWS);
  const int pad = ksize / 2;
  return sst_conv_wgrad_chunks(B, (H + 2 * pad - ksize) / stride + 1, (W + 2 * pad - ksize) / stride + 1, Cin, Cout, ksize);
}

// jobs: HOST array of njobs <= 24 records {slab, dw, nchunk, k*k, Cout, Cin, accumulate, 0} (two pointers + six ints = 40 bytes):
// dW[co][ci][tap] (+)= sum over the job's nchunk slab chunks, every job in one launch.
SST_API int sst_wgrad_reduce_multi(const void* jobs, int njobs, void* stream) {
  static_assert(sizeof(WrJob) == 40, "WrJob layout");
  SST_REQUIRE(jobs && njobs > 0 && njobs <= WR_TAB_MAX, "sst_wgrad_reduce_multi: 1..%d jobs", WR_TAB_MAX);
  WrJobTab tab{};
  memcpy(tab.j, jobs, (size_t)njobs * sizeof(WrJob));
  int maxb = 0;
  for (int i = 0; i < njobs; ++i) {
    WrJob& j = tab.j[i];
    SST_REQUIRE(j.slab && j.dw && j.nchunk > 0 && j.KK > 0 && j.Cout > 0 && j.Cin > 0, "sst_wgrad_reduce_multi: bad job %d", i);
    const int64_t total = (int64_t)j.KK * j.Cout * j.Cin;
    if ((j.Cin & 3) == 0 && j.nchunk >= 8) {
      j.blocks = (int)((total / 4 + 63) / 64);
    } else {
      const int64_t items = (j.Cin & 3) == 0 ? total / 4 : total;
      j.blocks = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
    }
    if (j.blocks > maxb) maxb = j.blocks;
  }
  wgrad_reduce_multi_kernel<<<dim3(maxb, njobs), 256, 0, sst_stream(stream)>>>(tab);
  SST_LAUNCH_CHECK("wgrad_reduce_multi_kernel");
  return SST_OK;
}

// Weight gradients of `njobs` layers of IDENTICAL shape in one launch (+ one grouped slab reduce).
// jobs: HOST array of WgJob {x, dy, slab, dw, in_scale, in_shift, in_slope, in_slope_const, in_act} (64 bytes each, device
// pointers inside); it travels to the kernels as a by-value argument, at most WG_TAB_MAX = 40 jobs per call.
// every job's slab holds sst_conv_wgrad_chunks(...) * k*k*Cout*Cin floats.
SST_API int sst_conv_wgrad_grouped(const void* jobs, int njobs, int B, int H, int W, int Cin, int Cout, int stride, int ksize,
                                   int accumulate, void* stream) {
  static_assert(sizeof(WgJob) == 64, "WgJob layout");
  SST_REQUIRE(jobs && njobs > 0 && njobs <= WG_TAB_MAX && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) &&
                  (ksize & 1) && ksize <= 9, "sst_conv_wgrad_grouped: bad argument (at most %d jobs per call)", WG_TAB_MAX);
  WgJobTab tab{};
  memcpy(tab.j, jobs, (size_t)njobs * sizeof(WgJob));
  WgradArgs a;
  a.grouped = 1;
  a.x = a.dy = nullptr; a.slab = nullptr; a.in_scale = a.in_shift = a.in_slope = nullptr; a.in_slope_const = 0.f; a.in_act = 0;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.KS = ksize; a.pad = ksize / 2;
  a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
  a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
  const int64_t M = (int64_t)B * a.Ho * a.Wo;
  SST_REQUIRE(M < (1ll << 31), "sst_conv_wgrad_grouped: too many pixels");
  // With many layers in one grid the chip is full without fine pixel chunks: aim at ~target workgroups in total
  // (fewer, longer chunks = less slab traffic).  Never more chunks than the per-layer slab was sized for.
  int nchunk = sst_conv_wgrad_chunks(B, a.Ho, a.Wo, Cin, Cout, ksize);
  {
    int target = 2048;
    if (const char* e = sst_env("SST_WGRAD_GROUP_WGS")) target = atoi(e);
    const int nblk = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    int want = (target + ksize * ksize * nblk * njobs - 1) / (ksize * ksize * nblk * njobs);
    if (want < 1) want = 1;
    if (target > 0 && want < nchunk) {
      const int64_t cpx = ((M + want - 1) / want + SUB - 1) / SUB * SUB;
      nchunk = (int)((M + cpx - 1) / cpx);
    }
  }
  a.nchunk = nchunk;
  a.chunk_px = (int)(((M + nchunk - 1) / nchunk + SUB - 1) / SUB * SUB);
  a.dbg = 0;
  const int KK = ksize * ksize;
  dim3 grid((unsigned)nchunk * njobs, KK, ((Cout + 63) / 64) * ((Cin + 63) / 64));
  const WgBandPlan pl = wgrad_band_plan(B, H, W, Cin, Cout, ksize, stride, njobs);
  if (pl.R) {
    const int rc = launch_wgrad_band(a, pl, njobs, sst_stream(stream), tab);
    if (rc != SST_OK) return rc;
    nchunk = pl.nchunk;
  } else if ((Cin & 3) == 0 && (Cout & 3) == 0)
    conv_wgrad_kernel<true><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, tab);
  else
    conv_wgrad_kernel<false><<<grid, CONV_NT, 0, sst_stream(stream)>>>(a, tab);
  SST_LAUNCH_CHECK("conv_wgrad_kernel (grouped)");
  launch_wgrad_reduce(nullptr, nullptr, nchunk, KK, Cout, Cin, accumulate, &tab, njobs, sst_stream(stream));
  SST_LAUNCH_CHECK("wgrad_reduce_kernel (grouped)");
  return SST_OK;
}

// 3x3 convolution (pad 1, stride 1 or 2), NHWC, fp32 MFMA implicit GEMM - the "N-split" form of the pipelined kernel (conv_pipe.hip) for
// layers with Cout % 128 == 0 and enough tiles (reference model.py:30-59: the discriminator's 128..512-channel layers and their
// stride-1 data-gradients).
//
// conv_pipe_kernel's 4 waves split K (16 input channels each): every wave stages a quarter of the patch for 72 MFMAs, and the 4
// partial 32 x 32 accumulators meet through LDS (two barriers, 16 + 16 LDS accesses per lane) before a wave-local epilogue on 8 of
// the 32 channels - per 64-channel block of ONE 32 px x 32 ch tile.  The ablation build (profiles/r03_ablate_conv_pipe.txt) puts
// staging stores at 12-17 %, that exchange + epilogue at 8-21 % and the patch loads at 17-22 % of a launch, all of it exposed.
// Here the 4 waves split N instead: they share ONE patch (all 64 channels of the block, staged cooperatively once) and each computes
// the full K for its own 32 output channels of the same 32-pixel tile - 288 MFMAs per wave per staged block instead of 72, no
// K-partial exchange at all (the accumulator IS the result: lane = output channel, 16 pixels per lane), statistics and backward
// partials as 16 in-register adds + one cross-half shuffle per lane.  Weight traffic per MFMA is unchanged (1 KiB fragment per 4 MFMAs,
// same packed layout), patch traffic per MFMA is a quarter.
// Everything else is conv_pipe.hip's: tall-image tiles (TW x 32/TW pixels, images separated by one shared zero row in the LDS
// patch), persistent workgroups walking units q = blockIdx.x + i * gridDim.x (unit = tile x group of 4 channel blocks, all input
// blocks), next patch in registers + 9-deep weight-fragment ring under the MFMAs with hand-counted vmcnt waits, per-pass BatchNorm
// coefficient groups.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

int sst_conv_band_rows(int B, int H, int W, int Cin, int Cout, int ksize, int stride);   // conv_band.hip

namespace {

constexpr int NS_RING = 9;
constexpr int NS_PSTR = CB + 4;      // floats per patch pixel in LDS (64 channels + 4 pad: 16-B aligned rows, conflict-free b128 reads)

constexpr int OUT_NHWC = 0, OUT_SHUFFLE = 1;             // the two stores of conv_epilogue.h's list this kernel has

struct NsArgs {
  const float* x; const float* wp; float* y; const float* bias;
  const float* in_scale; const float* in_shift; const float* in_slope; float in_slope_const; int in_act;
  float* stats; float* stats_cnt;                      // [n_mt][2][Cout], [n_mt] or null
  const float* epi_y; const float* epi_scale; const float* epi_shift; const float* epi_slope;
  float epi_slope_const; int epi_act; float* epi_partial;   // [n_mt][3][Cout] or null
  int B, H, W, Cin, Cout, Ho, Wo, R;
  int tiles_x, n_mt, nfg, ncb, units;
  int gB, grows;
  int shuffle;                                          // OUT_SHUFFLE store (PixelShuffle(2) of the result, conv_common.h)
};

template <int S, int TW>
struct NsGeom {
  static constexpr int TH = 32 / TW;
  static constexpr int NB = TW == 2 ? 3 : 1;                 // image boundaries a tile may cross
  static constexpr int PW = (TW - 1) * S + 3;
  static constexpr int PR = (TH - 1) * S + 3 + NB;
  static constexpr int NP = PW * PR;
  static constexpr int NUQ = (NP + 15) / 16;                 // patch quads per thread (16 threads x 16 B = one pixel's 64 channels)
  static constexpr int NSL = NUQ + 2;                        // vector loads of one stage's staging (quads + the two coefficient loads)
};

#define NS_GLOAD(dst, voff, sbase) \
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase))
#define NS_WAIT(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n))

struct NsStage {
  int b0, rem0, ix0, c0, gin;
  const float* w;              // this wave's weight block: [72 chunks][256 floats] (+ lane * 4 floats per lane)
};

template <int S, int TW>
__global__ __launch_bounds__(CONV_NT, 2) void conv_ns_kernel(NsArgs a) {
  using G = NsGeom<S, TW>;
  constexpr int TH = G::TH, NB = G::NB, PW = G::PW, NP = G::NP, NUQ = G::NUQ, NSL = G::NSL, PS = NS_PSTR;
  __shared__ __attribute__((aligned(16))) float patch[NP * PS];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int nwg = gridDim.x;
  const int HV = a.H + 1;
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const int cq4 = (tid & 15) * 4;                      // this thread's channel quad inside the 64-channel block
  const int p0 = tid >> 4;                             // its first patch pixel

  int q = blockIdx.x;
  if (q >= a.units) return;
  int mt, gq, r0, ox0, oy0, b0, cb;
  auto decode = [&](int qq, int& mt_, int& g_, int& r0_, int& ox0_, int& b0_, int& oy0_) {
    mt_ = qq % a.n_mt;
    g_ = qq / a.n_mt;
    const int ty = mt_ / a.tiles_x, tx = mt_ - ty * a.tiles_x;
    r0_ = ty * TH;
    ox0_ = tx * TW;
    b0_ = r0_ / a.Ho;
    oy0_ = r0_ - b0_ * a.Ho;
  };
  auto make_stage = [&](int b0_, int oy0_, int ox0_, int g_, int cb_) {
    NsStage s;
    s.b0 = b0_;
    s.rem0 = oy0_ * S;
    s.ix0 = ox0_ * S - 1;
    s.c0 = cb_ * CB;
    s.gin = a.gB ? (b0_ / a.gB) * a.Cin : 0;
    s.w = a.wp + ((size_t)((g_ * 4 + wave) * a.ncb + cb_) * 72) * 256;
    return s;
  };
  decode(q, mt, gq, r0, ox0, b0, oy0);
  cb = 0;
  NsStage cur = make_stage(b0, oy0, ox0, gq, 0), nxt;

  f32x4 sv[NUQ];
  unsigned okmask = 0;
  f32x4 ssc, ssh;
  const float* sc_base = a.in_scale ? a.in_scale : a.x;
  const float* sh_base = a.in_scale ? a.in_shift : a.x;
  unsigned boffs[NUQ];
  auto stage_load = [&](const NsStage& s, bool same_tile) {
    if (same_tile) {
#pragma unroll
      for (int u = 0; u < NUQ; ++u) {
        boffs[u] += CB * 4u;
        NS_GLOAD(sv[u], boffs[u], a.x);
      }
    } else {
      okmask = 0;
#pragma unroll
      for (int u = 0; u < NUQ; ++u) {
        const int p = p0 + 16 * u;
        const int pr = p / PW, pc = p - pr * PW;
        int rr = s.rem0 + pr, b = s.b0;
#pragma unroll
        for (int k = 0; k <= NB; ++k) {
          const bool wrap = rr >= HV;
          rr -= wrap ? HV : 0;
          b += wrap ? 1 : 0;
        }
        const int iy = rr - 1, ix = s.ix0 + pc;            // virtual row 0 of an image is its (shared) zero row
        const bool ok = p < NP && rr >= 1 && b < a.B && (unsigned)ix < (unsigned)a.W;
        const int off = ((b * a.H + iy) * a.W + ix) * a.Cin + s.c0 + cq4;        // < 2^29 floats (host check)
        const unsigned boff = (ok ? (unsigned)off : (unsigned)cq4) * 4u;
        boffs[u] = boff;
        NS_GLOAD(sv[u], boff, a.x);
        okmask |= ok ? (1u << u) : 0u;
      }
    }
    const unsigned coff = (unsigned)(a.in_scale ? s.gin + s.c0 + cq4 : cq4) * 4u;
    NS_GLOAD(ssc, coff, sc_base);
    NS_GLOAD(ssh, coff, sh_base);
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int u = 0; u < NUQ; ++u) {
      const int p = p0 + 16 * u;
      if (p >= NP) continue;
      f32x4 t = sv[u];
      if (a.in_scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaf(t[j], ssc[j], ssh[j]);
      }
      if (a.in_act == ACT_SLOPE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : t[j] * slope;
      }
      if (!((okmask >> u) & 1u)) t = f32x4{0.f, 0.f, 0.f, 0.f};            // zero padding stays exactly zero
      *reinterpret_cast<f32x4*>(&patch[p * PS + cq4]) = t;
    }
  };
  auto lane_base = [&](int oy0_) {
    const int tr = li / TW, tc = li - tr * TW;
    int cross = 0;
#pragma unroll
    for (int k = 1; k <= NB; ++k) cross += (oy0_ + tr >= k * a.Ho) ? 1 : 0;
    return ((S * tr + cross) * PW + tc * S) * PS + 4 * lh;
  };
  int a_base = lane_base(oy0);

  f32x4 ring[NS_RING];
  const unsigned wlane = lane * 16u;
#define NS_WCHUNK(dst, w, i) NS_GLOAD(dst, wlane, (w) + (i) * 256)

#pragma unroll
  for (int i = 0; i < NS_RING; ++i) NS_WCHUNK(ring[i], cur.w, i);
  stage_load(cur, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  stage_store();
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (;;) {
    const bool unit_end = cb + 1 == a.ncb;
    int n_q = q, n_mt = mt, n_g = gq, n_r0 = r0, n_ox0 = ox0, n_oy0 = oy0, n_b0 = b0, n_cb = cb + 1;
    bool more = true;
    if (unit_end) {
      n_q = q + nwg;
      more = n_q < a.units;
      n_cb = 0;
      if (more) {
        decode(n_q, n_mt, n_g, n_r0, n_ox0, n_b0, n_oy0);
        nxt = make_stage(n_b0, n_oy0, n_ox0, n_g, 0);
      } else {
        nxt = cur;                        // harmless re-loads of the last stage (never stored)
      }
    } else {
      nxt = cur;
      nxt.c0 += CB;
      nxt.w += (size_t)72 * 256;
    }

    // ---- K loop of the current stage: 72 chunks (9 taps x 8 channel octets) x 4 MFMAs.  Loads in issue order: chunk 0's refill, the
    // NSL staging loads of the next stage, then one refill per chunk.  When chunk c waits for ring slot c % 9 (loaded 9 chunks ago):
    //   c = 0       : 8 refills are younger                                   -> vmcnt(8)
    //   c = 1 .. 9  : the staging loads sit among the younger ones             -> vmcnt(8 + NSL)
    //   c >= 10     : again 8 refills, and the staging loads have landed too   -> vmcnt(8)
    {
      const float* ab = patch + a_base;
      f32x4 av = *reinterpret_cast<const f32x4*>(ab);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 72; ++c) {
        f32x4 an = av;
        if (c + 1 < 72) {
          const int t = (c + 1) >> 3, ks = (c + 1) & 7;
          an = *reinterpret_cast<const f32x4*>(ab + ((t / 3) * PW + (t % 3)) * PS + ks * 8);
        }
        if (c == 0 || c >= 10) NS_WAIT(ring[c % NS_RING], 8); else NS_WAIT(ring[c % NS_RING], 8 + NSL);
        const f32x4 bv = ring[c % NS_RING];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
        if (c + NS_RING < 72) NS_WCHUNK(ring[c % NS_RING], cur.w, c + NS_RING);
        else NS_WCHUNK(ring[c % NS_RING], nxt.w, c + NS_RING - 72);
        if (c == 0) stage_load(nxt, !unit_end);
        av = an;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (unit_end) {
      // ---- epilogue, in registers: acc[r] = pixel (r & 3) + 8 (r >> 2) + 4 lh of the tile, output channel nf * 32 + li
      const int co = (gq * 4 + wave) * 32 + li;
      const float bv = a.bias ? a.bias[co] : 0.f;
      const int nvalid = min(TH, a.R - r0) * TW;
      float v[16];
      int off[16];                         // < 2^31 floats (host check)
      bool ok[16];
      float s1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pi = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int rr = r0 + pi / TW, ox = ox0 + pi % TW;
        ok[r] = rr < a.R;
        // OUT_SHUFFLE: output channel co = 4c + 2i + j lands at row 2 rr + i (tall image: 2 (b Ho + oy) + i), column 2 ox + j, channel c
        off[r] = a.shuffle ? ((2 * rr + ((co >> 1) & 1)) * (2 * a.Wo) + 2 * ox + (co & 1)) * (a.Cout >> 2) + (co >> 2)
                           : (rr * a.Wo + ox) * a.Cout + co;
        v[r] = acc[r] + bv;
        acc[r] = 0.f;
        if (ok[r]) a.y[off[r]] = v[r];
        s1 += ok[r] ? v[r] : 0.f;
      }
      if (a.stats) {
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / (float)nvalid;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = ok[r] ? v[r] - mean : 0.f;
          m2 = fmaf(d, d, m2);
        }
        m2 += __shfl_xor(m2, 32, 64);
        if (lh == 0) {
          float* st = a.stats + (size_t)mt * 2 * a.Cout;
          st[co] = s1;
          st[a.Cout + co] = m2;
          if (gq == 0 && wave == 0 && li == 0) a.stats_cnt[mt] = (float)nvalid;
        }
      }
      if (a.epi_partial) {
        const float eslope = a.epi_slope ? a.epi_slope[0] : a.epi_slope_const;
        float es = 1.f, eh = 0.f;
        if (a.epi_scale) {
          const int ge = a.gB ? (r0 / a.grows) * a.Cout : 0;
          es = a.epi_scale[ge + co];
          eh = a.epi_shift[ge + co];
        }
        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float yv = ok[r] ? a.epi_y[off[r]] : 0.f;
          const float g = ok[r] ? v[r] : 0.f;
          const float z = a.epi_scale ? fmaf(yv, es, eh) : yv;
          float gz = g;
          if (a.epi_act) {
            q2 = fmaf(g, fminf(z, 0.f), q2);
            gz = z > 0.f ? g : g * eslope;
          }
          q0 += gz;
          q1 = fmaf(gz, yv, q1);
        }
        q0 += __shfl_xor(q0, 32, 64);
        q1 += __shfl_xor(q1, 32, 64);
        q2 += __shfl_xor(q2, 32, 64);
        if (lh == 0) {
          float* ep = a.epi_partial + (size_t)mt * 3 * a.Cout;
          ep[co] = q0;
          ep[a.Cout + co] = q1;
          ep[2 * a.Cout + co] = q2;
        }
      }
      if (!more) return;
      a_base = lane_base(n_oy0);
    }
    __syncthreads();                       // every wave has finished reading the patch
    stage_store();                         // the next stage's block (its loads landed before chunk 10)
    __syncthreads();
    q = n_q; mt = n_mt; gq = n_g; r0 = n_r0; ox0 = n_ox0; oy0 = n_oy0; b0 = n_b0; cb = n_cb;
    cur = nxt;
  }
}

struct NsPlan { int tw, n_mt, tiles_x, nfg, ncb; };

NsPlan ns_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int out_mode = OUT_NHWC) {
  NsPlan pl{};
  if (ksize != 3 || (stride != 1 && stride != 2) || (Cin % 64) || (Cout % 128) || B <= 0) return pl;
  if (stride == 2 && ((H | W) & 1)) return pl;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  int tw = 0;
  if (Wo % 8 == 0 && Ho >= 3) tw = 8;
  else if (Wo % 4 == 0 && Ho >= 7) tw = 4;
  else if (Wo % 2 == 0 && Ho >= 5) tw = 2;
  if (!tw) return pl;
  const int th = 32 / tw;
  pl.tiles_x = Wo / tw;
  pl.n_mt = pl.tiles_x * ((B * Ho + th - 1) / th);
  pl.nfg = Cout / 128;
  pl.ncb = Cin / 64;
  // measured against conv_pipe_kernel (tools/time_pipe.py): 2,304 / 1,152 units 93.7 vs 117.5 us, 51.0 vs 61.3, 91.1 vs 101.4 (116-119 TF/s
  // = 74-76 % of the fp32 MFMA peak); 576 units 103.5 vs 92.5, 104.7 vs 95.3, 55.7 vs 53.6: a unit is 4x longer here, with fewer than
  // ~4 per CU the tail costs more than the exchange it saves - those layers stay on the K-split kernel
  // (the pixel-shuffle store has no K-split kernel to fall back to: the general 32 x 32-tile kernel runs those layers at 42-55 %)
  if (out_mode != OUT_NHWC && out_mode != OUT_SHUFFLE) return pl;
  if ((long)pl.n_mt * pl.nfg < (out_mode == OUT_SHUFFLE ? 512 : 1024)) return pl;
  pl.tw = tw;
  return pl;
}

}  // namespace

SST_API int sst_conv_ns_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int out_mode) {
  return ns_plan(B, H, W, Cin, Cout, ksize, stride, out_mode).tw;
}

// Arguments as sst_conv_pipe_fwd_grp (no split-K workspace); stats / stats_cnt / epi_partial have sst_conv_pipe_stat_tiles rows (the
// same tall-image tiling).
SST_API int sst_conv_ns_fwd(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                            const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                            float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                            const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W,
                            int Cin, int Cout, int ksize, int stride, int out_mode, int grp_images, void* stream) {
  SST_REQUIRE(x && wp && y, "sst_conv_ns_fwd: null pointer");
  const NsPlan pl = ns_plan(B, H, W, Cin, Cout, ksize, stride, out_mode);
  SST_REQUIRE(out_mode == OUT_NHWC || (!stats && !epi_partial), "sst_conv_ns_fwd: statistics / backward partials need the NHWC store");
  SST_REQUIRE(pl.tw, "sst_conv_ns_fwd: shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d stride=%d is not taken by the N-split kernel", B, H, W, Cin,
              Cout, ksize, stride);
  SST_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sst_conv_ns_fwd: in_scale/in_shift must come together");
  SST_REQUIRE(!stats || stats_cnt, "sst_conv_ns_fwd: stats needs stats_cnt");
  SST_REQUIRE(!epi_partial || (epi_y && !stats && ((epi_scale == nullptr) == (epi_shift == nullptr))),
              "sst_conv_ns_fwd: backward partials need epi_y and exclude forward stats");
  SST_REQUIRE((int64_t)B * H * W * Cin < (1ll << 29), "sst_conv_ns_fwd: input too large for 32-bit byte offsets");
  SST_REQUIRE((int64_t)B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) * Cout < (1ll << 31), "sst_conv_ns_fwd: output too large for 32-bit offsets");
  const int Ho = (H - 1) / stride + 1;
  SST_REQUIRE(grp_images == 0 || grp_images == B || (B % grp_images == 0 && (grp_images * Ho) % (32 / pl.tw) == 0),
              "sst_conv_ns_fwd: coefficient groups of %d images do not end on tile boundaries", grp_images);
  NsArgs a;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.in_slope_const = in_slope_const; a.in_act = in_act; a.stats = stats; a.stats_cnt = stats_cnt;
  a.epi_y = epi_y; a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_slope = epi_slope; a.epi_slope_const = epi_slope_const;
  a.epi_act = epi_act; a.epi_partial = epi_partial;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Ho = Ho; a.Wo = (W - 1) / stride + 1; a.R = B * a.Ho;
  a.tiles_x = pl.tiles_x; a.n_mt = pl.n_mt; a.nfg = pl.nfg; a.ncb = pl.ncb; a.units = pl.n_mt * pl.nfg;
  a.gB = (grp_images > 0 && grp_images < B) ? grp_images : 0;
  a.grows = a.gB * a.Ho;
  a.shuffle = out_mode == OUT_SHUFFLE;
  const int wg_per_cu = stride == 1 ? 3 : 2;      // register-limited (149-161 / 196-212 VGPRs)
  const int grid = a.units < wg_per_cu * 256 ? a.units : wg_per_cu * 256;
  hipStream_t st = sst_stream(stream);
#define SST_NS_LAUNCH(S_, TW_) conv_ns_kernel<S_, TW_><<<grid, CONV_NT, 0, st>>>(a)
  if (stride == 1) {
    if (pl.tw == 8) SST_NS_LAUNCH(1, 8); else if (pl.tw == 4) SST_NS_LAUNCH(1, 4); else SST_NS_LAUNCH(1, 2);
  } else {
    if (pl.tw == 8) SST_NS_LAUNCH(2, 8); else if (pl.tw == 4) SST_NS_LAUNCH(2, 4); else SST_NS_LAUNCH(2, 2);
  }
#undef SST_NS_LAUNCH
  SST_LAUNCH_CHECK("conv_ns_kernel");
  return SST_OK;
}

// 3x3 convolution (pad 1, stride 1 or 2), NHWC, fp32 MFMA implicit GEMM - PERSISTENT, SOFTWARE-PIPELINED form for the
// discriminator's layers (reference model.py:30-59: 64..512 channels at 96..6 px) and their stride-1 data-gradients.
//
// Why a second general kernel (conv_fwd.hip stays for every shape this one does not take): on these shapes conv_fwd_kernel
// runs stage -> barrier -> K loop -> epilogue once per workgroup with nothing overlapped inside a workgroup, one workgroup
// per 32 px x 32 ch tile (4,608 dispatches for the 48-px layer) and an 8x4 pixel tile that wastes 25-44 % of every MFMA on
// the 12x12 / 6x6 layers; measured (tools/ablate_d.py) its K loop alone reaches 50-68 % of the fp32 MFMA peak and staging +
// epilogue add their full time on top.  Here:
//   * the batch is ONE TALL IMAGE of B*Ho rows: a tile is TW x (32/TW) pixels of it (TW = 8 / 4 / 2 chosen so that TW
//     divides Wo), tiles may straddle images.  The LDS patch holds rows of a virtual tall input in which consecutive images
//     are separated by ONE shared zero row (bottom pad of image b = top pad of image b+1); every lane adds its own
//     crossed-boundary count to its patch row, so no MFMA lane is spent on padding pixels;
//   * a fixed grid of workgroups (GPC per CU) walks its share of the work units (unit = tile x K-split part); while the MFMAs
//     of stage s run, the global loads of stage s+1's input patch are in flight in registers (written to LDS behind the
//     K loop) and the weight fragments come through a 9-deep register ring that is refilled one half stage ahead - the
//     K loop is fully unrolled (18 chunks per wave and 64-channel block: wave w owns k-steps {2w, 2w+1} of every tap), all
//     LDS / weight offsets are immediates;
//   * the 4 waves split K by input channel and every wave stages ITS OWN 16-channel slice of the patch into a wave-private LDS
//     region: no workgroup barrier inside a unit's stage loop, the waves drift freely and only meet at the end of a unit for
//     the K-partial reduction (two barriers); after it wave w owns channels 8w..8w+7 of the tile for all 32 pixels, so bias,
//     store, BatchNorm statistics and backward partials are wave-local (DPP row sums, no LDS, no further barrier);
//   * layers with few tiles split K over workgroups (fp32 partial slabs in a caller-provided workspace, summed in fixed order
//     by pipe_reduce_kernel, which then runs the same epilogue): every launch offers >= 4 equal units per CU.
// GEMM view, fragment layouts, packed weights (conv_common.h: packed_index) and the epilogue's arithmetic (bias, per-tile
// BatchNorm statistics (sum, centred M2), BatchNorm/activation backward partials) are those of conv_fwd.hip.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>
#include "conv_epilogue.h"

int sst_conv_band_rows(int B, int H, int W, int Cin, int Cout, int ksize, int stride);   // conv_band.hip: trunk shapes stay there

namespace {

constexpr int PIPE_RING = 9;   // weight-fragment ring (chunks): refilled half a stage ahead

struct PipeArgs {
  const float* x;           // [B,H,W,Cin]
  const float* wp;          // packed weights (forward or stride-1 data-gradient layout)
  float* y;                 // [B,Ho,Wo,Cout]
  const float* bias;
  const float* in_scale; const float* in_shift; const float* in_slope; float in_slope_const; int in_act;
  float* stats; float* stats_cnt;                      // [n_mt][2][Cout], [n_mt] or null
  const float* epi_y; const float* epi_scale; const float* epi_shift; const float* epi_slope;
  float epi_slope_const; int epi_act; float* epi_partial;   // [n_mt][3][Cout] or null
  float* ws;                // split-K slabs [ksplit][total_tiles][32 px][32 ch] (ksplit > 1)
  int B, H, W, Cin, Cout, Ho, Wo;
  int R;                    // B*Ho: rows of the tall output image
  int tiles_x, n_mt, nfc, total_tiles, ncb, ksplit, cb_per, units;
#ifdef SST_PIPE_ABLATE
  int dbg;                  // dev build only (tools/build_ablate.sh, tools/ablate_pipe.py): 1 no LDS staging writes, 2 no epilogue, 4 no patch loads,
                            // 8 weight refills from one cache-resident address, 16 no MFMAs
#endif
  int gB, grows;            // coefficient groups: images per group (0 = one group) and rows of the tall output image per group.  The
                            // per-channel arrays (in_scale / in_shift, epi_scale / epi_shift) are then [B / gB][channels]: every group
                            // of gB consecutive images carries the BatchNorm coefficients of its own pass (several discriminator
                            // passes - D(gt), D(sr) - batched as one tall image, train.py:155-158); tiles never straddle a group
};

// MODE 0: 3x3 conv (pad 1).  MODE 1: data-gradient of a 3x3 / stride-2 / pad-1 conv - the four parity classes of an 8x4 (TW x 32/TW)
// block of class pixels in one unit: they read the same (TW+1) x (TH+1) patch of dY (no left / top padding, the zero row of the
// virtual tall image sits BEHIND each image), the 9 (class, tap) pairs take the place of the 9 taps, one accumulator per class.
template <int S, int TW, int MODE>
struct PipeGeom {
  static constexpr int TH = 32 / TW;
  static constexpr int NB = TW == 2 ? 3 : 1;                 // image boundaries a tile may cross (host checks Ho against it)
  static constexpr int KSPAN = MODE ? 2 : 3;
  static constexpr int PW = (TW - 1) * S + KSPAN;
  static constexpr int PR = (TH - 1) * S + KSPAN + NB;
  static constexpr int NP = PW * PR;
  static constexpr int NU = (NP + 15) / 16;                  // patch pixels per lane (4 lanes x 16 B = one pixel's 16-channel slice)
  static constexpr int PS = 20;                              // floats per pixel in a wave's patch (16 channels + 4 pad: 16-B aligned rows)
  static constexpr int WAVE_FLOATS = NP * PS;                // one wave's private patch
  static constexpr int NACC = MODE ? 4 : 1;                  // accumulators (parity classes)
  static constexpr int NPAIR = MODE ? 2 : 1;                 // accumulators exchanged per round at the end of a unit
  static constexpr int SCRATCH = NPAIR * 4 * 32 * 33;        // K-partial exchange at the end of a unit
};
constexpr int S2D_CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};     // (class, tap) pair -> class / tap of the class (taps row-major, 1 + (cls & 1) wide)
constexpr int S2D_TAP[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};

// one stage = one 64-channel block of one tile's input patch (+ the matching weight block)
struct Stage {
  int b0, rem0, ix0, c0;       // image / virtual row inside it of patch row 0, first patch column, first channel
  int gin;                     // float offset of the tile's coefficient group in in_scale / in_shift
  const float* w;              // this wave's weight pointer for the block (wave-uniform: + wave*512 floats; lanes add lane*16 B)
};

// Vector-memory loads of the main loop are issued by hand and waited for by hand with COUNTED s_waitcnt vmcnt(N): hipcc's own
// waitcnt insertion falls back to vmcnt(0) around this loop's back edge (every weight refill became a full stall).  Loads
// return in order, so "at most N outstanding" proves that everything older than the N youngest loads has landed; stores the
// compiler issues in an epilogue only ever make such a wait longer, never too short.  The "+v" operand ties the consumer of a
// register to its wait.  __builtin_amdgcn_sched_barrier(0) after every chunk keeps the compiler from moving code across.
// (s_nop 4: the compiler may hand the base pointer over in SGPRs it has just restored with v_readlane - a VALU write of an SGPR
// needs 5 wait states before a vector-memory instruction reads it, and the hazard recognizer does not look inside inline asm.)
#define PIPE_GLOAD(dst, voff, sbase) \
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase))
#define PIPE_WAIT(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n))

// sum over the 32 lanes of a wave half (lanes with equal lane >> 5): DPP row rotations + one cross-row exchange
__device__ __forceinline__ float half_sum(float t) {
  auto ror = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  t += ror(t, std::integral_constant<int, 0x128>{});   // row_ror:8
  t += ror(t, std::integral_constant<int, 0x124>{});   // row_ror:4
  t += ror(t, std::integral_constant<int, 0x122>{});   // row_ror:2
  t += ror(t, std::integral_constant<int, 0x121>{});   // row_ror:1
  t += __shfl_xor(t, 16, 64);
  return t;
}

// Epilogue of one 32 px x 32 ch tile, wave-local: this lane holds v[j] = conv result of pixel px = lane & 31 of the tile,
// channels c0 .. c0+3 (c0 = nf*32 + 8*wave + 4*(lane >> 5)).  No barriers, no LDS.
template <int TW, int MODE = 0>
__device__ __forceinline__ void pipe_epilogue(const PipeArgs& a, float v[4], int r0, int ox0, int c0, int mt, int lane, bool first,
                                              int cls = 0) {
  constexpr int TH = 32 / TW;
  const int px = lane & 31;
  const int r = r0 + px / TW, ox = ox0 + px % TW;
  const bool pix_ok = r < a.R && ox < a.Wo;
  size_t obase;
  if (MODE) {        // stride-2 data-gradient: class (sy, sx) pixel (r, ox) of the tall class grid -> dX pixel (2r + sy, 2ox + sx)
    obase = ((size_t)(2 * r + (cls >> 1)) * (2 * a.Wo) + 2 * ox + (cls & 1)) * a.Cout + c0;
    if (pix_ok) *reinterpret_cast<f32x4*>(a.y + obase) = f32x4{v[0], v[1], v[2], v[3]};
  }
  if (!MODE && a.bias) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += bv[j];
  }
  if (!MODE) {
    obase = ((size_t)r * a.Wo + ox) * a.Cout + c0;
    if (pix_ok) *reinterpret_cast<f32x4*>(a.y + obase) = f32x4{v[0], v[1], v[2], v[3]};
  }
  if (!MODE && a.stats) {
    // per-tile (sum, centred M2) per output channel over the tile's valid pixels (combined by bn_finalize with Chan's formula)
    const int nvalid = min(TH, a.R - r0) * min(TW, a.Wo - ox0);
    f32x4 s1, m2;
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] = half_sum(pix_ok ? v[j] : 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mean = s1[j] / (float)nvalid;
      const float d = pix_ok ? (v[j] - mean) : 0.f;
      m2[j] = half_sum(d * d);
    }
    if (px == 0) {
      float* st = a.stats + (size_t)mt * 2 * a.Cout;
      *reinterpret_cast<f32x4*>(st + c0) = s1;
      *reinterpret_cast<f32x4*>(st + a.Cout + c0) = m2;
      if (first) a.stats_cnt[mt] = (float)nvalid;
    }
  }
  if (a.epi_partial) {
    // backward partials of the stored value g against the saved conv output epi_y (see conv_epilogue.h: Conv3Args)
    const float eslope = a.epi_slope ? a.epi_slope[0] : a.epi_slope_const;
    f32x4 yv = {0.f, 0.f, 0.f, 0.f}, es = {1.f, 1.f, 1.f, 1.f}, eh = {0.f, 0.f, 0.f, 0.f};
    if (pix_ok) yv = *reinterpret_cast<const f32x4*>(a.epi_y + obase);
    if (a.epi_scale) {
      const int ge = a.gB ? (r0 / a.grows) * a.Cout : 0;          // the tile's coefficient group (wave-uniform)
      es = *reinterpret_cast<const f32x4*>(a.epi_scale + ge + c0);
      eh = *reinterpret_cast<const f32x4*>(a.epi_shift + ge + c0);
    }
    f32x4 q[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = pix_ok ? v[j] : 0.f;
      const float z = a.epi_scale ? fmaf(yv[j], es[j], eh[j]) : yv[j];
      float gz = g, q2 = 0.f;
      if (a.epi_act) {
        q2 = g * fminf(z, 0.f);
        gz = z > 0.f ? g : g * eslope;
      }
      q[0][j] = half_sum(gz);
      q[1][j] = half_sum(gz * yv[j]);
      q[2][j] = half_sum(q2);
    }
    if (px == 0) {
      const size_t prow = MODE ? (size_t)mt * 4 + cls : (size_t)mt;     // MODE 1: one row of partials per (tile, parity class)
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<f32x4*>(a.epi_partial + (prow * 3 + k) * a.Cout + c0) = q[k];
    }
  }
}

// MS: units of several stages (cb_per > 1) - the staging addresses of a tile are kept in registers and moved from one channel block
// to the next; single-stage layers (64 input channels) get the instantiation without them (measured: the extra registers cost the
// 96-px stride-2 layer 4 us).
template <int S, int TW, int MODE, bool MS>
__global__ __launch_bounds__(CONV_NT, (S == 1 && MODE == 0 ? 3 : 2)) void conv_pipe_kernel(PipeArgs a) {
  using G = PipeGeom<S, TW, MODE>;
  constexpr int NACC = G::NACC, NPAIR = G::NPAIR;
  constexpr int PW = G::PW, NP = G::NP, NU = G::NU, NB = G::NB;
  constexpr int PS = G::PS;
  __shared__ __attribute__((aligned(16))) float patch[4 * G::WAVE_FLOATS];   // [wave][patch pixel][16 channels + pad]
  __shared__ __attribute__((aligned(16))) float scratch[G::SCRATCH];         // [wave][tile pixel 32][33]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int nwg = gridDim.x;
  const int HV = a.H + 1;                                 // rows of one image in the virtual tall input (its top zero row + H)
  const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
  const int c4 = wave * 16 + (lane & 3) * 4;              // this lane's channel quad inside a 64-channel block: its wave's slice
  const int p0 = lane >> 2;                               // its first patch pixel (4 lanes per pixel)
  float* const lds = patch + wave * G::WAVE_FLOATS;       // this wave's private patch

  // ---- unit / stage sequencing (all wave-uniform)
  int q = blockIdx.x;                                     // current unit
  if (q >= a.units) return;
  int mt, nf, r0, ox0, oy0, cb, cb_end;                   // of the CURRENT unit
  Stage cur, nxt;
  auto decode = [&](int qq, int& mt_, int& nf_, int& ksi_, int& r0_, int& ox0_, int& b0_, int& oy0_) {
    mt_ = qq % a.n_mt;
    const int rest = qq / a.n_mt;
    nf_ = rest % a.nfc;
    ksi_ = rest / a.nfc;
    const int ty = mt_ / a.tiles_x, tx = mt_ - ty * a.tiles_x;
    r0_ = ty * G::TH;
    ox0_ = tx * TW;
    b0_ = r0_ / a.Ho;
    oy0_ = r0_ - b0_ * a.Ho;
  };
  auto make_stage = [&](int b0_, int oy0_, int ox0_, int nf_, int cb_) {
    Stage s;
    s.b0 = b0_;
    s.rem0 = oy0_ * S;
    s.ix0 = ox0_ * S - (MODE ? 0 : 1);
    s.c0 = cb_ * CB;
    s.gin = a.gB ? (b0_ / a.gB) * a.Cin : 0;
    s.w = a.wp + ((size_t)(nf_ * a.ncb + cb_) * 9 * 8) * 256 + wave * 512;
    return s;
  };
  int ksi, b0;
  decode(q, mt, nf, ksi, r0, ox0, b0, oy0);
  cb = ksi * a.cb_per;
  cb_end = cb + a.cb_per;
  cur = make_stage(b0, oy0, ox0, nf, cb);

  // ---- staging: loads of one stage's patch into registers / transform + LDS write.  NS = loads per stage_load call.
  constexpr int NS = NU + 2;
  f32x4 sv[NU];
  unsigned okmask = 0;
  f32x4 ssc, ssh;
  const float* sc_base = a.in_scale ? a.in_scale : a.x;          // no affine: two dummy (never used) loads keep the load count fixed
  const float* sh_base = a.in_scale ? a.in_shift : a.x;
  // same_tile (wave-uniform): the stage is the next 64-channel block of the tile just loaded - every address moves by one block and
  // the padding mask stays; the per-quad address arithmetic (division, image-boundary wraps, bounds) is only redone for a new
  // tile.  VALU work is not hidden behind the MFMAs on this chip: per stage it was 25 instructions per quad.
  unsigned boffs[MS ? NU : 1];
  auto stage_load = [&](const Stage& s, bool same_tile) {
#ifdef SST_PIPE_ABLATE
    if (a.dbg & 4) return;
#endif
    if (MS && same_tile) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        boffs[u] += CB * 4u;                                       // masked quads keep reading (in bounds) near the tensor's origin
        PIPE_GLOAD(sv[u], boffs[u], a.x);
      }
    } else {
      okmask = 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int p = p0 + 16 * u;
        const int pr = p / PW, pc = p - pr * PW;
        int rr = s.rem0 + pr, b = s.b0;
#pragma unroll
        for (int k = 0; k <= NB; ++k) {
          const bool wrap = rr >= HV;
          rr -= wrap ? HV : 0;
          b += wrap ? 1 : 0;
        }
        const int iy = MODE ? rr : rr - 1, ix = s.ix0 + pc;      // MODE 0: virtual row 0 of an image is its (shared) zero row; MODE 1: row H is
        const bool ok = p < NP && (MODE ? rr < a.H : rr >= 1) && b < a.B && (unsigned)ix < (unsigned)a.W;
        const int off = ((b * a.H + iy) * a.W + ix) * a.Cin + s.c0 + c4;        // < 2^29 floats (host check)
        const unsigned boff = (ok ? (unsigned)off : (unsigned)c4) * 4u;
        if (MS) boffs[u] = boff;
        PIPE_GLOAD(sv[u], boff, a.x);
        okmask |= ok ? (1u << u) : 0u;
      }
    }
    const unsigned coff = (unsigned)(a.in_scale ? s.gin + s.c0 + c4 : c4) * 4u;
    PIPE_GLOAD(ssc, coff, sc_base);
    PIPE_GLOAD(ssh, coff, sh_base);
  };
  auto stage_store = [&]() {
#ifdef SST_PIPE_ABLATE
    if (a.dbg & 1) return;
#endif
#pragma unroll
    for (int u = 0; u < NU; ++u) {
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
      *reinterpret_cast<f32x4*>(&lds[p * PS + (lane & 3) * 4]) = t;
    }
  };

  // ---- this lane's A-fragment base in its wave's patch: pixel li of the tile (+ 4*lh channels)
  auto lane_base = [&](int oy0_) {
    const int tr = li / TW, tc = li - tr * TW;
    int cross = 0;
#pragma unroll
    for (int k = 1; k <= NB; ++k) cross += (oy0_ + tr >= k * a.Ho) ? 1 : 0;
    return ((S * tr + cross) * PW + tc * S) * PS + 4 * lh;
  };
  int a_base = lane_base(oy0);

  // ---- weight ring: chunk i of a stage = tap i/2, k-step 2*wave + (i&1)
  f32x4 ring[PIPE_RING];
  const unsigned wlane = lane * 16u;
#define PIPE_WCHUNK(dst, w, i) PIPE_GLOAD(dst, wlane, (w) + (((i) >> 1) * 8 + ((i) & 1)) * 256)

  // prologue: first stage
#pragma unroll
  for (int i = 0; i < PIPE_RING; ++i) PIPE_WCHUNK(ring[i], cur.w, i);
  stage_load(cur, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  stage_store();

  f32x16 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  for (;;) {
    // ---- what comes after the current stage (wave-uniform)
    const bool unit_end = cb + 1 == cb_end;
    int n_q = q, n_mt = mt, n_nf = nf, n_r0 = r0, n_ox0 = ox0, n_oy0 = oy0, n_cb = cb + 1, n_cb_end = cb_end;
    bool more = true;
    if (unit_end) {
      n_q = q + nwg;
      more = n_q < a.units;
      if (more) {
        int n_ksi, n_b0;
        decode(n_q, n_mt, n_nf, n_ksi, n_r0, n_ox0, n_b0, n_oy0);
        n_cb = n_ksi * a.cb_per;
        n_cb_end = n_cb + a.cb_per;
        nxt = make_stage(n_b0, n_oy0, n_ox0, n_nf, n_cb);
      } else {
        nxt = cur;                        // harmless re-loads of the last stage (never stored)
      }
    } else {
      nxt = cur;
      nxt.c0 += CB;
      nxt.w += (size_t)9 * 8 * 256;
    }

    // ---- K loop of the current stage: 18 chunks x 4 MFMAs; the next stage's patch loads go out behind the first chunk.
    // Outstanding loads when chunk i waits, oldest first (x' = issued during this stage):
    //   i = 0      : slots 0..8                                   -> slot 0 has landed at vmcnt(8)
    //   i = 1..8   : slots i..8, 0'..(i-1)' with the NS patch loads after 0'   -> vmcnt(8 + NS)
    //   i = 9      : 0', patch loads, 1'..8'                      -> vmcnt(8 + NS)
    //   i = 10..17 : patch loads, i'.., next stage's slots        -> vmcnt(8): the patch has landed as well
    {
      const float* ab = lds + a_base;
      f32x4 av = *reinterpret_cast<const f32x4*>(ab);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        f32x4 an = av;
        if (i + 1 < 18) {
          const int t = (i + 1) >> 1;                  // tap (MODE 0) / (class, tap) pair (MODE 1) of the next chunk
          const int ntx = MODE ? 1 + (S2D_CLS[t] & 1) : 3, tt = MODE ? S2D_TAP[t] : t;
          an = *reinterpret_cast<const f32x4*>(ab + ((tt / ntx) * PW + (tt % ntx)) * PS + ((i + 1) & 1) * 8);
        }
        if (i == 0 || i >= 10) PIPE_WAIT(ring[i % PIPE_RING], 8); else PIPE_WAIT(ring[i % PIPE_RING], 8 + NS);
        const f32x4 bv = ring[i % PIPE_RING];
        constexpr int dummy_ = 0;
        (void)dummy_;
        const int ai = MODE ? S2D_CLS[i >> 1] : 0;     // compile-time after unrolling
#ifdef SST_PIPE_ABLATE
        if (!(a.dbg & 16))
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[ai], 0, 0, 0);
#ifdef SST_PIPE_ABLATE
        if (a.dbg & 8) { PIPE_GLOAD(ring[i % PIPE_RING], wlane, a.wp); } else      // one cache-resident address: the load count (vmcnt) stays
#endif
        if (i + PIPE_RING < 18) PIPE_WCHUNK(ring[i % PIPE_RING], cur.w, i + PIPE_RING);
        else PIPE_WCHUNK(ring[i % PIPE_RING], nxt.w, i + PIPE_RING - 18);
        if (i == 0) stage_load(nxt, !unit_end);
        av = an;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef SST_PIPE_ABLATE
    if (unit_end && (a.dbg & 2)) {
      if (acc[0][0] == 12345.f) a.y[0] = acc[0][0];
      if (!more) return;
      a_base = lane_base(n_oy0);
    } else
#endif
    if (unit_end) {
      // ---- the 4 waves' K-partials -> LDS (NPAIR accumulators per round); afterwards wave w owns channels 8w .. 8w+7 of the
      // tile for all 32 pixels
      const int cl = 8 * wave + 4 * lh;    // first of this lane's 4 channels inside the 32-channel block
#pragma unroll
      for (int k0 = 0; k0 < NACC; k0 += NPAIR) {
        __syncthreads();                   // everyone has read the previous round's partials
#pragma unroll
        for (int k = 0; k < NPAIR; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            scratch[((k * 4 + wave) * 32 + row) * 33 + li] = acc[k0 + k][r];
            acc[k0 + k][r] = 0.f;
          }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPAIR; ++k) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = scratch[((k * 4 + 0) * 32 + li) * 33 + cl + j];
            t += scratch[((k * 4 + 1) * 32 + li) * 33 + cl + j];
            t += scratch[((k * 4 + 2) * 32 + li) * 33 + cl + j];
            t += scratch[((k * 4 + 3) * 32 + li) * 33 + cl + j];
            v[j] = t;
          }
          if (a.ksplit > 1) {
            const int tile = nf * a.n_mt + mt;
            const int part = cb / a.cb_per;
            *reinterpret_cast<f32x4*>(a.ws + (((size_t)part * a.total_tiles + tile) * NACC + k0 + k) * 1024 + li * 32 + cl) =
                f32x4{v[0], v[1], v[2], v[3]};
          } else {
            pipe_epilogue<TW, MODE>(a, v, r0, ox0, nf * 32 + cl, mt, lane, nf == 0 && wave == 0, k0 + k);
          }
        }
      }
      if (!more) return;
      a_base = lane_base(n_oy0);
    }
    stage_store();                         // next stage's patch slice: registers -> this wave's LDS region (LDS ops of a wave
                                           // complete in order: the K loop's reads above are done before these writes land)
    q = n_q; mt = n_mt; nf = n_nf; r0 = n_r0; ox0 = n_ox0; oy0 = n_oy0; cb = n_cb; cb_end = n_cb_end;
    cur = nxt;
  }
}

// Split-K: sum the partial slabs of one tile (and parity class) in fixed order, then the ordinary (wave-local) epilogue.
template <int TW, int MODE>
__global__ __launch_bounds__(CONV_NT) void pipe_reduce_kernel(PipeArgs a) {
  constexpr int NACC = MODE ? 4 : 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x / NACC, cls = blockIdx.x - tile * NACC;
  const int nf = tile / a.n_mt, mt = tile - nf * a.n_mt;
  const int ty = mt / a.tiles_x, tx = mt - ty * a.tiles_x;
  const int li = lane & 31, cl = 8 * wave + 4 * (lane >> 5);
  const size_t slab = (size_t)a.total_tiles * NACC * 1024;
  const float* src = a.ws + ((size_t)tile * NACC + cls) * 1024 + li * 32 + cl;
  f32x4 s = *reinterpret_cast<const f32x4*>(src);
  for (int k = 1; k < a.ksplit; ++k) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(src + k * slab);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += t[j];
  }
  float v[4] = {s[0], s[1], s[2], s[3]};
  pipe_epilogue<TW, MODE>(a, v, ty * (32 / TW), tx * TW, nf * 32 + cl, mt, lane, nf == 0 && wave == 0, cls);
}

struct PipePlan {
  int tw;          // 0 = shape not taken
  int ksplit, n_mt, tiles_x, nfc, ncb;
};

PipePlan pipe_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  PipePlan pl{};
  if (ksize != 3 || (stride != 1 && stride != 2) || (Cin % 64) || (Cout % 32) || B <= 0) return pl;
  if (stride == 2 && ((H | W) & 1)) return pl;
  if (Cout <= 64 && sst_conv_band_rows(B, H, W, Cin, Cout, ksize, stride)) return pl;   // the generator's trunk shape stays on the band kernel
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  int tw = 0;
  if (Wo % 8 == 0 && Ho >= 3) tw = 8;
  else if (Wo % 4 == 0 && Ho >= 7) tw = 4;
  else if (Wo % 2 == 0 && Ho >= 5) tw = 2;
  if (!tw) return pl;
  const int th = 32 / tw;
  pl.tiles_x = Wo / tw;
  pl.n_mt = pl.tiles_x * ((B * Ho + th - 1) / th);
  pl.nfc = Cout / 32;
  pl.ncb = Cin / 64;
  const long tiles = (long)pl.n_mt * pl.nfc;
  int maxks = 1;
  while (maxks < 4 && pl.ncb % (maxks * 2) == 0) maxks *= 2;
  if (tiles * maxks < 256) return pl;     // too little work to fill the chip this way: the general kernel keeps it
  int ks = 1;
  while (tiles * ks < 1024 && ks < maxks) ks *= 2;
  pl.ksplit = ks;
  pl.tw = tw;
  return pl;
}

constexpr int PIPE_CUS = 256;

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_conv_pipe_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  return pipe_plan(B, H, W, Cin, Cout, ksize, stride).tw;      // tile width (8 / 4 / 2) when taken, else 0
}
SST_API int sst_conv_pipe_stat_tiles(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  return pipe_plan(B, H, W, Cin, Cout, ksize, stride).n_mt;
}
SST_API int64_t sst_conv_pipe_ws_floats(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  const PipePlan pl = pipe_plan(B, H, W, Cin, Cout, ksize, stride);
  return (pl.tw && pl.ksplit > 1) ? (int64_t)pl.ksplit * pl.n_mt * pl.nfc * 1024 : 0;
}

// Can a batch of B images be run with per-group coefficients (groups of grp_images consecutive images)?  The group boundary must fall
// on a tile boundary of the tall image.
SST_API int sst_conv_pipe_groups_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images) {
  const PipePlan pl = pipe_plan(B, H, W, Cin, Cout, ksize, stride);
  if (!pl.tw || grp_images <= 0 || B % grp_images) return 0;
  const int Ho = (H - 1) / stride + 1;
  return (grp_images * Ho) % (32 / pl.tw) == 0;
}

// grp_images > 0: the per-channel coefficient arrays (in_scale / in_shift, epi_scale / epi_shift) are [B / grp_images][channels],
// images b*grp_images .. (b+1)*grp_images - 1 use row b (several passes of the network batched as one tall image, each with the
// BatchNorm coefficients of its own batch statistics).  0: one row for the whole batch.
SST_API int sst_conv_pipe_fwd_grp(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                                  const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                                  float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                                  const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, float* ws, int B,
                                  int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images, void* stream) {
  SST_REQUIRE(x && wp && y, "sst_conv_pipe_fwd: null pointer");
  const PipePlan pl = pipe_plan(B, H, W, Cin, Cout, ksize, stride);
  SST_REQUIRE(grp_images == 0 || grp_images == B || sst_conv_pipe_groups_ok(B, H, W, Cin, Cout, ksize, stride, grp_images),
              "sst_conv_pipe_fwd: coefficient groups of %d images do not end on tile boundaries (B=%d H=%d W=%d stride=%d)", grp_images, B,
              H, W, stride);
  SST_REQUIRE(pl.tw, "sst_conv_pipe_fwd: shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d stride=%d is not taken by the pipelined kernel", B,
              H, W, Cin, Cout, ksize, stride);
  SST_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sst_conv_pipe_fwd: in_scale/in_shift must come together");
  SST_REQUIRE(!stats || stats_cnt, "sst_conv_pipe_fwd: stats needs stats_cnt");
  SST_REQUIRE(!epi_partial || (epi_y && !stats && ((epi_scale == nullptr) == (epi_shift == nullptr))),
              "sst_conv_pipe_fwd: backward partials need epi_y and exclude forward stats");
  SST_REQUIRE(pl.ksplit == 1 || ws, "sst_conv_pipe_fwd: this shape splits K over workgroups and needs the workspace");
  SST_REQUIRE((int64_t)B * H * W * Cin < (1ll << 29), "sst_conv_pipe_fwd: input too large for 32-bit byte offsets");
  PipeArgs a;
  a.x = x; a.wp = wp; a.y = y; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.in_slope_const = in_slope_const; a.in_act = in_act; a.stats = stats; a.stats_cnt = stats_cnt;
  a.epi_y = epi_y; a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_slope = epi_slope; a.epi_slope_const = epi_slope_const;
  a.epi_act = epi_act; a.epi_partial = epi_partial; a.ws = ws;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.R = B * a.Ho;
  a.tiles_x = pl.tiles_x; a.n_mt = pl.n_mt; a.nfc = pl.nfc; a.total_tiles = pl.n_mt * pl.nfc; a.ncb = pl.ncb;
  a.ksplit = pl.ksplit; a.cb_per = pl.ncb / pl.ksplit; a.units = a.total_tiles * pl.ksplit;
  a.gB = (grp_images > 0 && grp_images < B) ? grp_images : 0;
  a.grows = a.gB * a.Ho;
#ifdef SST_PIPE_ABLATE
  a.dbg = sst_env("SST_PIPE_DBG") ? atoi(sst_env("SST_PIPE_DBG")) : 0;
#endif
  int wg_per_cu = stride == 1 ? 3 : 2;                 // = the kernels' launch bounds (register-limited)
  if (const char* e = sst_env("SST_PIPE_GPC")) {        // dev: fewer resident workgroups leave room for kernels of other streams
    const int v = atoi(e);
    if (v >= 1 && v < wg_per_cu) wg_per_cu = v;
  }
  const int grid = a.units < wg_per_cu * PIPE_CUS ? a.units : wg_per_cu * PIPE_CUS;
  hipStream_t st = sst_stream(stream);
#define SST_PIPE_LAUNCH(S_, TW_)                                                                  \
  do {                                                                                            \
    if (a.cb_per > 1) conv_pipe_kernel<S_, TW_, 0, true><<<grid, CONV_NT, 0, st>>>(a);            \
    else conv_pipe_kernel<S_, TW_, 0, false><<<grid, CONV_NT, 0, st>>>(a);                        \
    SST_LAUNCH_CHECK("conv_pipe_kernel");                                                         \
    if (pl.ksplit > 1) {                                                                          \
      pipe_reduce_kernel<TW_, 0><<<a.total_tiles, CONV_NT, 0, st>>>(a);                           \
      SST_LAUNCH_CHECK("pipe_reduce_kernel");                                                     \
    }                                                                                             \
  } while (0)
  if (stride == 1) {
    if (pl.tw == 8) SST_PIPE_LAUNCH(1, 8); else if (pl.tw == 4) SST_PIPE_LAUNCH(1, 4); else SST_PIPE_LAUNCH(1, 2);
  } else {
    if (pl.tw == 8) SST_PIPE_LAUNCH(2, 8); else if (pl.tw == 4) SST_PIPE_LAUNCH(2, 4); else SST_PIPE_LAUNCH(2, 2);
  }
#undef SST_PIPE_LAUNCH
  return SST_OK;
}

SST_API int sst_conv_pipe_fwd(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                              const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                              float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                              const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, float* ws, int B,
                              int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
  return sst_conv_pipe_fwd_grp(x, wp, y, bias, in_scale, in_shift, in_slope, in_slope_const, in_act, stats, stats_cnt, epi_y, epi_scale,
                               epi_shift, epi_slope, epi_slope_const, epi_act, epi_partial, ws, B, H, W, Cin, Cout, ksize, stride, 0, stream);
}

// ---- stride-2 data-gradient on the pipelined kernel (MODE 1).  dx [B,H,W,Cin] = conv_transpose(dy [B,H/2,W/2,Cout]) for
// y = conv3x3(x, stride 2, pad 1); wp = the buffer of sst_conv_s2_dgrad_pack (its 5th section).  Even H, W; Cout % 64 == 0
// (K blocks), Cin % 32 == 0.
int64_t sst_conv_s2_dgrad_pipe_section(int Cout, int Cin);      // conv_fwd.hip
SST_API int sst_conv_s2_dgrad_pipe_bwdstats_grp(const float* dy, const float* wp, float* dx, float* ws, const float* epi_y,
                                                const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                                float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                                int Cout, int grp_images, void* stream);

namespace {
PipePlan pipe_plan_s2d(int B, int H, int W, int Cin, int Cout) {
  PipePlan pl{};
  if (B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || (Cout % 64) || (Cin % 32)) return pl;
  const int Hd = H / 2, Wd = W / 2;                  // class grid = dY size
  int tw = 0;
  if (Wd % 8 == 0 && Hd >= 3) tw = 8;
  else if (Wd % 4 == 0 && Hd >= 7) tw = 4;
  else if (Wd % 2 == 0 && Hd >= 5) tw = 2;
  if (!tw) return pl;
  const int th = 32 / tw;
  pl.tiles_x = Wd / tw;
  pl.n_mt = pl.tiles_x * ((B * Hd + th - 1) / th);
  pl.nfc = Cin / 32;
  pl.ncb = Cout / 64;
  const long tiles = (long)pl.n_mt * pl.nfc;
  int maxks = 1;
  while (maxks < 4 && pl.ncb % (maxks * 2) == 0) maxks *= 2;
  if (tiles * maxks < 256) return pl;
  int ks = 1;
  while (tiles * ks < 1024 && ks < maxks) ks *= 2;
  // measured against conv_s2dgrad4_kernel (tools/time_s2d.py, B = 16): 96 px 38.7 vs 39.7 us, 48 px 34.0 vs 35.4, 12 px 38.2 vs 46.9
  // (its 8x4 tiles waste 44 % of the MFMAs on a 6x6 class grid), but 24 px 40.2 vs 37.6: four epilogues per unit + a split-K
  // reduce launch cost more than the 25 % tile waste they remove - that middle case stays on the merged-classes kernel
  if (tw == 4 && ks > 1) return pl;
  pl.ksplit = ks;
  pl.tw = tw;
  return pl;
}
}  // namespace

SST_API int sst_conv_s2_dgrad_pipe_supported(int B, int H, int W, int Cin, int Cout) { return pipe_plan_s2d(B, H, W, Cin, Cout).tw; }
SST_API int64_t sst_conv_s2_dgrad_pipe_ws_floats(int B, int H, int W, int Cin, int Cout) {
  const PipePlan pl = pipe_plan_s2d(B, H, W, Cin, Cout);
  return (pl.tw && pl.ksplit > 1) ? (int64_t)pl.ksplit * pl.n_mt * pl.nfc * 4 * 1024 : 0;
}
SST_API int sst_conv_s2_dgrad_pipe_stat_tiles(int B, int H, int W, int Cin, int Cout) { return pipe_plan_s2d(B, H, W, Cin, Cout).n_mt * 4; }
// ... with the BatchNorm / activation backward partials of dx against epi_y (the saved output of the layer below, [B,H,W,Cin]) in
// the epilogue: epi_partial [sst_conv_s2_dgrad_pipe_stat_tiles()][3][Cin], the layout sst_bwd_finalize consumes (null: none).
SST_API int sst_conv_s2_dgrad_pipe_bwdstats(const float* dy, const float* wp, float* dx, float* ws, const float* epi_y,
                                            const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                            float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                            int Cout, void* stream) {
  return sst_conv_s2_dgrad_pipe_bwdstats_grp(dy, wp, dx, ws, epi_y, epi_scale, epi_shift, epi_slope, epi_slope_const, epi_act, epi_partial,
                                             B, H, W, Cin, Cout, 0, stream);
}
// ... with coefficient groups (see sst_conv_pipe_fwd_grp): epi_scale / epi_shift [B / grp_images][Cin], the partial rows of a pass are
// the p-th of B / grp_images equal consecutive ranges (a tile of the class grid never straddles a pass: sst_conv_s2_dgrad_pipe_groups_ok)
SST_API int sst_conv_s2_dgrad_pipe_groups_ok(int B, int H, int W, int Cin, int Cout, int grp_images) {
  const PipePlan pl = pipe_plan_s2d(B, H, W, Cin, Cout);
  if (!pl.tw || grp_images <= 0 || B % grp_images) return 0;
  return (grp_images * (H / 2)) % (32 / pl.tw) == 0;
}
SST_API int sst_conv_s2_dgrad_pipe_bwdstats_grp(const float* dy, const float* wp, float* dx, float* ws, const float* epi_y,
                                                const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                                float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                                int Cout, int grp_images, void* stream) {
  SST_REQUIRE(dy && wp && dx, "sst_conv_s2_dgrad_pipe: null pointer");
  SST_REQUIRE(grp_images == 0 || grp_images == B || !epi_partial || sst_conv_s2_dgrad_pipe_groups_ok(B, H, W, Cin, Cout, grp_images),
              "sst_conv_s2_dgrad_pipe: coefficient groups of %d images do not end on tile boundaries", grp_images);
  SST_REQUIRE(!epi_partial || (epi_y && ((epi_scale == nullptr) == (epi_shift == nullptr))), "sst_conv_s2_dgrad_pipe: backward partials need epi_y");
  const PipePlan pl = pipe_plan_s2d(B, H, W, Cin, Cout);
  SST_REQUIRE(pl.tw, "sst_conv_s2_dgrad_pipe: shape B=%d H=%d W=%d Cin=%d Cout=%d is not taken by the pipelined kernel", B, H, W, Cin, Cout);
  SST_REQUIRE(pl.ksplit == 1 || ws, "sst_conv_s2_dgrad_pipe: this shape splits K over workgroups and needs the workspace");
  SST_REQUIRE((int64_t)B * (H / 2) * (W / 2) * Cout < (1ll << 29), "sst_conv_s2_dgrad_pipe: input too large for 32-bit byte offsets");
  PipeArgs a{};
  a.x = dy; a.wp = wp + sst_conv_s2_dgrad_pipe_section(Cout, Cin); a.y = dx; a.ws = ws;
  a.in_act = ACT_NONE;
  a.epi_y = epi_y; a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.epi_slope = epi_slope; a.epi_slope_const = epi_slope_const;
  a.epi_act = epi_act; a.epi_partial = epi_partial;
  a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout; a.Cout = Cin;      // GEMM view: K = dY channels, N = dX channels
  a.Ho = a.H; a.Wo = a.W; a.R = B * a.Ho;
  a.tiles_x = pl.tiles_x; a.n_mt = pl.n_mt; a.nfc = pl.nfc; a.total_tiles = pl.n_mt * pl.nfc; a.ncb = pl.ncb;
  a.ksplit = pl.ksplit; a.cb_per = pl.ncb / pl.ksplit; a.units = a.total_tiles * pl.ksplit;
#ifdef SST_PIPE_ABLATE
  a.dbg = 0;
#endif
  a.gB = (grp_images > 0 && grp_images < B && epi_partial) ? grp_images : 0;
  a.grows = a.gB * a.Ho;
  const int grid = a.units < 2 * PIPE_CUS ? a.units : 2 * PIPE_CUS;
  hipStream_t st = sst_stream(stream);
#define SST_S2D_LAUNCH(TW_)                                                                       \
  do {                                                                                            \
    if (a.cb_per > 1) conv_pipe_kernel<1, TW_, 1, true><<<grid, CONV_NT, 0, st>>>(a);             \
    else conv_pipe_kernel<1, TW_, 1, false><<<grid, CONV_NT, 0, st>>>(a);                         \
    SST_LAUNCH_CHECK("conv_pipe_kernel (stride-2 data-gradient)");                                \
    if (pl.ksplit > 1) {                                                                          \
      pipe_reduce_kernel<TW_, 1><<<a.total_tiles * 4, CONV_NT, 0, st>>>(a);                       \
      SST_LAUNCH_CHECK("pipe_reduce_kernel (stride-2 data-gradient)");                            \
    }                                                                                             \
  } while (0)
  if (pl.tw == 8) SST_S2D_LAUNCH(8); else if (pl.tw == 4) SST_S2D_LAUNCH(4); else SST_S2D_LAUNCH(2);
#undef SST_S2D_LAUNCH
  return SST_OK;
}
SST_API int sst_conv_s2_dgrad_pipe(const float* dy, const float* wp, float* dx, float* ws, int B, int H, int W, int Cin, int Cout,
                                   void* stream) {
  return sst_conv_s2_dgrad_pipe_bwdstats(dy, wp, dx, ws, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, B, H, W, Cin, Cout, stream);
}


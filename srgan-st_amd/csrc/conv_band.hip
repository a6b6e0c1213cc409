// 3x3 / stride 1 / pad 1 convolution with 64 input channels, NHWC, fp32 MFMA - "band" decomposition for the
// SRResNet trunk shape (reference model.py:173,176 _ResidualConvBlock convs, model.py:113 conv2, and their
// data-gradients incl. the fused BatchNorm-backward stage, see sst_conv_dgrad_fused).
//
// Why a second kernel: the trunk conv is small (B*24*24 = 9216 pixels x 64 channels at the bench size).  The general
// kernel (conv_fwd.hip) cuts it into 576 tiles of 32 px x 32 ch, i.e. 2.25 workgroups per CU: a quarter of the CUs run
// 3 tiles while the rest run 2, and every tile re-reads its 74 KB of weights from L2.  Here a workgroup owns a BAND of
// R full image rows (R*W a multiple of 16) x 16 output channels:
//   * v_mfma_f32_16x16x4_f32 with M = 16 output channels (A = weights), N = 16 pixels (B = activations), K = 4 inputs;
//     the band is NB = R*W/16 pixel blocks = NB accumulators per wave;
//   * the 4 waves split K by INPUT CHANNEL (wave w owns channels 16w..16w+15 of all 9 taps), so each wave stages its own
//     channel slice of the patch into its own LDS region and needs no workgroup barrier before the K loop;
//   * per tap a wave needs ONE 16-B weight load per lane (9 per kernel, all issued before staging) and one
//     ds_read_b128 per pixel block, which feeds 4 MFMAs (the 4 k-steps take channels {s, 4+s, 8+s, 12+s});
//   * R = 6 at 24x24 (R = 3 at 48x48) gives B*4 bands x 4 channel groups = 256 workgroups for B = 16: one per CU, every
//     SIMD runs exactly 324 MFMAs.
// The K partials of the 4 waves are summed through LDS; the epilogue (bias, residual, BatchNorm statistics, backward
// partials) matches conv_epilogue.h with one statistics tile per band.
#include "conv_common.h"
#include "conv_epilogue.h"
#include <cstdlib>
#include <type_traits>

int sst_launch_conv_band(const Conv3Args& a, int R, hipStream_t st, const BandAcc* acc);
int sst_conv_band_rows(int B, int H, int W, int Cin, int Cout, int ksize, int stride);

namespace {

}  // namespace


namespace {

constexpr int PSTR = 20;   // LDS floats per patch pixel of one wave's 16-channel slice (16 + 4 pad: conflict-free b128 reads)

// FUSED: the two-tensor input forms (BatchNorm-backward apply, residual sum) and the backward epilogue sums are compiled in;
// the plain forward instance carries neither their registers nor their code.
template <int NB, bool FUSED>
__global__ __launch_bounds__(CONV_NT, (NB == 3 ? 3 : 1)) void conv_band_kernel(Conv3Args a, int R, int nbands, BandAcc ba) {   // NB = 3: keep 3 workgroups per CU (<= 168 VGPRs)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sstat[4][3][16];
  __shared__ float saff[4][3][16];
  __shared__ float sslope[4];
  __shared__ double ssq[4][16];
  __shared__ double sbw[4][3][16];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int W = a.W, PW = W + 2, npatch = (R + 2) * PW;
  const int mt = blockIdx.x;
  const int b = mt / nbands, y0 = (mt - b * nbands) * R;
  const int g = blockIdx.y;                     // group of 16 output channels
  float* P = lds + wave * npatch * PSTR;        // this wave's patch slice

  // ---- weights of this (channel group, wave): 9 x 16 B per lane, in flight while the patch is staged
  f32x4 wv[9];
  {
    const float* wb = a.wp + packed_floats_base(a.Cout, 64, 9) + ((size_t)g * 9 * 4 + wave) * 256 + lane * 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(wb + t * 4 * 256);
  }

  // ---- stage this wave's 16-channel slice of the (R+2) x (W+2) patch; padding stays zero
  {
    const int q4 = (lane & 3) * 4, c = wave * 16 + q4;
    const float slope = a.in_slope ? a.in_slope[0] : a.in_slope_const;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    f32x4 kA = {1.f, 1.f, 1.f, 1.f}, kB = {0.f, 0.f, 0.f, 0.f}, kC = {0.f, 0.f, 0.f, 0.f};
    if (a.in_scale) {
      sc = *reinterpret_cast<const f32x4*>(a.in_scale + c);
      sh = *reinterpret_cast<const f32x4*>(a.in_shift + c);
    }
    if (a.in_cA) {
      kA = *reinterpret_cast<const f32x4*>(a.in_cA + c);
      kB = *reinterpret_cast<const f32x4*>(a.in_cB + c);
      kC = *reinterpret_cast<const f32x4*>(a.in_cC + c);
    }
    bool have_aff = a.in_scale != nullptr, have_k = a.in_cA != nullptr;
    // BatchNorm affine of the input from the producer's accumulators (accumulator mode), before the patch loads (keeping the
    // 32 accumulator registers alive across the patch loads was measured: spills, every variant slower).
    // All 64 lanes load: lane = (channel j = lane % 16, part = lane / 16) takes replicas part, part + 4, ... (independent
    // 16-B loads); the 4 parts are combined in fixed order with shuffles; lane j < 16 of wave w then owns channel 16w + j.
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    constexpr int MAXR = 4;                                 // replicas per lane (nrep <= 16, checked by the entry points)
    f64x2 accv[MAXR];
    float pre_g = 0.f, pre_b = 0.f;                       // gamma / beta of the lane's channel: in flight with the accumulators
    if (ba.in_acc) {
      const int ch = wave * 16 + (lane & 15), part = lane >> 4;
      pre_g = ba.in_gamma[ch];
      pre_b = ba.in_beta[ch];
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const int r = part + 4 * k;
        accv[k] = f64x2{0.0, 0.0};
        if (r < ba.nrep) accv[k] = *reinterpret_cast<const f64x2*>(ba.in_acc + ((size_t)r * 64 + ch) * 2);
      }
    }
    auto affine_from_acc = [&]() {
      double A = 0.0, Bq = 0.0;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) { A += accv[k][0]; Bq += accv[k][1]; }
      A += __shfl_xor(A, 16, 64);
      Bq += __shfl_xor(Bq, 16, 64);
      A += __shfl_xor(A, 32, 64);
      Bq += __shfl_xor(Bq, 32, 64);
      if (lane < 16) {
        const int ch = wave * 16 + lane;
        const double n = (double)ba.in_n;
        const double mean = A / n;
        double m2 = Bq - A * mean;                         // = sum (x - mean)^2
        if (m2 < 0.0) m2 = 0.0;
        const float var = (float)(m2 / n);
        const float rstd = 1.f / sqrtf(var + ba.in_eps);
        const float scv = pre_g * rstd;
        const float shv = pre_b - (float)mean * scv;
        saff[wave][0][lane] = scv;
        saff[wave][1][lane] = shv;
        if (blockIdx.x == 0 && blockIdx.y == 0) {          // one workgroup publishes the statistics for the backward pass
          if (ba.o_mean) { ba.o_mean[ch] = (float)mean; ba.o_rstd[ch] = rstd; ba.o_scale[ch] = scv; ba.o_shift[ch] = shv; }
          if (ba.run_mean) {
            ba.run_mean[ch] = (1.f - ba.momentum) * ba.run_mean[ch] + ba.momentum * (float)mean;
            ba.run_var[ch] = (1.f - ba.momentum) * ba.run_var[ch] + ba.momentum * (float)(m2 / fmax(n - 1.0, 1.0));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private exchange through LDS
      f32x4 s4, t4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s4[j] = saff[wave][0][q4 + j];
        t4[j] = saff[wave][1][q4 + j];
      }
      if (ba.in_target == 0) { sc = s4; sh = t4; have_aff = true; }
      else { kB = s4; kC = t4; have_k = true; }
    };
    // FUSED instance: reduce before the patch loads (reducing after issuing them was measured there: the compiler spills in every
    // variant tried, 3x slower).  Plain forward instance (92 VGPRs): the accumulator loads were issued first, so the reduction
    // only waits for them (in-order vmcnt) while the patch loads issued below stay in flight behind it.
    constexpr bool LATE_ACC = !FUSED;
    if (ba.in_acc && !LATE_ACC) affine_from_acc();
    if (FUSED && ba.bw_in_acc) {
      // BatchNorm-backward coefficients of the input from the producer's accumulators [nrep][64][4] (same lane mapping)
      double S0 = 0.0, S1 = 0.0, S2 = 0.0;
      const float mu = ba.bw_mean[wave * 16 + (lane & 15)], rs = ba.bw_rstd[wave * 16 + (lane & 15)],
                  ga = ba.bw_gamma[wave * 16 + (lane & 15)];          // in flight with the accumulator loads
      {
        const int ch = wave * 16 + (lane & 15), part = lane >> 4;
        f64x2 v0[MAXR], v1[MAXR];
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
          const int r = part + 4 * k;
          v0[k] = v1[k] = f64x2{0.0, 0.0};
          if (r < ba.nrep) {
            const double* q = ba.bw_in_acc + ((size_t)r * 64 + ch) * 4;
            v0[k] = *reinterpret_cast<const f64x2*>(q);
            v1[k] = *reinterpret_cast<const f64x2*>(q + 2);
          }
        }
#pragma unroll
        for (int k = 0; k < MAXR; ++k) { S0 += v0[k][0]; S1 += v0[k][1]; S2 += v1[k][0]; }
        S0 += __shfl_xor(S0, 16, 64); S1 += __shfl_xor(S1, 16, 64); S2 += __shfl_xor(S2, 16, 64);
        S0 += __shfl_xor(S0, 32, 64); S1 += __shfl_xor(S1, 32, 64); S2 += __shfl_xor(S2, 32, 64);
      }
      const bool pub = blockIdx.x == 0 && blockIdx.y == 0;
      float s2f = 0.f;
      if (lane < 16) {
        const int ch = wave * 16 + lane;
        // sum gz*(y - mean) = S1 - mean*S0 cancels (both terms grow with |mean|): evaluated in fp64 like the sums themselves - an
        // error here is a COMMON-MODE error of dy over the whole channel, which later signed sums amplify by sqrt(N)
        const double sgh_d = (double)rs * (S1 - (double)mu * S0);     // same arithmetic as bwd_finalize2_kernel
        const float s0 = (float)S0, sgh = (float)sgh_d;
        const float m1 = (float)(S0 / (double)ba.bw_n), m2 = (float)(sgh_d / (double)ba.bw_n);
        const float aa = ga * rs;
        saff[wave][0][lane] = aa;
        saff[wave][1][lane] = -aa * rs * m2;
        saff[wave][2][lane] = -aa * m1 + aa * rs * mu * m2;
        if (pub) { ba.o_dgamma[ch] = sgh; ba.o_dbeta[ch] = s0; }
        s2f = (float)S2;
      }
      if (pub && ba.o_dslope) {                              // scalar slope gradient: sum over all 64 channels, fixed order
        s2f += __shfl_xor(s2f, 1, 64); s2f += __shfl_xor(s2f, 2, 64); s2f += __shfl_xor(s2f, 4, 64); s2f += __shfl_xor(s2f, 8, 64);
        if (lane == 0) sslope[wave] = s2f;
        __syncthreads();                                     // workgroup-uniform condition
        if (tid == 0) ba.o_dslope[0] = ((sslope[0] + sslope[1]) + sslope[2]) + sslope[3];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kA[j] = saff[wave][0][q4 + j];
        kB[j] = saff[wave][1][q4 + j];
        kC[j] = saff[wave][2][q4 + j];
      }
      have_k = true;
    }
    const float* xb = a.x + (size_t)b * a.H * W * 64 + c;
    const float* x2b = (FUSED && a.in2) ? a.in2 + (size_t)b * a.H * W * 64 + c : nullptr;
    float* sob = (a.side_out && g == 0) ? a.side_out + (size_t)b * a.H * W * 64 + c : nullptr;
    // patch pixel p = it*16 + lane/4  ->  (py, px), advanced incrementally (no per-iteration division)
    int p = lane >> 2;
    int py = p / PW, px = p - py * PW;
    constexpr int UN = 7;      // 24x24: the 104-pixel patch is ONE batch of loads (two batches = two exposed latencies)
    const int nit = (a.dbg & 1) ? 0 : (npatch + 15) / 16;
    auto do_batch = [&](const int it0, const bool late) {
      f32x4 v[UN], yv[UN];
      int off[UN], lp[UN];
      bool ok[UN], own[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int iy = y0 - 1 + py, ix = px - 1;
        ok[u] = (it0 + u < nit) && p < npatch && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)W;
        own[u] = py >= 1 && py <= R;
        lp[u] = p < npatch ? p : -1;
        off[u] = (iy * W + ix) * 64;
        // branch-free loads (padding slots read the image origin, in bounds, and are zeroed below): straight-line code keeps
        // the compiler's vmcnt waits counted, so the accumulator reduction does not wait for the patch
        const int ldo = ok[u] ? off[u] : 0;
        v[u] = *reinterpret_cast<const f32x4*>(xb + ldo);
        yv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (x2b) yv[u] = *reinterpret_cast<const f32x4*>(x2b + ldo);
        p += 16;
        px += 16;
        if (px >= PW) { px -= PW; ++py; }
        if (px >= PW) { px -= PW; ++py; }
      }
      if (late) affine_from_acc();
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (lp[u] < 0 || it0 + u >= nit) continue;
        f32x4 t = v[u];
        {
          if (x2b) {                        // fused BatchNorm-backward apply (see Conv3Args)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float gz = t[j];
              if (a.in_act == ACT_SLOPE) {
                const float z = fmaf(yv[u][j], sc[j], sh[j]);
                gz = z > 0.f ? gz : gz * slope;
              }
              t[j] = have_k ? fmaf(kA[j], gz, fmaf(kB[j], yv[u][j], kC[j])) : gz;
            }
            if (sob && own[u] && ok[u]) *reinterpret_cast<f32x4*>(sob + off[u]) = t;
          } else {
            if (have_aff) {
#pragma unroll
              for (int j = 0; j < 4; ++j) t[j] = fmaf(t[j], sc[j], sh[j]);
            }
            if (a.in_act == ACT_SLOPE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : t[j] * slope;
            }
          }
        }
        if (!ok[u]) t = f32x4{0.f, 0.f, 0.f, 0.f};            // padding stays exactly zero
        *reinterpret_cast<f32x4*>(&P[lp[u] * PSTR + q4]) = t;
      }
    };
    if (LATE_ACC && ba.in_acc && nit <= UN) {
      do_batch(0, true);                  // one batch of loads: accumulator reduction behind them
    } else {
      if (LATE_ACC && ba.in_acc) affine_from_acc();
      for (int it0 = 0; it0 < nit; it0 += UN) do_batch(it0, false);
    }
  }

  // ---- per-lane LDS offsets of the NB pixel blocks: pixel n = blk*16 + lane%16 of the band -> patch (r, c) top-left
  const int lp16 = lane & 15, lj = lane >> 4;
  int boff[NB];
  {
    int r = lp16 / W, c = lp16 - r * W;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      boff[k] = (r * PW + c) * PSTR + 4 * lj;
      c += 16;
      if (c >= W) { c -= W; ++r; }
      if (c >= W) { c -= W; ++r; }
    }
  }

  // ---- epilogue operands (wave w finalises blocks w, w+4, w+8: lane -> pixel blk*16 + lane%16, output channels
  // 16g + 4*(lane/16) .. +3): addresses now, and the epilogue's global reads (residual, bias, epi_y + its affine) are
  // issued here so that their latency hides behind the K loop instead of following the cross-wave reduction
  constexpr int MYB = (NB + 3) / 4;
  const int c0 = g * 16 + 4 * lj;
  size_t obase[MYB];
  bool have[MYB];
  f32x4 pre_res[MYB], pre_ey[MYB];
  f32x4 pre_bias = {0.f, 0.f, 0.f, 0.f}, esc = {1.f, 1.f, 1.f, 1.f}, esh = {0.f, 0.f, 0.f, 0.f};
  const bool epi_sums = FUSED && (a.epi_partial || ba.bw_st_acc);
#pragma unroll
  for (int i = 0; i < MYB; ++i) {
    const int blk = wave + 4 * i;
    have[i] = blk < NB;
    const int n = blk * 16 + lp16;
    const int r = n / W, c = n - r * W;
    obase[i] = have[i] ? (((size_t)b * a.H + y0 + r) * W + c) * a.Cout + c0 : 0;
    pre_res[i] = pre_ey[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (have[i] && a.residual) pre_res[i] = *reinterpret_cast<const f32x4*>(a.residual + obase[i]);
    if (have[i] && epi_sums) pre_ey[i] = *reinterpret_cast<const f32x4*>(a.epi_y + obase[i]);
  }
  if (a.bias) pre_bias = *reinterpret_cast<const f32x4*>(a.bias + c0);
  if (epi_sums && a.epi_scale) {
    esc = *reinterpret_cast<const f32x4*>(a.epi_scale + c0);
    esh = *reinterpret_cast<const f32x4*>(a.epi_shift + c0);
  }

  f32x4 acc[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (!(a.dbg & 2)) {
    // the patch slice is wave-private: only this wave's own LDS writes have to land (no workgroup barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 bv[2][NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bv[0][k] = *reinterpret_cast<const f32x4*>(&P[boff[k]]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) {
        const int toff = (((t + 1) / 3) * PW + (t + 1) % 3) * PSTR;
#pragma unroll
        for (int k = 0; k < NB; ++k) bv[(t + 1) & 1][k] = *reinterpret_cast<const f32x4*>(&P[boff[k] + toff]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][s], bv[t & 1][k][s], acc[k], 0, 0, 0);
    }
  }

  if (a.dbg & 4) {
    if (acc[0][0] == 12345.f) a.y[0] = acc[0][0];
    return;
  }

  // ---- sum the 4 waves' K partials through LDS (overlays the patches): red[wave][blk][lane] (16 B each)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NB; ++k) *reinterpret_cast<f32x4*>(&lds[((wave * NB + k) * 64 + lane) * 4]) = acc[k];
  __syncthreads();

  const int npx = R * W;
  f32x4 v[MYB];
#pragma unroll
  for (int i = 0; i < MYB; ++i) {
    const int blk = wave + 4 * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (have[i]) {
      f32x4 s = *reinterpret_cast<const f32x4*>(&lds[((0 * NB + blk) * 64 + lane) * 4]);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[((w * NB + blk) * 64 + lane) * 4]);
        s += t;
      }
      s += pre_bias;
      s += pre_res[i];
      *reinterpret_cast<f32x4*>(a.y + obase[i]) = s;
      v[i] = s;
    }
  }

  // sum over the 16 lanes that share lane/16 (same output channels): DPP row rotations (VALU speed, no LDS crossbar);
  // every lane of the row ends up with the row total
  auto ror = [](float t, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  auto reduce16 = [&](float t) {
    t += ror(t, std::integral_constant<int, 0x128>{});   // row_ror:8
    t += ror(t, std::integral_constant<int, 0x124>{});   // row_ror:4
    t += ror(t, std::integral_constant<int, 0x122>{});   // row_ror:2
    t += ror(t, std::integral_constant<int, 0x121>{});   // row_ror:1
    return t;
  };
  auto ror_d = [](double t, auto ctrl) {
    const long long b = __builtin_bit_cast(long long, t);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), decltype(ctrl)::value, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  };
  auto reduce16_d = [&](double t) {
    t += ror_d(t, std::integral_constant<int, 0x128>{});
    t += ror_d(t, std::integral_constant<int, 0x124>{});
    t += ror_d(t, std::integral_constant<int, 0x122>{});
    t += ror_d(t, std::integral_constant<int, 0x121>{});
    return t;
  };

  if (a.stats || ba.st_acc) {
    // per-band (sum, sum of squares) per output channel in ONE pass: the squares are accumulated in fp64, so the centred
    // M2 = q - sum^2/n (partial-tile mode) and Chan's combination (accumulator mode adds q itself) lose nothing to
    // cancellation, and the second pass over the values with its extra barrier is gone
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f};
    double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < MYB; ++i) {
      s1 += v[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = fma((double)v[i][j], (double)v[i][j], q[j]);     // v is 0 where !have
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] = reduce16(s1[j]);
      q[j] = reduce16_d(q[j]);
    }
    if (lp16 == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sstat[wave][0][4 * lj + j] = s1[j];
        ssq[wave][4 * lj + j] = q[j];
      }
    }
    __syncthreads();
    if (tid < 16) {
      const float tot = sstat[0][0][tid] + sstat[1][0][tid] + sstat[2][0][tid] + sstat[3][0][tid];
      const double qt = ((ssq[0][tid] + ssq[1][tid]) + ssq[2][tid]) + ssq[3][tid];
      if (a.stats) {
        float* st = a.stats + (size_t)mt * 2 * a.Cout + g * 16 + tid;
        const double m2 = qt - (double)tot * (double)tot / (double)npx;
        st[0] = tot;
        st[a.Cout] = (float)(m2 > 0.0 ? m2 : 0.0);
        if (tid == 0 && g == 0) a.stats_cnt[mt] = (float)npx;
      }
      if (ba.st_acc) {      // A += sum, Bq += sum of squares   (fp64 hardware atomics, no return value needed)
        double* p = ba.st_acc + ((size_t)(b % ba.nrep) * a.Cout + g * 16 + tid) * 2;
        __builtin_amdgcn_global_atomic_fadd_f64(p, (double)tot);
        __builtin_amdgcn_global_atomic_fadd_f64(p + 1, qt);
      }
    }
  }

  if (epi_sums) {
    // backward partials of the stored g against epi_y (see Conv3Args): sums of (gz, gz*y, g*min(z,0)) over the band
    const float eslope = a.epi_slope ? a.epi_slope[0] : a.epi_slope_const;
    // fp64 from the first add (terms formed in fp32 like the reference's): the sums are signed and largely cancel, and what a band
    // loses in fp32 is not given back by the fp64 accumulators behind it
    double q[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) q[k][j] = 0.0;
#pragma unroll
    for (int i = 0; i < MYB; ++i) {
      if (!have[i]) continue;
      const f32x4 yv = pre_ey[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gq = v[i][j];
        const float z = a.epi_scale ? fmaf(yv[j], esc[j], esh[j]) : yv[j];
        float gz = gq, q2 = 0.f;
        if (a.epi_act) {
          q2 = gq * fminf(z, 0.f);
          gz = z > 0.f ? gq : gq * eslope;
        }
        q[0][j] += (double)gz;
        q[1][j] += (double)(gz * yv[j]);
        q[2][j] += (double)q2;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) q[k][j] = reduce16_d(q[k][j]);
    __syncthreads();
    if (lp16 == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sbw[wave][k][4 * lj + j] = q[k][j];
    }
    __syncthreads();
    if (tid < 48) {
      const int k = tid >> 4, c = tid & 15;
      const double t = ((sbw[0][k][c] + sbw[1][k][c]) + sbw[2][k][c]) + sbw[3][k][c];
      if (a.epi_partial) a.epi_partial[((size_t)mt * 3 + k) * a.Cout + g * 16 + c] = (float)t;
      if (ba.bw_st_acc) __builtin_amdgcn_global_atomic_fadd_f64(ba.bw_st_acc + ((size_t)(b % ba.nrep) * a.Cout + g * 16 + c) * 4 + k, t);
    }
  }
}

inline size_t band_lds_bytes(int R, int W, int NB) {
  const size_t patch = (size_t)4 * (R + 2) * (W + 2) * PSTR, red = (size_t)4 * NB * 256;
  return sizeof(float) * (patch > red ? patch : red);
}

}  // namespace

// Rows per band for this conv shape, or 0 when the band kernel does not apply (the caller then uses conv_fwd_kernel).
// Pure function of the shape (plus the SST_CONV_BAND dev override), so that sst_conv_stat_tiles can predict the dispatch.
int sst_conv_band_rows(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (ksize != 3 || stride != 1 || !band_eligible(Cout, Cin, 9) || W < 8) return 0;
  int want = 3;   // measured on the bench step: 768 bands of 48 px (3 workgroups per CU) beat 256 bands of 144 px by 3.6 %
  if (const char* e = sst_env("SST_CONV_BAND")) {
    want = atoi(e);
    if (want == 0) return 0;
  }
  // NB = R*W/16 must be one of the compiled block counts; prefer `want`, then the other
  const int cand[2] = {want == 9 ? 9 : 3, want == 9 ? 3 : 9};
  for (int k = 0; k < 2; ++k) {
    const int px = cand[k] * 16;
    if (px % W) continue;
    const int R = px / W;
    if (R <= 0 || H % R) continue;
    if (band_lds_bytes(R, W, cand[k]) > 128 * 1024) continue;
    return R;
  }
  return 0;
}

static long g_band_launches = 0;
SST_API long sst_debug_band_launches(void) { return g_band_launches; }   // test hook: how often the band kernel was chosen

int sst_launch_conv_band(const Conv3Args& a, int R, hipStream_t st, const BandAcc* acc) {
  ++g_band_launches;
  BandAcc ba{};
  if (acc) ba = *acc;
  const int NB = R * a.W / 16, nbands = a.H / R;
  dim3 grid((unsigned)(a.B * nbands), a.Cout / 16);
  const size_t lds = band_lds_bytes(R, a.W, NB);
  static bool big_lds_enabled = false;     // > 64 KB of dynamic LDS needs the opt-in (gfx950 has 160 KB per CU)
  if (!big_lds_enabled) {
    const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band_kernel<3, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e0 != hipSuccess) return sst_set_error(SST_ERR_HIP, "conv_band: cannot raise the LDS limit");
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band_kernel<9, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_band_kernel<3, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e1 != hipSuccess || e2 != hipSuccess) return sst_set_error(SST_ERR_HIP, "conv_band: cannot raise the LDS limit");
    big_lds_enabled = true;
  }
  if (NB == 9)
    conv_band_kernel<9, true><<<grid, CONV_NT, lds, st>>>(a, R, nbands, ba);
  else if (NB == 3)
    if (a.in2 || a.epi_partial || ba.bw_in_acc || ba.bw_st_acc)
      conv_band_kernel<3, true><<<grid, CONV_NT, lds, st>>>(a, R, nbands, ba);
    else
      conv_band_kernel<3, false><<<grid, CONV_NT, lds, st>>>(a, R, nbands, ba);
  else
    return sst_set_error(SST_ERR_UNSUPPORTED, "conv_band: NB=%d not built", NB);
  SST_LAUNCH_CHECK("conv_band_kernel");
  return SST_OK;
}

// BatchNorm (train mode) statistics / backward reductions and the elementwise glue of the
// generator / discriminator graphs, NHWC fp32 viewed as [R = B*H*W rows, C channels].
//
// Replaces (reference file:line)
//   nn.BatchNorm2d in train mode        model.py:36-57,114,174,177  (eps 1e-5, momentum 0.1, unbiased running_var)
//   nn.PReLU / nn.LeakyReLU backward    model.py:102,161,175 / 33-58
//   residual adds                       model.py:146,183
// All reductions are two-stage with a fixed order (no atomics): bit-reproducible.
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------
// Forward finalize: per-tile (sum, M2, count) partials of a conv output -> batch mean / rstd, the
// fused affine (scale = gamma*rstd, shift = beta - mean*scale) and the running-stat update.
// One workgroup per channel.
// groups > 1: the tiles are `groups` equal consecutive ranges (several passes of the network batched as one tall image): each range
// gets its own statistics / affine (outputs [groups][C]) and the running statistics take one momentum step per range, in order.
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ cnt,
                                                         int ntiles, int C, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ run_mean,
                                                         float* __restrict__ run_var, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, float* __restrict__ scale,
                                                         float* __restrict__ shift, float eps, float momentum, int groups) {
  __shared__ float red[2][NT / 64];
  const int c = blockIdx.x;
  for (int grp = 0; grp < groups; ++grp, stats += (size_t)ntiles * 2 * C, cnt += ntiles, mean_out += C, rstd_out += C, scale += C,
           shift += C) {
  // One global round trip: a thread keeps its (up to KT) tiles' (sum, M2, count) in registers for both passes of Chan's
  // combination, and the two first-pass sums share one pair of barriers (this kernel is pure latency: ~1 us of work
  // behind a ~2.5 us launch, 33 times per step).
  constexpr int KT = 4;
  float ts[KT], tm[KT], tn[KT];
  float s = 0.f, n = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int t = threadIdx.x + k * NT;
    ts[k] = tm[k] = tn[k] = 0.f;
    if (t < ntiles) {
      ts[k] = stats[(size_t)t * 2 * C + c];
      tm[k] = stats[(size_t)t * 2 * C + C + c];
      tn[k] = cnt[t];
    }
    s += ts[k];
    n += tn[k];
  }
  for (int t = threadIdx.x + KT * NT; t < ntiles; t += NT) {      // more than KT*NT tiles: plain loop for the rest
    s += stats[(size_t)t * 2 * C + c];
    n += cnt[t];
  }
  {
    s = wave_sum(s);
    n = wave_sum(n);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wid] = s; red[1][wid] = n; }
    __syncthreads();
    s = n = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { s += red[0][i]; n += red[1][i]; }
    __syncthreads();
  }
  const float mean = s / n;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (tn[k] > 0.f) {
      const float d = ts[k] / tn[k] - mean;
      m2 += tm[k] + tn[k] * d * d;                                  // Chan et al. parallel variance
    }
  }
  for (int t = threadIdx.x + KT * NT; t < ntiles; t += NT) {
    const float nt = cnt[t];
    const float d = stats[(size_t)t * 2 * C + c] / nt - mean;
    m2 += stats[(size_t)t * 2 * C + C + c] + nt * d * d;
  }
  m2 = block_sum<NT>(m2, red[0]);
  if (threadIdx.x == 0) {
    const float var = m2 / n;
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (m2 / fmaxf(n - 1.f, 1.f));
    }
  }
  __syncthreads();          // red[] is reused by the next group
  }
}

// Forward finalize from fp64 accumulators acc[nrep][C][2] = (sum, sum of squares in Chan form), the layout the band conv
// kernel adds into in accumulator mode (conv_epilogue.h: BandAcc) - for BatchNorm layers whose consumer is not a band conv.
__global__ void bn_finalize_acc_kernel(const double* __restrict__ acc, int nrep, int C, float nf, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
                                       float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ scale,
                                       float* __restrict__ shift, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double A = 0.0, Bq = 0.0;
  for (int r = 0; r < nrep; ++r) {
    A += acc[((size_t)r * C + c) * 2];
    Bq += acc[((size_t)r * C + c) * 2 + 1];
  }
  const double n = (double)nf, mean = A / n;
  double m2 = Bq - A * mean;
  if (m2 < 0.0) m2 = 0.0;
  const float rstd = 1.f / sqrtf((float)(m2 / n) + eps);
  const float sc = gamma[c] * rstd;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(m2 / fmax(n - 1.0, 1.0));
  }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                      float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(run_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - run_mean[c] * sc;
  }
}

// ---------------------------------------------------------------------------------------------
// out = y*scale + shift + act(res)        (residual add after a BatchNorm; act = optional slope activation)
__global__ __launch_bounds__(NT) void bn_residual_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res,
                                                         const float* __restrict__ res_slope, float* __restrict__ out,
                                                         int64_t R, int C) {
  const int c4n = C >> 2;
  const int64_t total = R * c4n;
  const float sl = res_slope ? res_slope[0] : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int c = (int)(i % c4n) * 4;
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float rr = r[j];
      if (res_slope) rr = rr > 0.f ? rr : rr * sl;
      o[j] = fmaf(v[j], scale[c + j], shift[c + j]) + rr;
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// Arguments of the backward finalize (bwd_finalize2_kernel below).
struct FinArgs {
  const float* mean; const float* rstd; const float* gamma;
  float* dgamma; float* dbeta; float* cA; float* cB; float* cC; float* dslope;
  float n; int accumulate;
  int groups;        // > 1: nblk partial blocks PER GROUP, consecutive; mean / rstd / cA / cB / cC are [groups][C], n counts one group's
                     // elements; dgamma / dbeta / dslope sum over the groups (group order)
};

// ---------------------------------------------------------------------------------------------
// Backward reductions over rows.  With z = y*scale+shift (or z = y when scale == null) and
// gz = g * (act ? (z > 0 ? 1 : slope) : 1):
//   partial[blk][0][c] = sum gz          partial[blk][1][c] = sum gz*y        partial[blk][2][c] = sum g*min(z,0)
// Thread layout: C/4 float4-columns x (NT / (C/4)) row lanes.
__global__ __launch_bounds__(NT) void bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ g2,
                                                        const float* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ slope_p,
                                                        float slope_c, int act, float* __restrict__ partial, int64_t R,
                                                        int C, int rows_per_block) {
  // blockIdx.y = coefficient group (passes batched as one tall tensor): R rows per group, own scale / shift row, own partial blocks
  {
    const size_t go = (size_t)blockIdx.y * R * C;
    g += go;
    y += go;
    if (g2) g2 += go;
    if (scale) {
      scale += blockIdx.y * C;
      shift += blockIdx.y * C;
    }
    partial += (size_t)blockIdx.y * gridDim.x * 3 * C;
  }
  // The three sums are signed and largely cancel (the BatchNorm gradients and, summed over channels, the scalar PReLU-slope
  // gradient): every accumulation is fp64 like the reference's CPU path (acc_type<float> = double), only the block partial is
  // rounded to fp32 once.  Step time unchanged (5.80 ms before and after, the kernel is bound by its loads).
  extern __shared__ double smd[];  // [rowlanes][3][C]
  const int c4n = C >> 2;
  const int rowlanes = NT / c4n;
  const int cl = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  const int c = cl * 4;
  const float slope = slope_p ? slope_p[0] : slope_c;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
  if (scale) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = scale[c + j];
      sh[j] = shift[c + j];
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(R, r0 + rows_per_block);
  if (rl < rowlanes) {
    auto accumulate = [&](const f32x4& gv, const f32x4& yv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = fmaf(yv[j], sc[j], sh[j]);
        float gz = gv[j];
        if (act) {
          s2[j] += (double)(gz * fminf(z, 0.f));
          gz = z > 0.f ? gz : gz * slope;
        }
        s0[j] += (double)gz;
        s1[j] += (double)(gz * yv[j]);
      }
    };
    int64_t r = r0 + rl;
    constexpr int U = 4;             // rows in flight per thread (one row at a time left the kernel latency-bound: 1.3 TB/s)
    for (; r + (int64_t)(U - 1) * rowlanes < r1; r += (int64_t)U * rowlanes) {
      f32x4 gv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t o = (r + (int64_t)u * rowlanes) * c4n + cl;
        gv[u] = reinterpret_cast<const f32x4*>(g)[o];
        yv[u] = reinterpret_cast<const f32x4*>(y)[o];
        if (g2) gv[u] += reinterpret_cast<const f32x4*>(g2)[o];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(gv[u], yv[u]);      // same row order as the rolled loop: same sums
    }
    for (; r < r1; r += rowlanes) {
      f32x4 gv = reinterpret_cast<const f32x4*>(g)[r * c4n + cl];
      if (g2) gv += reinterpret_cast<const f32x4*>(g2)[r * c4n + cl];
      accumulate(gv, reinterpret_cast<const f32x4*>(y)[r * c4n + cl]);
    }
    double* d = smd + (size_t)rl * 3 * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[c + j] = s0[j];
      d[C + c + j] = s1[j];
      d[2 * C + c + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += NT) {
    double t = 0.0;
    for (int q = 0; q < rowlanes; ++q) t += smd[(size_t)q * 3 * C + i];
    partial[(size_t)blockIdx.x * 3 * C + i] = (float)t;
  }
}

// Finalize of the above.  If mean != null (BatchNorm): writes dgamma, dbeta and the coefficients of
//   dy = cA[c]*gz + cB[c]*y + cC[c]          (gz as defined above)
// else (no BN: bias-only layer): writes dbias = sum gz.   dslope (scalar) += sum_c partial[2] when dslope != null.
// Channel-parallel finalize: one workgroup of 1024 threads per 64 channels (or ONE workgroup looping over all channels
// when the scalar slope gradient is wanted, so that its sum over channels stays in one fixed-order reduction).
// Thread (c = tid & 63, q = tid >> 6): 16 lanes per channel split the nblk partials (coalesced along c, 4-5 loads deep).
constexpr int F2T = 1024;
__global__ __launch_bounds__(F2T) void bwd_finalize2_kernel(const float* __restrict__ partial, int nblk, int C, FinArgs f,
                                                            float* __restrict__ scratch, unsigned* __restrict__ counter) {
  __shared__ double sm[3][F2T];
  __shared__ float red[F2T / 64];
  __shared__ unsigned s_last;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  float al = 0.f;
  const int cstep = gridDim.x * 64;
  for (int cb = blockIdx.x * 64; cb < C; cb += cstep) {
    const int c = cb + cl;
    float dg_tot = 0.f, db_tot = 0.f;
    for (int grp = 0; grp < f.groups; ++grp) {
    // the partials are summed in fp64 and sum gz*(y - mean) = S1 - mean*S0 is evaluated in fp64: it cancels, and what fp32 loses
    // there comes back as a common-mode error of dy over the whole channel (see conv_band.hip, same arithmetic)
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    if (c < C) {
#pragma unroll 12
      for (int b = q; b < nblk; b += F2T / 64) {
        const float* p = partial + ((size_t)grp * nblk + b) * 3 * C + c;
        d0 += (double)p[0];
        d1 += (double)p[C];
        d2 += (double)p[2 * C];
      }
    }
    __syncthreads();
    sm[0][threadIdx.x] = d0;
    sm[1][threadIdx.x] = d1;
    sm[2][threadIdx.x] = d2;
    __syncthreads();
    if (q == 0 && c < C) {
      d0 = d1 = d2 = 0.0;
#pragma unroll
      for (int i = 0; i < F2T / 64; ++i) {
        d0 += sm[0][cl + 64 * i];
        d1 += sm[1][cl + 64 * i];
        d2 += sm[2][cl + 64 * i];
      }
      const float s0 = (float)d0, s2 = (float)d2;
      al += s2;
      if (f.mean) {
        const int gc = grp * C + c;
        const float mu = f.mean[gc], rs = f.rstd[gc], ga = f.gamma[c];
        const double sgh_d = (double)rs * (d1 - (double)mu * d0);
        const float sgh = (float)sgh_d;
        const float m1 = (float)(d0 / (double)f.n), m2 = (float)(sgh_d / (double)f.n);
        dg_tot = grp ? dg_tot + sgh : sgh;
        db_tot = grp ? db_tot + s0 : s0;
        const float a = ga * rs;
        f.cA[gc] = a;
        f.cB[gc] = -a * rs * m2;
        f.cC[gc] = -a * m1 + a * rs * mu * m2;
      } else {
        db_tot = grp ? db_tot + s0 : s0;
      }
    }
    }   // groups
    if (q == 0 && c < C) {
      if (f.mean) {
        if (f.accumulate) {
          f.dgamma[c] += dg_tot;
          f.dbeta[c] += db_tot;
        } else {
          f.dgamma[c] = dg_tot;
          f.dbeta[c] = db_tot;
        }
      } else if (f.dbeta) {
        if (f.accumulate) f.dbeta[c] += db_tot; else f.dbeta[c] = db_tot;
      }
    }
  }
  if (f.dslope) {
    al = block_sum<F2T>(al, red);
    if (gridDim.x == 1) {
      if (threadIdx.x == 0) {
        if (f.accumulate) f.dslope[0] += al; else f.dslope[0] = al;
      }
    } else {
      // several workgroups (scratch / counter given): each publishes the sum over its channels, the last arriver adds them in
      // workgroup order.  The hand-off is cheap HERE: this kernel has next to no dirty L2 lines for the release to write back.
      if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + blockIdx.x, al, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (publish_and_ticket(counter) == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
          acquire_after_ticket();
          float t = 0.f;
          for (unsigned i = 0; i < gridDim.x; ++i) t += load_agent(scratch + i);
          if (f.accumulate) f.dslope[0] += t; else f.dslope[0] = t;
          __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// The same finalize without the scalar slope gradient (every LeakyReLU layer: the whole discriminator), laid out for latency: a
// workgroup owns 16 channels x 64 row lanes (thread: cl = tid & 15, q = tid >> 4), so a layer of C channels gets C/16 workgroups and a
// thread walks nblk/64 partial rows instead of nblk/16 (the 64-channel layers ran ONE workgroup of 16 row lanes over up to 2,304
// per-tile rows of a conv epilogue: 24-48 us of dependent loads for 0.6 MB).  The third partial (sum g*min(z,0)) is not read at all.
constexpr int F3T = 1024;
__global__ __launch_bounds__(F3T) void bwd_finalize3_kernel(const float* __restrict__ partial, int nblk, int C, FinArgs f) {
  __shared__ double sm[2][F3T / 64][16];
  const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 16 + cl;
  float dg_tot = 0.f, db_tot = 0.f;
  for (int grp = 0; grp < f.groups; ++grp) {
    double d0 = 0.0, d1 = 0.0;
    if (c < C) {
      const float* p = partial + ((size_t)grp * nblk + q) * 3 * C + c;
      const size_t stride = (size_t)64 * 3 * C;
#pragma unroll 4
      for (int b = q; b < nblk; b += 64, p += stride) {
        d0 += (double)p[0];
        d1 += (double)p[C];
      }
    }
    d0 += __shfl_xor(d0, 16, 64);
    d1 += __shfl_xor(d1, 16, 64);
    d0 += __shfl_xor(d0, 32, 64);
    d1 += __shfl_xor(d1, 32, 64);
    __syncthreads();                                   // the previous group's sums have been read
    if (lane < 16) {
      sm[0][wave][lane] = d0;
      sm[1][wave][lane] = d1;
    }
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
      d0 = d1 = 0.0;
#pragma unroll
      for (int i = 0; i < F3T / 64; ++i) {
        d0 += sm[0][i][cl];
        d1 += sm[1][i][cl];
      }
      const float s0 = (float)d0;
      if (f.mean) {
        // sum gz*(y - mean) = S1 - mean*S0 in fp64: it cancels (see bwd_finalize2_kernel)
        const int gc = grp * C + c;
        const float mu = f.mean[gc], rs = f.rstd[gc], ga = f.gamma[c];
        const double sgh_d = (double)rs * (d1 - (double)mu * d0);
        const float sgh = (float)sgh_d;
        const float m1 = (float)(d0 / (double)f.n), m2 = (float)(sgh_d / (double)f.n);
        dg_tot = grp ? dg_tot + sgh : sgh;
        db_tot = grp ? db_tot + s0 : s0;
        const float a = ga * rs;
        f.cA[gc] = a;
        f.cB[gc] = -a * rs * m2;
        f.cC[gc] = -a * m1 + a * rs * mu * m2;
      } else {
        db_tot = grp ? db_tot + s0 : s0;
      }
    }
  }
  if (threadIdx.x < 16 && c < C) {
    if (f.mean) {
      if (f.accumulate) {
        f.dgamma[c] += dg_tot;
        f.dbeta[c] += db_tot;
      } else {
        f.dgamma[c] = dg_tot;
        f.dbeta[c] = db_tot;
      }
    } else if (f.dbeta) {
      if (f.accumulate) f.dbeta[c] += db_tot; else f.dbeta[c] = db_tot;
    }
  }
}

// dy = cA*gz + cB*y + cC  (BatchNorm input gradient), or dy = gz when cA == null (activation only).
__global__ __launch_bounds__(NT) void bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ g2,
                                                       const float* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ slope_p,
                                                       float slope_c, int act, const float* __restrict__ cA,
                                                       const float* __restrict__ cB, const float* __restrict__ cC,
                                                       float* __restrict__ dy, int64_t R, int C, int uH, int uW) {
  // blockIdx.y = coefficient group (passes batched as one tall tensor): R rows per group, own coefficient rows
  {
    const size_t go = (size_t)blockIdx.y * R * C;
    g += go;
    y += go;
    dy += go;
    if (g2) g2 += go;
    if (scale) {
      scale += blockIdx.y * C;
      shift += blockIdx.y * C;
    }
    if (cA) {
      cA += blockIdx.y * C;
      cB += blockIdx.y * C;
      cC += blockIdx.y * C;
    }
  }
  const int c4n = C >> 2;
  const int64_t total = R * c4n;
  const float slope = slope_p ? slope_p[0] : slope_c;
  const int64_t stride = (int64_t)gridDim.x * NT;
  constexpr int U = 4;               // items in flight per thread (loads of a batch issued before the first is used)
  // The grid stride is a multiple of the channel-quad count for every layer of the path (power-of-two channels): a thread then
  // keeps ONE channel quad, and its five coefficient quads (scale, shift, cA, cB, cC) are loaded once instead of 20 scalar
  // loads per item - the kernel was bound by issuing those, not by its 48 B of streaming traffic per item.
  const bool fixed_c = stride % c4n == 0;
  f32x4 hsc = {1.f, 1.f, 1.f, 1.f}, hsh = {0.f, 0.f, 0.f, 0.f}, hA = {1.f, 1.f, 1.f, 1.f}, hB = {0.f, 0.f, 0.f, 0.f}, hC = {0.f, 0.f, 0.f, 0.f};
  if (fixed_c) {
    const int c = (int)((blockIdx.x * (int64_t)NT + threadIdx.x) % c4n) * 4;
    if (scale) {
      hsc = *reinterpret_cast<const f32x4*>(scale + c);
      hsh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    if (cA) {
      hA = *reinterpret_cast<const f32x4*>(cA + c);
      hB = *reinterpret_cast<const f32x4*>(cB + c);
      hC = *reinterpret_cast<const f32x4*>(cC + c);
    }
  }
  for (int64_t i0 = blockIdx.x * (int64_t)NT + threadIdx.x; i0 < total; i0 += stride * U) {
    f32x4 gvs[U], yvs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      const int64_t il = i < total ? i : i0;              // past the end: re-read a valid item, result dropped
      gvs[u] = reinterpret_cast<const f32x4*>(g)[il];
      yvs[u] = reinterpret_cast<const f32x4*>(y)[il];
      if (g2) gvs[u] += reinterpret_cast<const f32x4*>(g2)[il];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + u * stride;
    if (i >= total) break;
    const int c = (int)(i % c4n) * 4;
    const f32x4 gv = gvs[u];
    const f32x4 yv = yvs[u];
    f32x4 o;
    f32x4 ksc = hsc, ksh = hsh, kA = hA, kB = hB, kC = hC;
    if (!fixed_c) {
      if (scale) {
        ksc = *reinterpret_cast<const f32x4*>(scale + c);
        ksh = *reinterpret_cast<const f32x4*>(shift + c);
      }
      if (cA) {
        kA = *reinterpret_cast<const f32x4*>(cA + c);
        kB = *reinterpret_cast<const f32x4*>(cB + c);
        kC = *reinterpret_cast<const f32x4*>(cC + c);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gz = gv[j];
      if (act) {
        const float z = scale ? fmaf(yv[j], ksc[j], ksh[j]) : yv[j];
        gz = z > 0.f ? gz : gz * slope;
      }
      o[j] = cA ? fmaf(kA[j], gz, fmaf(kB[j], yv[j], kC[j])) : gz;
    }
    if (uW == 0) {
      reinterpret_cast<f32x4*>(dy)[i] = o;
    } else {  // inverse PixelShuffle(2) store: rows are pixels (b,Y,X) of the [B,uH,uW,C] shuffled tensor
      const int64_t r = i / c4n;
      const int X = (int)(r % uW);
      const int64_t t = r / uW;
      const int Y = (int)(t % uH);
      const int64_t b = t / uH;
      float* d = dy + (((b * (uH >> 1) + (Y >> 1)) * (uW >> 1) + (X >> 1)) * (int64_t)(4 * C)) + 2 * (Y & 1) + (X & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[4 * (c + j)] = o[j];
    }
    }
  }
}

// out[i] = a[i] + b[i]
__global__ __launch_bounds__(NT) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + NT - 1) / NT;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_bn_finalize_grp(const float* stats, const float* cnt, int ntiles, int C, int groups, const float* gamma,
                                const float* beta, float* run_mean, float* run_var, float* mean, float* rstd, float* scale,
                                float* shift, float eps, float momentum, void* stream);
SST_API int sst_bn_finalize(const float* stats, const float* cnt, int ntiles, int C, const float* gamma, const float* beta,
                            float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift,
                            float eps, float momentum, void* stream) {
  SST_REQUIRE(stats && cnt && gamma && beta && mean && rstd && scale && shift && ntiles > 0 && C > 0,
              "sst_bn_finalize: bad argument");
  SST_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "sst_bn_finalize: running stats must come together");
  return sst_bn_finalize_grp(stats, cnt, ntiles, C, 1, gamma, beta, run_mean, run_var, mean, rstd, scale, shift, eps, momentum, stream);
}

// The same for `groups` passes batched as one tall image: stats / cnt hold ntiles tiles = groups equal consecutive ranges; mean / rstd /
// scale / shift are [groups][C]; the running statistics take one momentum step per group, in group order (the order the reference runs
// the passes in, train.py:155-158).
SST_API int sst_bn_finalize_grp(const float* stats, const float* cnt, int ntiles, int C, int groups, const float* gamma,
                                const float* beta, float* run_mean, float* run_var, float* mean, float* rstd, float* scale,
                                float* shift, float eps, float momentum, void* stream) {
  SST_REQUIRE(stats && cnt && gamma && beta && mean && rstd && scale && shift && ntiles > 0 && C > 0 && groups > 0 && ntiles % groups == 0,
              "sst_bn_finalize_grp: bad argument (ntiles=%d groups=%d)", ntiles, groups);
  SST_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "sst_bn_finalize_grp: running stats must come together");
  // (A 16-channels-per-workgroup form of this kernel - 1,024 threads, coalesced 64-B reads - measured SLOWER inside the step, 10.0 vs
  // 6.8 us per launch, and taught something else on the way: its first version spilled registers to scratch memory and then returned
  // wrong, run-to-run different statistics inside the two-branch hipGraph while being exact in eager launches: DESIGN.md section 5,
  // tests/test_abi_symbols.py.)
  bn_finalize_kernel<<<C, NT, 0, sst_stream(stream)>>>(stats, cnt, ntiles / groups, C, gamma, beta, run_mean, run_var, mean, rstd,
                                                        scale, shift, eps, momentum, groups);
  SST_LAUNCH_CHECK("bn_finalize_kernel");
  return SST_OK;
}

SST_API int sst_bn_finalize_acc(const double* acc, int nrep, int C, float n, const float* gamma, const float* beta, float* run_mean,
                                float* run_var, float* mean, float* rstd, float* scale, float* shift, float eps, float momentum,
                                void* stream) {
  SST_REQUIRE(acc && nrep > 0 && C > 0 && n > 0.f && gamma && beta && mean && rstd && scale && shift, "sst_bn_finalize_acc: bad argument");
  bn_finalize_acc_kernel<<<(C + 63) / 64, 64, 0, sst_stream(stream)>>>(acc, nrep, C, n, gamma, beta, run_mean, run_var, mean, rstd, scale,
                                                                       shift, eps, momentum);
  SST_LAUNCH_CHECK("bn_finalize_acc_kernel");
  return SST_OK;
}

SST_API int sst_bn_eval_affine(const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                               float* scale, float* shift, int C, float eps, void* stream) {
  SST_REQUIRE(gamma && beta && run_mean && run_var && scale && shift && C > 0, "sst_bn_eval_affine: bad argument");
  bn_eval_affine_kernel<<<(C + 63) / 64, 64, 0, sst_stream(stream)>>>(gamma, beta, run_mean, run_var, scale, shift, C, eps);
  SST_LAUNCH_CHECK("bn_eval_affine_kernel");
  return SST_OK;
}

SST_API int sst_bn_residual(const float* y, const float* scale, const float* shift, const float* res,
                            const float* res_slope, float* out, int64_t R, int C, void* stream) {
  SST_REQUIRE(y && scale && shift && res && out && R > 0 && C > 0 && (C & 3) == 0, "sst_bn_residual: bad argument");
  bn_residual_kernel<<<grid_for(R * (C / 4)), NT, 0, sst_stream(stream)>>>(y, scale, shift, res, res_slope, out, R, C);
  SST_LAUNCH_CHECK("bn_residual_kernel");
  return SST_OK;
}

SST_API int sst_bwd_reduce_blocks(int64_t R, int C) {
  // ~8 float4 per thread, at most 256 workgroups (the finalize kernel walks nblk partials per channel; 128 -> 256: -0.3 % srgan step)
  int64_t nb = (R * (C / 4) + 256 * 8 - 1) / (256 * 8);
  if (nb > 256) nb = 256;
  if (nb > R) nb = R;
  return (int)(nb < 1 ? 1 : nb);
}

static int launch_bwd_reduce(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                             const float* slope, float slope_const, int act, float* partial, int64_t R, int C,
                             void* stream, int groups = 1) {
  SST_REQUIRE(g && y && partial && R > 0 && C >= 4 && (C & 3) == 0 && C <= 1024, "sst_bwd_reduce: bad argument (C=%d)", C);
  SST_REQUIRE(NT % (C / 4) == 0 || C / 4 > NT, "sst_bwd_reduce: C/4 must divide %d", NT);
  SST_REQUIRE(C / 4 <= NT, "sst_bwd_reduce: C too large");
  const int nblk = sst_bwd_reduce_blocks(R, C);
  const int rpb = (int)((R + nblk - 1) / nblk);
  const int rowlanes = NT / (C / 4);
  const size_t smem = (size_t)rowlanes * 3 * C * sizeof(double);
  bwd_reduce_kernel<<<dim3(nblk, groups), NT, smem, sst_stream(stream)>>>(g, g2, y, scale, shift, slope, slope_const, act, partial, R, C, rpb);
  SST_LAUNCH_CHECK("bwd_reduce_kernel");
  return SST_OK;
}

SST_API int sst_bwd_reduce(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                           const float* slope, float slope_const, int act, float* partial, int64_t R, int C,
                           void* stream) {
  return launch_bwd_reduce(g, g2, y, scale, shift, slope, slope_const, act, partial, R, C, stream);
}

// ---- the same three steps for `groups` passes batched as one tall tensor [groups * R rows][C] (R rows per group): scale / shift / mean /
// rstd / cA / cB / cC are [groups][C] (each pass has its own batch statistics), partial is [groups][sst_bwd_reduce_blocks(R, C)][3][C],
// n counts ONE group's elements per channel; dgamma / dbeta (parameters are shared by the passes) sum over the groups.
SST_API int sst_bwd_reduce_grp(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                               const float* slope, float slope_const, int act, float* partial, int64_t R, int C, int groups,
                               void* stream) {
  SST_REQUIRE(groups > 0 && groups <= 65535, "sst_bwd_reduce_grp: bad group count %d", groups);
  return launch_bwd_reduce(g, g2, y, scale, shift, slope, slope_const, act, partial, R, C, stream, groups);
}

SST_API int sst_bwd_finalize_grp(const float* partial, int nblk, int C, float n, int groups, const float* mean, const float* rstd,
                                 const float* gamma, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int accumulate,
                                 void* stream) {
  SST_REQUIRE(partial && nblk > 0 && C > 0 && groups > 0, "sst_bwd_finalize_grp: bad argument");
  SST_REQUIRE(!mean || (rstd && gamma && dgamma && dbeta && cA && cB && cC), "sst_bwd_finalize_grp: BN mode needs all BN pointers");
  FinArgs fin = {mean, rstd, gamma, dgamma, dbeta, cA, cB, cC, nullptr, n, accumulate, groups};
  bwd_finalize3_kernel<<<(C + 15) / 16, F3T, 0, sst_stream(stream)>>>(partial, nblk, C, fin);
  SST_LAUNCH_CHECK("bwd_finalize_kernel (groups)");
  return SST_OK;
}

SST_API int sst_bwd_apply_grp(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                              const float* slope, float slope_const, int act, const float* cA, const float* cB, const float* cC,
                              float* dy, int64_t R, int C, int groups, void* stream) {
  SST_REQUIRE(g && y && dy && R > 0 && C > 0 && (C & 3) == 0 && groups > 0 && groups <= 65535, "sst_bwd_apply_grp: bad argument");
  bwd_apply_kernel<<<dim3(grid_for(R * (C / 4)), groups), NT, 0, sst_stream(stream)>>>(g, g2, y, scale, shift, slope, slope_const, act,
                                                                                      cA, cB, cC, dy, R, C, 0, 0);
  SST_LAUNCH_CHECK("bwd_apply_kernel (groups)");
  return SST_OK;
}

SST_API int sst_bwd_finalize(const float* partial, int nblk, int C, float n, const float* mean, const float* rstd,
                             const float* gamma, float* dgamma, float* dbeta, float* cA, float* cB, float* cC,
                             float* dslope, int accumulate, void* stream) {
  SST_REQUIRE(partial && nblk > 0 && C > 0, "sst_bwd_finalize: bad argument");
  SST_REQUIRE(!mean || (rstd && gamma && dgamma && dbeta && cA && cB && cC), "sst_bwd_finalize: BN mode needs all BN pointers");
  FinArgs fin = {mean, rstd, gamma, dgamma, dbeta, cA, cB, cC, dslope, n, accumulate, 1};
  if (dslope) bwd_finalize2_kernel<<<1, F2T, 0, sst_stream(stream)>>>(partial, nblk, C, fin, nullptr, nullptr);
  else bwd_finalize3_kernel<<<(C + 15) / 16, F3T, 0, sst_stream(stream)>>>(partial, nblk, C, fin);
  SST_LAUNCH_CHECK("bwd_finalize_kernel");
  return SST_OK;
}

// Same, channel-parallel (one workgroup per 64 channels) also when the scalar slope gradient is wanted: scratch = (C+63)/64
// floats, counter = one zeroed 32-bit word (left zero again).  For wide layers (the up-sampler's 256 channels).
SST_API int sst_bwd_finalize_wide(const float* partial, int nblk, int C, float n, const float* mean, const float* rstd,
                                  const float* gamma, float* dgamma, float* dbeta, float* cA, float* cB, float* cC,
                                  float* dslope, int accumulate, float* scratch, unsigned* counter, void* stream) {
  SST_REQUIRE(partial && nblk > 0 && C > 0 && scratch && counter, "sst_bwd_finalize_wide: bad argument");
  SST_REQUIRE(!mean || (rstd && gamma && dgamma && dbeta && cA && cB && cC), "sst_bwd_finalize_wide: BN mode needs all BN pointers");
  FinArgs fin = {mean, rstd, gamma, dgamma, dbeta, cA, cB, cC, dslope, n, accumulate, 1};
  bwd_finalize2_kernel<<<(C + 63) / 64, F2T, 0, sst_stream(stream)>>>(partial, nblk, C, fin, scratch, counter);
  SST_LAUNCH_CHECK("bwd_finalize_kernel (wide)");
  return SST_OK;
}

SST_API int sst_bwd_apply(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                          const float* slope, float slope_const, int act, const float* cA, const float* cB,
                          const float* cC, float* dy, int64_t R, int C, int unshuffle_H, int unshuffle_W, void* stream) {
  SST_REQUIRE(g && y && dy && R > 0 && C > 0 && (C & 3) == 0, "sst_bwd_apply: bad argument");
  SST_REQUIRE(unshuffle_W == 0 || ((unshuffle_H & 1) == 0 && (unshuffle_W & 1) == 0 &&
                                   R % ((int64_t)unshuffle_H * unshuffle_W) == 0),
              "sst_bwd_apply: bad unshuffle geometry");
  bwd_apply_kernel<<<grid_for(R * (C / 4)), NT, 0, sst_stream(stream)>>>(g, g2, y, scale, shift, slope, slope_const, act, cA,
                                                                        cB, cC, dy, R, C, unshuffle_H, unshuffle_W);
  SST_LAUNCH_CHECK("bwd_apply_kernel");
  return SST_OK;
}

// ---- activation-only backward with the partial sums of the bias / slope gradients in the same pass (no BatchNorm):
//   gz = act'(y) * (g + g2),  dy = gz (optionally stored in the pre-PixelShuffle layout),
//   partial[blk][0][c'] = sum gz,  partial[blk][1][c'] = 0,  partial[blk][2][c'] = sum (g+g2)*min(y,0)     over the block's rows
// (c' = channel of the STORED tensor: 4c + 2(Y&1) + (X&1) when unshuffling) - the layout sst_bwd_finalize consumes, so the
// bias gradient of the conv that produced y and the PReLU slope gradient come out of ONE finalize.  Replaces
// [bwd_reduce, finalize, bwd_apply, bwd_reduce, finalize] of the up-sampling blocks (model.py:159-161 backward).
// Unshuffle mode: a thread owns one OUTPUT pixel x one input channel quad: it reads the 4 sub-pixels (16 B each, coalesced
// along channels) and writes 16 consecutive output channels (64 B).
namespace {
constexpr int AP_ROWS = 64;   // stored rows (pixels) per workgroup

template <bool UNSH>
__global__ __launch_bounds__(NT) void act_bwd_partial_kernel(const float* __restrict__ g, const float* __restrict__ g2,
                                                             const float* __restrict__ y, const float* __restrict__ slope_p,
                                                             float slope_c, float* __restrict__ dy, float* __restrict__ partial,
                                                             int64_t Ro, int C, int oH, int oW) {
  // Ro = stored rows; C = channels of g / y.  UNSH: stored tensor is [B,oH,oW,4C], g / y are [B,2oH,2oW,C].
  constexpr int NV = UNSH ? 4 : 1;                 // f32x4 values per thread and row
  __shared__ float sm[2][NT][NV * 4 + 1];
  const int nq = C >> 2;                           // channel quads per input pixel (NT % nq == 0)
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq, npl = NT / nq;
  const float slope = slope_p ? slope_p[0] : slope_c;
  f32x4 s0[NV], s2[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) s0[k] = s2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * AP_ROWS;
  for (int rr = pl; rr < AP_ROWS; rr += npl) {
    const int64_t r = r0 + rr;
    if (r >= Ro) break;
    if (UNSH) {
      const int ox = (int)(r % oW);
      const int64_t t = r / oW;
      const int oy = (int)(t % oH);
      const int64_t b = t / oH;
      f32x4 o[4];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {             // sub-pixel (i, j) = (sp >> 1, sp & 1)
        const int64_t ip = ((b * 2 * oH + 2 * oy + (sp >> 1)) * 2 * oW + 2 * ox + (sp & 1)) * nq + q;
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[ip];
        if (g2) gv += reinterpret_cast<const f32x4*>(g2)[ip];
        const f32x4 yv = reinterpret_cast<const f32x4*>(y)[ip];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gz = yv[j] > 0.f ? gv[j] : gv[j] * slope;
          // stored channel 4*(4q + j) + sp  ->  value j of the thread's sp-th ... regroup: output quad j holds sub-pixels 0..3
          o[j][sp] = gz;
          s0[j][sp] += gz;
          s2[j][sp] = fmaf(gv[j], fminf(yv[j], 0.f), s2[j][sp]);
        }
      }
      f32x4* d = reinterpret_cast<f32x4*>(dy) + r * (int64_t)(4 * nq) + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = o[j];
    } else {
      const int64_t ip = r * nq + q;
      f32x4 gv = reinterpret_cast<const f32x4*>(g)[ip];
      if (g2) gv += reinterpret_cast<const f32x4*>(g2)[ip];
      const f32x4 yv = reinterpret_cast<const f32x4*>(y)[ip];
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gz = yv[j] > 0.f ? gv[j] : gv[j] * slope;
        o[j] = gz;
        s0[0][j] += gz;
        s2[0][j] = fmaf(gv[j], fminf(yv[j], 0.f), s2[0][j]);
      }
      reinterpret_cast<f32x4*>(dy)[ip] = o;
    }
  }
  // combine the npl row lanes (fixed order) and store the block's partial sums
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sm[0][threadIdx.x][k * 4 + j] = s0[k][j];
      sm[1][threadIdx.x][k * 4 + j] = s2[k][j];
    }
  __syncthreads();
  const int Cs = C * NV;                           // stored channels
  for (int i = threadIdx.x; i < 2 * Cs; i += NT) {
    const int which = i / Cs, c = i - which * Cs;  // stored channel c = (4*qq + k) * 4 + j  (UNSH)  or  4*qq + j
    const int qq = c / (4 * NV), e = c - qq * 4 * NV;
    float t = 0.f;
    for (int l = 0; l < npl; ++l) t += sm[which][l * nq + qq][e];
    partial[((size_t)blockIdx.x * 3 + (which ? 2 : 0)) * Cs + c] = t;
  }
  for (int c = threadIdx.x; c < Cs; c += NT) partial[((size_t)blockIdx.x * 3 + 1) * Cs + c] = 0.f;
}
}  // namespace

SST_API int sst_act_bwd_partial_blocks(int64_t stored_rows) { return (int)((stored_rows + AP_ROWS - 1) / AP_ROWS); }

// g, g2 (or null), y: [rows, C] of the activation's output/input; slope: device scalar or null (slope_const).
// unshuffle_H/W = 0: dy [rows, C], partial [blocks][3][C].  Else g / y are [B, H, W, C] (H = unshuffle_H, ...) and dy is
// [B, H/2, W/2, 4C] (inverse PixelShuffle(2)), partial [blocks][3][4C];  blocks = sst_act_bwd_partial_blocks(stored rows).
SST_API int sst_act_bwd_partial(const float* g, const float* g2, const float* y, const float* slope, float slope_const, float* dy,
                                float* partial, int64_t R, int C, int unshuffle_H, int unshuffle_W, void* stream) {
  SST_REQUIRE(g && y && dy && partial && R > 0 && C >= 4 && (C & 3) == 0 && NT % (C >> 2) == 0,
              "sst_act_bwd_partial: bad argument (C=%d must be a multiple of 4 with 256 %% (C/4) == 0)", C);
  if (unshuffle_W) {
    SST_REQUIRE((unshuffle_H & 1) == 0 && (unshuffle_W & 1) == 0 && R % ((int64_t)unshuffle_H * unshuffle_W) == 0,
                "sst_act_bwd_partial: bad unshuffle geometry");
    const int64_t Ro = R / 4;
    act_bwd_partial_kernel<true><<<sst_act_bwd_partial_blocks(Ro), NT, 0, sst_stream(stream)>>>(
        g, g2, y, slope, slope_const, dy, partial, Ro, C, unshuffle_H / 2, unshuffle_W / 2);
  } else {
    act_bwd_partial_kernel<false><<<sst_act_bwd_partial_blocks(R), NT, 0, sst_stream(stream)>>>(g, g2, y, slope, slope_const, dy,
                                                                                                 partial, R, C, 0, 0);
  }
  SST_LAUNCH_CHECK("act_bwd_partial_kernel");
  return SST_OK;
}

SST_API int sst_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  SST_REQUIRE(a && b && out && n > 0 && (n & 3) == 0, "sst_add: bad argument");
  add_kernel<<<grid_for(n / 4), NT, 0, sst_stream(stream)>>>(a, b, out, n / 4);
  SST_LAUNCH_CHECK("add_kernel");
  return SST_OK;
}


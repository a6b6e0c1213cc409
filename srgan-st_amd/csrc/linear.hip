// Fully-connected layers of the discriminator classifier (reference model.py:61-65, 69-70):
//   flatten(NCHW order) -> Linear(8C*(HR/16)^2 -> 1024) -> LeakyReLU(0.2) -> Linear(1024 -> 1)
//
// Linear-1 is HBM-bound (75.5 MB of fp32 weights for 0.6 GFLOP at B=16): every kernel here streams the
// weight matrix exactly once in its reference layout [N][K] with 16-B lanes; the arithmetic rides along on
// v_mfma_f32_16x16x4_f32 (M = batch <= 16 per tile) or plain FMAs.
//   fwd   : y[m][n]  = sum_k x[m][k] w[n][k] (+ bias)          split-K over workgroups, slab reduce
//   dgrad : dx[m][k] = sum_n dy[m][n] w[n][k]                   optional NHWC scatter of the k index
//   wgrad : dw[n][k] = sum_m dy[m][n] x[m][k]                   rank-M update, write-bound
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NT = 256;
constexpr int MAXM = 64;   // batch rows supported (4 MFMA row tiles)

// ---- forward: grid (N/16, ksplit); each wave walks its k range 16 at a time (one float4 per lane per operand)
__global__ __launch_bounds__(NT) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ slab, int M, int N, int K, int kslice) {
  __shared__ float red[4][MAXM][17];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lj = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kb = blockIdx.y * kslice, ke = min(K, kb + kslice);
  const int mt = (M + 15) / 16;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n = min(n0 + lj, N - 1);
  const float* wrow = w + (size_t)n * K;
  for (int k = kb + wave * 16; k < ke; k += 64) {
    const int kk = k + 4 * lq;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (kk + 3 < ke) {
      bv = *reinterpret_cast<const f32x4*>(wrow + kk);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (kk + j < ke) bv[j] = wrow[kk + j];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < mt) {
        const int m = t * 16 + lj;
        f32x4 av = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
          if (kk + 3 < ke) {
            av = *reinterpret_cast<const f32x4*>(x + (size_t)m * K + kk);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (kk + j < ke) av[j] = x[(size_t)m * K + kk + j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[t], 0, 0, 0);
      }
    }
  }
  // acc[t][r]: row m = 16t + 4*lq + r, col n = n0 + lj
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t * 16 + 4 * lq + r][lj] = acc[t][r];
  __syncthreads();
  for (int i = threadIdx.x; i < M * 16; i += NT) {
    const int m = i >> 4, j = i & 15;
    if (n0 + j < N)
      slab[((size_t)blockIdx.y * M + m) * N + n0 + j] = red[0][m][j] + red[1][m][j] + red[2][m][j] + red[3][m][j];
  }
}

// y[m][n] = bias[n] + sum_s slab[s][m][n]
__global__ __launch_bounds__(NT) void linear_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                           float* __restrict__ y, int nslab, int M, int N) {
  const int total = M * N;
  for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
    float t = bias ? bias[i % N] : 0.f;
    for (int s = 0; s < nslab; ++s) t += slab[(size_t)s * total + i];
    y[i] = t;
  }
}

// ---- dgrad: grid (K/64); wave w handles n in [w*N/4, (w+1)*N/4), 4 rows of W per step, 64 k-columns per workgroup.
// Lane (lj, lq) loads float4 W[n+lq][k0 + 4 lj ..]: MFMA t (t = 0..3) uses element t => column k0 + 4 lj + t.
// MT = 16-row tiles of the batch (M <= 16 MT): W is streamed ONCE for all of them (a 32-row batch - the discriminator step's two
// passes as one - went over the 75.5 MB weight twice in the first form of this kernel).
template <int MT>
__global__ __launch_bounds__(NT) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, int M, int N, int K, int C, int HW) {
  __shared__ float red[4][16][65];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lj = lane & 15, lq = lane >> 4;
  const int k0 = blockIdx.x * 64;
  const int nper = ((N + 3) / 4 + 3) / 4 * 4;
  const int nb = wave * nper, ne = min(N, nb + nper);
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int n = nb;
  if (k0 + 64 <= K) {
    // main part: U steps (4 U rows of W) per iteration with all 16-B loads issued before the first MFMA - the op streams
    // W once (75 MB for the 1024 x 18432 classifier) and a wave with ONE load in flight ran it at 1.7 TB/s
    constexpr int U = MT == 1 ? 8 : 4;
    const float* wcol = w + k0 + 4 * lj;
    for (; n + 4 * U <= ne; n += 4 * U) {
      f32x4 bv[U];
      float av[MT][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int nn = n + 4 * u + lq;
        bv[u] = *reinterpret_cast<const f32x4*>(wcol + (size_t)nn * K);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 16 + lj;
          av[mt][u] = m < M ? dy[(size_t)m * N + nn] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][u], bv[u][t], acc[mt][t], 0, 0, 0);
    }
  }
  for (; n < ne; n += 4) {
    const int nn = n + lq;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (nn < ne) {
      const int kk = k0 + 4 * lj;
      if (kk + 3 < K) {
        bv = *reinterpret_cast<const f32x4*>(w + (size_t)nn * K + kk);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kk + j < K) bv[j] = w[(size_t)nn * K + kk + j];
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + lj;
      const float av = (nn < ne && m < M) ? dy[(size_t)m * N + nn] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[mt][t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * lq + r][4 * lj + t] = acc[mt][t][r];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += NT) {
      const int mm = i >> 6, kc = i & 63, mg = mt * 16 + mm, k = k0 + kc;
      if (mg < M && k < K) {
        const float v = red[0][mm][kc] + red[1][mm][kc] + red[2][mm][kc] + red[3][mm][kc];
        // k indexes the NCHW flatten (c, hw); HW > 0 scatters to NHWC [M][HW][C]
        const size_t o = HW > 0 ? (size_t)mg * K + (size_t)(k % HW) * C + k / HW : (size_t)mg * K + k;
        dx[o] = v;
      }
    }
  }
}

// ---- wgrad: grid (K/1024, N/16); thread owns 4 consecutive k; dw[n][k] (+)= sum_m dy[m][n] x[m][k]
__global__ __launch_bounds__(NT) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dw, int M, int N, int K, int accumulate) {
  __shared__ float sdy[MAXM][16];
  const int n0 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < M * 16; i += NT) {
    const int m = i >> 4, j = i & 15;
    sdy[m][j] = (n0 + j < N) ? dy[(size_t)m * N + n0 + j] : 0.f;
  }
  __syncthreads();
  const int k = (blockIdx.x * NT + threadIdx.x) * 4;
  if (k >= K) return;
  const bool vec = (K & 3) == 0;
  f32x4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < M; ++m) {
    f32x4 xv = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * K + k);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < K) xv[j] = x[(size_t)m * K + k + j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] += sdy[m][j] * xv;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (n0 + j >= N) break;
    float* d = dw + (size_t)(n0 + j) * K + k;
    if (vec) {
      f32x4 o = acc[j];
      if (accumulate) o += *reinterpret_cast<f32x4*>(d);
      *reinterpret_cast<f32x4*>(d) = o;
    } else {
      for (int q = 0; q < 4; ++q)
        if (k + q < K) d[q] = accumulate ? d[q] + acc[j][q] : acc[j][q];
    }
  }
}

// ---- wgrad on the matrix cores (K % 32 == 0, N % 64 == 0): dw = dy^T x as a GEMM with the batch as its (short) K dimension.  A wave owns
// a 64 (n) x 32 (k) block of dw: per pair of batch rows one dword of x per lane (128-B row segments) feeds two
// v_mfma_f32_32x32x2_f32, the block leaves as 128-B row segments of dw - the kernel streams dw once (write-bound); the FMA form above
// spends 16 LDS reads per 64 FMAs per batch row and took 41 us at 32 rows (28 at 16).  Sum order over m: the MFMA's (pairs in order).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(NT) void linear_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dw, int M, int N, int K, int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int k0 = (blockIdx.x * 4 + wave) * 32, n0 = blockIdx.y * 64;
  if (k0 >= K) return;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  const float* xp = x + (size_t)lh * K + k0 + li;          // x[2j + lh][k0 + li]
  const float* dp = dy + (size_t)lh * N + n0 + li;         // dy[2j + lh][n0 + li] (and + 32)
  const int steps = (M + 1) >> 1;
  for (int j0 = 0; j0 < steps; j0 += 8) {                  // 8 row pairs per batch: the loads of a batch are issued before its MFMAs
    float xv[8], d0[8], d1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (j0 + u) + lh;
      const bool ok = j0 + u < steps && m < M;
      xv[u] = ok ? xp[(size_t)2 * (j0 + u) * K] : 0.f;
      d0[u] = ok ? dp[(size_t)2 * (j0 + u) * N] : 0.f;
      d1[u] = ok ? dp[(size_t)2 * (j0 + u) * N + 32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d0[u], xv[u], acc0, 0, 0, 0);      // rows n0 .. n0+31
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[u], xv[u], acc1, 0, 0, 0);      // rows n0+32 .. n0+63
    }
  }
  // acc[r]: row (r & 3) + 8 * (r >> 2) + 4 * lh of the 32-row block, column li
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    float* d = dw + (size_t)(n0 + row) * K + k0 + li;
    float* e = d + (size_t)32 * K;
    if (accumulate) {
      *d += acc0[r];
      *e += acc1[r];
    } else {
      *d = acc0[r];
      *e = acc1[r];
    }
  }
}

// db[n] (+)= sum_m dy[m][n]
__global__ void colsum_small_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int N, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    float t = 0.f;
    for (int m = 0; m < M; ++m) t += dy[(size_t)m * N + n];
    db[n] = accumulate ? db[n] + t : t;
  }
}

// ---- the tiny head: y[m][n] = b[n] + sum_k lrelu(h[m][k]) w[n][k]   (N small); one workgroup per (m,n)
__global__ __launch_bounds__(NT) void head_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y, int M, int N, int K,
                                                      float slope) {
  __shared__ float red[NT / 64];
  const int m = blockIdx.x / N, n = blockIdx.x - m * N;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += NT) {
    float v = h[(size_t)m * K + k];
    v = v > 0.f ? v : v * slope;
    s = fmaf(v, w[(size_t)n * K + k], s);
  }
  s = block_sum<NT>(s, red);
  if (threadIdx.x == 0) y[(size_t)m * N + n] = s + (b ? b[n] : 0.f);
}

// backward of the head: dh[m][k] = lrelu'(h) * sum_n dy[m][n] w[n][k];  dw[n][k] (+)= sum_m dy[m][n] lrelu(h[m][k]);
// db[n] (+)= sum_m dy[m][n].   grid (k blocks, row blocks): a thread takes one k column and HB_ROWS batch rows for dh; the workgroups
// of the first row block also form dw over all rows (one load of h per row either way - the kernel is a few dependent loads deep
// instead of M x N of them; the one-thread-per-column form took 50 us at 32 rows).
constexpr int HB_ROWS = 8;
__global__ __launch_bounds__(NT) void head_bwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                      const float* __restrict__ dy, float* __restrict__ dh,
                                                      float* __restrict__ dw, float* __restrict__ db, int M, int N, int K,
                                                      float slope, int accumulate) {
  const int k = blockIdx.x * NT + threadIdx.x;
  const int m0 = blockIdx.y * HB_ROWS, m1 = min(M, m0 + HB_ROWS);
  if (k < K) {
    float hv[HB_ROWS];
#pragma unroll
    for (int i = 0; i < HB_ROWS; ++i) hv[i] = m0 + i < m1 ? h[(size_t)(m0 + i) * K + k] : 0.f;
    float t[HB_ROWS];
#pragma unroll
    for (int i = 0; i < HB_ROWS; ++i) t[i] = 0.f;
    for (int n = 0; n < N; ++n) {
      const float wv = w[(size_t)n * K + k];
#pragma unroll
      for (int i = 0; i < HB_ROWS; ++i)
        if (m0 + i < m1) t[i] = fmaf(dy[(size_t)(m0 + i) * N + n], wv, t[i]);
    }
#pragma unroll
    for (int i = 0; i < HB_ROWS; ++i)
      if (m0 + i < m1) dh[(size_t)(m0 + i) * K + k] = hv[i] > 0.f ? t[i] : t[i] * slope;
    if (dw && blockIdx.y == 0) {
      for (int n = 0; n < N; ++n) {
        float a = 0.f;
        for (int mb = 0; mb < M; mb += HB_ROWS) {
          float v[HB_ROWS];
#pragma unroll
          for (int i = 0; i < HB_ROWS; ++i) v[i] = mb + i < M ? h[(size_t)(mb + i) * K + k] : 0.f;      // loads of a batch in flight together
#pragma unroll
          for (int i = 0; i < HB_ROWS; ++i) {
            if (mb + i >= M) break;
            const float x = v[i] > 0.f ? v[i] : v[i] * slope;
            a = fmaf(dy[(size_t)(mb + i) * N + n], x, a);
          }
        }
        dw[(size_t)n * K + k] = accumulate ? dw[(size_t)n * K + k] + a : a;
      }
    }
  }
  if (db && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < N) {
    float t = 0.f;
    for (int m = 0; m < M; ++m) t += dy[(size_t)m * N + threadIdx.x];
    db[threadIdx.x] = accumulate ? db[threadIdx.x] + t : t;
  }
}

// flat[b][c*HW + hw] = act(y[b][hw][c] * scale[c] + shift[c])        (NHWC -> NCHW flatten, model.py:69)
__global__ __launch_bounds__(NT) void flatten_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope, int act,
                                                         float* __restrict__ flat, int B, int HW, int C, int gB) {
  // gB: images per coefficient group (scale / shift are [B / gB][C]: passes batched as one tall tensor)
  const int64_t total = (int64_t)B * HW * C;
  for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int hw = (int)(p % HW);
    const int64_t b = p / HW;
    float v = y[i];
    if (scale) {
      const int gc = (int)(b / gB) * C + c;
      v = fmaf(v, scale[gc], shift[gc]);
    }
    if (act) v = v > 0.f ? v : v * slope;
    flat[(b * C + c) * HW + hw] = v;
  }
}
}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
SST_API int sst_linear_ksplit(int M, int N, int K) {
  int ks = 1024 / ((N + 15) / 16);          // ~1024 workgroups
  if (ks < 1) ks = 1;
  while (ks > 1 && K / ks < 256) ks >>= 1;  // at least 256 k per slice
  return ks;
}

SST_API int sst_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* slab, int M, int N, int K,
                           void* stream) {
  SST_REQUIRE(x && w && y && slab && M > 0 && M <= MAXM && N > 0 && K > 0, "sst_linear_fwd: bad argument (M=%d)", M);
  const int ks = sst_linear_ksplit(M, N, K);
  const int kslice = ((K + ks - 1) / ks + 63) / 64 * 64;
  dim3 grid((N + 15) / 16, (K + kslice - 1) / kslice);
  linear_fwd_kernel<<<grid, NT, 0, sst_stream(stream)>>>(x, w, slab, M, N, K, kslice);
  SST_LAUNCH_CHECK("linear_fwd_kernel");
  const int rb = (M * N + NT - 1) / NT;
  linear_reduce_kernel<<<rb < 512 ? rb : 512, NT, 0, sst_stream(stream)>>>(slab, bias, y, grid.y, M, N);
  SST_LAUNCH_CHECK("linear_reduce_kernel");
  return SST_OK;
}

SST_API int sst_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, int nhwc_C, int nhwc_HW,
                             void* stream) {
  SST_REQUIRE(dy && w && dx && M > 0 && M <= MAXM && N > 0 && K > 0, "sst_linear_dgrad: bad argument");
  SST_REQUIRE(nhwc_HW == 0 || nhwc_C * nhwc_HW == K, "sst_linear_dgrad: C*HW must equal K");
  const int grid = (K + 63) / 64;
  hipStream_t st = sst_stream(stream);
  switch ((M + 15) / 16) {
    case 1: linear_dgrad_kernel<1><<<grid, NT, 0, st>>>(dy, w, dx, M, N, K, nhwc_C, nhwc_HW); break;
    case 2: linear_dgrad_kernel<2><<<grid, NT, 0, st>>>(dy, w, dx, M, N, K, nhwc_C, nhwc_HW); break;
    case 3: linear_dgrad_kernel<3><<<grid, NT, 0, st>>>(dy, w, dx, M, N, K, nhwc_C, nhwc_HW); break;
    default: linear_dgrad_kernel<4><<<grid, NT, 0, st>>>(dy, w, dx, M, N, K, nhwc_C, nhwc_HW); break;
  }
  SST_LAUNCH_CHECK("linear_dgrad_kernel");
  return SST_OK;
}

SST_API int sst_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, int accumulate,
                             void* stream) {
  SST_REQUIRE(dy && x && dw && M > 0 && M <= MAXM && N > 0 && K > 0, "sst_linear_wgrad: bad argument");
  if ((K & 31) == 0 && (N & 63) == 0 && !sst_env("SST_LINEAR_WGRAD_FMA")) {
    linear_wgrad_mfma_kernel<<<dim3((K / 32 + 3) / 4, N / 64), NT, 0, sst_stream(stream)>>>(dy, x, dw, M, N, K, accumulate);
    SST_LAUNCH_CHECK("linear_wgrad_mfma_kernel");
  } else {
    dim3 grid((K + 4 * NT - 1) / (4 * NT), (N + 15) / 16);
    linear_wgrad_kernel<<<grid, NT, 0, sst_stream(stream)>>>(dy, x, dw, M, N, K, accumulate);
    SST_LAUNCH_CHECK("linear_wgrad_kernel");
  }
  if (db) {
    colsum_small_kernel<<<(N + 255) / 256, 256, 0, sst_stream(stream)>>>(dy, db, M, N, accumulate);
    SST_LAUNCH_CHECK("colsum_small_kernel");
  }
  return SST_OK;
}

SST_API int sst_head_fwd(const float* h, const float* w, const float* b, float* y, int M, int N, int K, float slope,
                         void* stream) {
  SST_REQUIRE(h && w && y && M > 0 && N > 0 && N <= 64 && K > 0, "sst_head_fwd: bad argument");
  head_fwd_kernel<<<M * N, NT, 0, sst_stream(stream)>>>(h, w, b, y, M, N, K, slope);
  SST_LAUNCH_CHECK("head_fwd_kernel");
  return SST_OK;
}

SST_API int sst_head_bwd(const float* h, const float* w, const float* dy, float* dh, float* dw, float* db, int M, int N,
                         int K, float slope, int accumulate, void* stream) {
  SST_REQUIRE(h && w && dy && dh && M > 0 && N > 0 && N <= 64 && K > 0, "sst_head_bwd: bad argument");
  head_bwd_kernel<<<dim3((K + NT - 1) / NT, (M + HB_ROWS - 1) / HB_ROWS), NT, 0, sst_stream(stream)>>>(h, w, dy, dh, dw, db, M, N, K, slope,
                                                                                                    accumulate);
  SST_LAUNCH_CHECK("head_bwd_kernel");
  return SST_OK;
}

// grp_images > 0: scale / shift are [B / grp_images][C] (passes batched as one tensor, each with its own BatchNorm coefficients)
SST_API int sst_flatten_act_grp(const float* y, const float* scale, const float* shift, float slope, int act, float* flat,
                                int B, int HW, int C, int grp_images, void* stream) {
  SST_REQUIRE(y && flat && B > 0 && HW > 0 && C > 0 && ((scale == nullptr) == (shift == nullptr)), "sst_flatten_act: bad argument");
  SST_REQUIRE(grp_images >= 0 && (grp_images == 0 || B % grp_images == 0), "sst_flatten_act: bad group size %d for B=%d", grp_images, B);
  const int64_t total = (int64_t)B * HW * C;
  const int blocks = (int)((total + NT - 1) / NT < 2048 ? (total + NT - 1) / NT : 2048);
  flatten_act_kernel<<<blocks, NT, 0, sst_stream(stream)>>>(y, scale, shift, slope, act, flat, B, HW, C, grp_images > 0 ? grp_images : B);
  SST_LAUNCH_CHECK("flatten_act_kernel");
  return SST_OK;
}
SST_API int sst_flatten_act(const float* y, const float* scale, const float* shift, float slope, int act, float* flat,
                            int B, int HW, int C, void* stream) {
  return sst_flatten_act_grp(y, scale, shift, slope, act, flat, B, HW, C, 0, stream);
}

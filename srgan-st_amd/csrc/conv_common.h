// Shared pieces of the fp32-MFMA implicit-GEMM convolution kernels (gfx950).
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain (no TF32/bf16 anywhere), so the
// only difference from the reference's fp32 conv is summation order.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_NT = 256;   // threads per workgroup (4 waves)
constexpr int CB = 64;         // input-channel block staged in LDS at a time
constexpr int LDSC = CB + 4;   // padded per-pixel stride in LDS (floats): keeps 16-B alignment, spreads banks

// activation applied to the conv INPUT while it is staged (fusion of the producer's BN-apply + activation)
enum : int { ACT_NONE = 0, ACT_SLOPE = 1 /* x>0 ? x : slope*x  (PReLU scalar / LeakyReLU / ReLU with slope 0) */ };

// Packed weight layout for the MFMA B operand ("fragment-major"):
//   idx(o, i, tap) = ((((o/32 * NCB + i/64) * KK + tap) * 8 + (i%64)/8) * 64 + ((i%8)/4)*32 + o%32) * 4 + i%4
// so that one wave's B fragment for (32 outputs, 8 inputs, one tap) is 1 KiB contiguous and lane l reads 16 B at l*16.
__host__ __device__ inline int64_t packed_index(int o, int i, int tap, int ncb, int kk) {
  return ((((int64_t)(o >> 5) * ncb + (i >> 6)) * kk + tap) * 8 + ((i & 63) >> 3)) * 256 + (((i & 7) >> 2) * 32 + (o & 31)) * 4 +
         (i & 3);
}
constexpr int PACK_PAD = 8 * 256;   // the K loop prefetches up to 4 chunks (1 KiB each) past a block: keep them in-bounds
__host__ __device__ inline int64_t packed_floats_base(int cout, int cin, int kk) {
  return (int64_t)((cout + 31) / 32) * ((cin + 63) / 64) * kk * 8 * 256 + PACK_PAD;
}

// Second section of the packed buffer, present when band_eligible(): the same weights in the layout of the band kernel
// (conv_band.hip: v_mfma_f32_16x16x4_f32 with M = 16 outputs, K = 4 inputs; wave w owns inputs 16w..16w+15 and reads its
// A fragments for the 4 k-steps of one tap as ONE 16-B load per lane, 1 KiB per wave):
//   band_index(o, i, tap) = (((o/16 * 9 + tap) * 4 + i/16) * 64 + (o%16) + 16*((i%16)/4)) * 4 + i%4
__host__ __device__ inline bool band_eligible(int cout, int cin, int kk) { return kk == 9 && cin == 64 && (cout & 15) == 0; }
__host__ __device__ inline int64_t band_index(int o, int i, int tap) {
  return ((((int64_t)(o >> 4) * 9 + tap) * 4 + (i >> 4)) * 64 + (o & 15) + 16 * ((i & 15) >> 2)) * 4 + (i & 3);
}
__host__ __device__ inline int64_t packed_floats(int cout, int cin, int kk) {
  return packed_floats_base(cout, cin, kk) + (band_eligible(cout, cin, kk) ? (int64_t)cout * cin * 9 : 0);
}

// 9x9 convs with a 3-channel side (conv9_c3.hip): packed value at flat index idx.  Shared by the stand-alone pack kernels and
// the multi-tensor pack (conv_fwd.hip: PackJob modes 2..4), so that one launch packs every weight of a network.
//   c3 (mode 0: w [Cout][3][9][9] conv1 forward; mode 1: w [3][C][9][9] data-gradient of conv3): [O/32][36 chunks][64 lanes][4]
__host__ __device__ inline int64_t c3_packed_floats(int O) { return (int64_t)((O + 31) / 32) * 36 * 256; }
__device__ inline float pack_c3_value(const float* __restrict__ w, int64_t idx, int Cout, int Cin, int mode) {
  const int O = mode ? Cin : Cout;
  const int jj = idx & 3, l = (idx >> 2) & 63;
  const int chunk = (int)((idx >> 8) % 36), of = (int)((idx >> 8) / 36);
  const int ky = chunk >> 2, c4 = chunk & 3;
  const int o = of * 32 + (l & 31), j = c4 * 8 + (l >> 5) * 4 + jj;
  if (o >= O || j >= 27) return 0.f;
  const int kx = j / 3, ch = j - 3 * kx;
  return mode == 0 ? w[(((size_t)o * Cin + ch) * 9 + ky) * 9 + kx] : w[(((size_t)ch * Cin + o) * 9 + (8 - ky)) * 9 + (8 - kx)];
}
//   to3 (w [3][C][9][9], conv3 forward): [C/64][ky][ks 0..7][64 lanes][4] = B[k = (ky, ci)][n = 3 kx + co]  (+ PACK_PAD zeros)
__host__ __device__ inline int64_t to3_packed_floats(int C) { return (int64_t)((C + 63) / 64) * 9 * 8 * 256 + PACK_PAD; }
__device__ inline float pack_to3_value(const float* __restrict__ w, int64_t idx, int C) {
  const int jj = idx & 3, l = (idx >> 2) & 63, ks = (idx >> 8) & 7;
  const int64_t rest = idx >> 11;
  const int ky = (int)(rest % 9), cb = (int)(rest / 9);
  const int q = l & 31, ci = cb * 64 + ks * 8 + (l >> 5) * 4 + jj;
  if (cb >= (C + 63) / 64 || q >= 27 || ci >= C) return 0.f;
  const int kx = q / 3, co = q - 3 * kx;
  return w[(((size_t)co * C + ci) * 9 + ky) * 9 + kx];
}

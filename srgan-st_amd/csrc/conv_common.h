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

/* libsrganst.so - C ABI of the MI355X (gfx950) SRGAN-ST training hot path.
 *
 * The reference (SebastianBitsch/SRGAN-ST) has no FFI of its own: its boundary is the Python
 * module surface.  These entry points sit underneath that surface; each one cites
 * the reference code it replaces.  Conventions:
 *   - every function returns int: 0 = ok, <0 = error (message: sst_last_error(), thread-local);
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer
 *     (inputs, outputs, saved-for-backward, workspace) - the library never allocates;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*), re-entrant,
 *     graph-capturable (no sync, no malloc inside);
 *   - `counter` words must be zero on first use; kernels leave them zero again.
 */
#ifndef SRGANST_H
#define SRGANST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* sst_last_error(void);
int sst_version(void);
const char* sst_arch(void);

/* ---- structure-tensor loss: loss.py:380-413 + utils.py:194-280 (sigma=.5, rho=2 default) -------
 * sr, gt, dsr: NCHW [B,3,H,W].  gS: saved [B,3,H,W].  partials: sst_st_loss_workspace() floats.
 * fwd writes loss[0] = mean_b mean_px d  and gS = d(sum_px d)/d(Jxx,Jyy,Jxy)(sr).
 * bwd: dsr (+)= scale_host * (scale_dev ? *scale_dev : 1) * d loss / d sr.                        */
int sst_st_loss_workspace(int B, int H, int W, int64_t* partial_floats);
int sst_st_loss_fwd(const float* sr, const float* gt, float* loss, float* gS, float* partials,
                    unsigned* counter, int B, int H, int W, float sigma, float rho, int normalize,
                    void* stream);
int sst_st_loss_bwd(const float* sr, const float* gS, float* dsr, const float* scale_dev,
                    float scale_host, int accumulate, int B, int H, int W, float sigma, float rho,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif

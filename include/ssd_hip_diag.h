/*
 * ssd_hip_diag.h -- diagnostics entry points of libssd_hip_diag.so, the -DSSD_DIAG build of the same
 * sources (single-shot-detector_amd/_lib.py build_diag()).  Only scripts/ load that library.  The shipped
 * libssd_hip.so exports none of this and contains no ablation kernels, tile overrides or timestamp code paths:
 * some of these switches produce WRONG results by design (timing experiments).
 */
#ifndef SSD_HIP_DIAG_H
#define SSD_HIP_DIAG_H
#include "ssd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics (scripts/bench_conv.py): average milliseconds of `reps` launches of one dense
 * convolution with BN + ReLU on random data, with an explicit implicit-GEMM tile variant
 * (0: 128x128, 1: 128x64, 2: 128x32, 5: 64x64, 6: 128x96; -1: the library's choice).
 * pyramid != 0 runs the five-level head-tower launch shape (H,W halved per level). */
int ssd_bench_conv(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t k,
                   int32_t stride, int32_t tile, int32_t reps, int32_t pyramid, double *avg_ms,
                   double *gflop);

/* Diagnostics (scripts/bench_dwpw.py): average milliseconds of `reps` launches of one depthwise +
 * pointwise block (BN + ReLU6 after each) on random data: fused != 0 as the single ssd_dw_pw
 * kernel, else as the depthwise kernel followed by the implicit-GEMM kernel. */
int ssd_bench_dwpw(int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t stride,
                   int32_t fused, int32_t reps, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SSD_HIP_DIAG_H */

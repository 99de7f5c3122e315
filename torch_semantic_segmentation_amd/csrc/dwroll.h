// Row-pipelined depthwise 3x3 kernels (dwroll.hip): entry points for dwconv.hip's C ABI functions.
#pragma once
#include <hip/hip_runtime.h>

namespace tss {

// true when the (bf16, dilation 1, stride 1 / 2) layer is inside the row-pipelined kernels' envelope
bool dwroll_supported(int C, int stride, int dil, int dtype);

// same operands as tss_dwconv3x3_fwd (bf16); launches on `stream`
void dwroll_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                const float* w, void* y, long ldy, double* stats,
                int B, int Hin, int Win, int C, int stride, hipStream_t stream);

// stride-1 / stride-2 backward in one sweep (input gradient + weight-gradient partial rows in ws): same operands as
// tss_dwconv3x3_bwd_fused; returns the number of workspace rows written (to be summed into dw by the caller)
bool dwroll_bwd_fused_supported(int C, int stride, int dil, int dtype);
int dwroll_bwd_fused(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                     const float* gmu, const float* w, const void* x, long ldx, const float* in_mean, const float* in_scale,
                     const float* in_bias, int in_relu, int x_pending, void* e_in, long ldei, double* bstats, float* ws,
                     int B, int H, int W, int C, int stride, hipStream_t stream);

}  // namespace tss

// Wire-format decode for the training batch (SURVEY.md section 8d / VERDICT r01 #10): the reference ships float32 images and
// int64 targets from the DataLoader to the device every step (TSS/engine.py:27: 335 MB at 8 x 3 x 1024 x 2048 -- as long as
// the whole step over PCIe Gen5).  The loader can ship what the dataset actually holds instead -- uint8 pixels, uint8
// labels, 67 MB -- and this kernel writes the float32 NCHW image (albumentations.Normalize semantics:
// (x / 255 - mean[c]) / std[c], scripts/train_fastscnn.py:62-68) and the int64 target straight into the buffers the
// captured step reads.
#include "common.h"

namespace {

// image: one thread = 8 consecutive pixels of one (b, c) plane row segment.  src layout: HWC (hwc = 1, what image
// decoders produce) or CHW.
__global__ __launch_bounds__(256) void decode_image_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                              long B, int C, long HW, int hwc, float3 scale, float3 shift) {
  const long groups = B * C * (HW / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long plane = i / (HW / 8);
    const long off = (i - plane * (HW / 8)) * 8;
    const int c = (int)(plane % C);
    const long b = plane / C;
    const float sc = c == 0 ? scale.x : (c == 1 ? scale.y : scale.z);
    const float sh = c == 0 ? shift.x : (c == 1 ? shift.y : shift.z);
    float v[8];
    if (hwc) {
      const unsigned char* p = src + (b * HW + off) * C + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)p[(long)j * C] * sc + sh;
    } else {
      const uint2 r = *reinterpret_cast<const uint2*>(src + plane * HW + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = (float)((r.x >> (8 * j)) & 0xffu) * sc + sh; v[4 + j] = (float)((r.y >> (8 * j)) & 0xffu) * sc + sh; }
    }
    V8<float>::store(dst + plane * HW + off, v);
  }
}

__global__ __launch_bounds__(256) void decode_target_u8_kernel(const unsigned char* __restrict__ src, long long* __restrict__ dst, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const uint2 r = *reinterpret_cast<const uint2*>(src + i);
    long long v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (long long)((r.x >> (8 * j)) & 0xffu); v[4 + j] = (long long)((r.y >> (8 * j)) & 0xffu); }
#pragma unroll
    for (int j = 0; j < 8; j += 2) *reinterpret_cast<longlong2*>(dst + i + j) = make_longlong2(v[j], v[j + 1]);
  }
}

}  // namespace

extern "C" {

int tss_decode_batch_u8(const unsigned char* image, int image_is_hwc, const float* mean3, const float* std3, float* image_out,
                        const unsigned char* target, long long* target_out, long B, int C, long HW, void* stream) {
  TSS_REQUIRE(B >= 0 && C >= 1 && C <= 3 && HW > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE((!image || (image_out && tss::aligned16(image_out) && (image_is_hwc || (reinterpret_cast<uintptr_t>(image) & 7u) == 0))) &&
              (!target || (target_out && tss::aligned16(target_out) && (reinterpret_cast<uintptr_t>(target) & 7u) == 0)), TSS_ERR_ALIGN);
  if (B == 0) return TSS_OK;
  if (image) {
    float sc[3] = {1.f / 255.f, 1.f / 255.f, 1.f / 255.f}, sh[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {            // host arrays: (x/255 - mean) / std = x * (1/(255 std)) - mean/std
      const float m = mean3 ? mean3[c] : 0.f, s = std3 ? std3[c] : 1.f;
      sc[c] = 1.f / (255.f * s); sh[c] = -m / s;
    }
    const long groups = B * C * (HW / 8);
    long grid = (groups + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(decode_image_u8_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, image, image_out, B, C, HW,
                       image_is_hwc, make_float3(sc[0], sc[1], sc[2]), make_float3(sh[0], sh[1], sh[2]));
  }
  if (target) {
    const long n = B * HW;
    long grid = (n / 8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(decode_target_u8_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, target, target_out, n);
  }
  return tss::check_last("decode_batch_u8");
}

}  // extern "C"

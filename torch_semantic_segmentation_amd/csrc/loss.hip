// Cross-entropy over NCHW-planar logits (the caller side of the hot path: nn.CrossEntropyLoss(ignore_index)
// as the training scripts use it), plus the argmax / confusion-matrix pass of the evaluator.
// One lane owns 8 consecutive pixels of a row: every class plane is read/written as 16-byte vectors.
// forward : online log-sum-exp over the C planes -> per-pixel lse (saved), sum of losses + valid count (f64 atomics)
// backward: dlogits[c] = (exp(l_c - lse) - [c == t]) * grad_out / count, recomputed from the saved lse
#include "common.h"

namespace {

constexpr int NT = 256;

template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T* logits, const long long* target, float* lse_out,
                                                    double* acc /*[2]: loss sum, valid count*/, long B, int C,
                                                    long HW, int ignore_index) {
  __shared__ double red[2][NT / 64];
  const long groups = B * (HW / 8);
  double lsum = 0.0, lcnt = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    const T* base = logits + b * C * HW + off;
    float m[8], s[8], lt[8];
    long long t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; s[j] = 0.f; lt[j] = 0.f; t[j] = target[b * HW + off + j]; }
    for (int c = 0; c < C; ++c) {
      float v[8];
      V8<T>::load(base + (long)c * HW, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mn = fmaxf(m[j], v[j]);
        s[j] = s[j] * __expf(m[j] - mn) + __expf(v[j] - mn);
        m[j] = mn;
        if (t[j] == c) lt[j] = v[j];
      }
    }
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = m[j] + __logf(s[j]);
      if (t[j] != ignore_index) { lsum += (double)(l[j] - lt[j]); lcnt += 1.0; }
    }
    V8<float>::store(lse_out + b * HW + off, l);
  }
  lsum = wave_sum(lsum);
  lcnt = wave_sum(lcnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = lsum; red[1][wave] = lcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < NT / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    atomicAdd(acc, a);
    atomicAdd(acc + 1, c);
  }
}

__global__ void ce_finalize_kernel(const double* acc, float* loss, float* inv_count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double n = acc[1];
    *loss = (float)(acc[0] / n);          // n == 0 -> nan, like torch
    *inv_count = n > 0.0 ? (float)(1.0 / n) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const T* logits, const long long* target, const float* lse,
                                                    const float* inv_count, const float* grad_out, T* dlogits,
                                                    long B, int C, long HW, int ignore_index) {
  const long groups = B * (HW / 8);
  const float gs = (*inv_count) * (grad_out ? *grad_out : 1.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    float l[8], w[8];
    long long t[8];
    V8<float>::load(lse + b * HW + off, l);
#pragma unroll
    for (int j = 0; j < 8; ++j) { t[j] = target[b * HW + off + j]; w[j] = (t[j] != ignore_index) ? gs : 0.f; }
    for (int c = 0; c < C; ++c) {
      float v[8], d[8];
      V8<T>::load(logits + (b * C + c) * HW + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = (__expf(v[j] - l[j]) - (t[j] == c ? 1.f : 0.f)) * w[j];
      V8<T>::store(dlogits + (b * C + c) * HW + off, d);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Fused decoder head + loss: cross-entropy of the x`scale` bilinearly upsampled logits
// computed straight from the LOW-RES NHWC logits.  The (B, C, H, W) full-resolution logits -- 318.8 M elements at
// 8x1024x2048, the largest tensor of the step -- and their gradient are never written to HBM.
//   forward : lane = 8 consecutive output pixels of a row; per class the 8 logits are interpolated from <= 3 source
//             columns x 2 rows, folded into an online log-sum-exp; saves lse (f32) and the target as u8.
//   backward: rows pass recomputes logit_c(oy, ox), d = (exp(l - lse) - [t == c]) / count and gathers it down the
//             column window of source row iy (deterministic); the columns pass is tss_upsample_head_bwd's.
template <typename T>
__global__ __launch_bounds__(NT) void upsample_ce_fwd_kernel(const T* low, long ldl, const long long* target,
                                                             float* lse_out, unsigned char* t8, double* acc,
                                                             int B, int C, int h, int w, int H, int W, int ignore_index) {
  __shared__ double red[2][NT / 64];
  const int W8 = W / 8;
  const long groups = (long)B * H * W8;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  double lsum = 0.0, lcnt = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(i % W8);
    long p = i / W8;
    const int oy = (int)(p % H);
    const long b = p / H;
    const Tap ty = ac_tap(sy, oy, h);
    Tap tx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tx[j] = ac_tap(sx, xg * 8 + j, w);
    const T* r0 = low + ((b * h + ty.i0) * (long)w) * ldl;
    const T* r1 = low + ((b * h + ty.i1) * (long)w) * ldl;
    const long pix = (b * H + oy) * (long)W + xg * 8;
    long long t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = target[pix + j];
    float m[8], s[8], lt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; s[j] = 0.f; lt[j] = 0.f; }
    for (int c = 0; c < C; ++c) {
      int cached = -1;
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (tx[j].i0 != cached) {
          a0 = (float)r0[(long)tx[j].i0 * ldl + c]; a1 = (float)r1[(long)tx[j].i0 * ldl + c];
          b0 = (float)r0[(long)tx[j].i1 * ldl + c]; b1 = (float)r1[(long)tx[j].i1 * ldl + c];
          cached = tx[j].i0;
        }
        const float v = ty.l0 * (tx[j].l0 * a0 + tx[j].l1 * b0) + ty.l1 * (tx[j].l0 * a1 + tx[j].l1 * b1);
        const float mn = fmaxf(m[j], v);
        s[j] = s[j] * __expf(m[j] - mn) + __expf(v - mn);
        m[j] = mn;
        if (t[j] == c) lt[j] = v;
      }
    }
    float l[8];
    unsigned long long packed = 0ull;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = m[j] + __logf(s[j]);
      const bool valid = t[j] != ignore_index;
      if (valid) { lsum += (double)(l[j] - lt[j]); lcnt += 1.0; }
      packed |= (unsigned long long)(valid ? (unsigned char)t[j] : 255u) << (8 * j);
    }
    V8<float>::store(lse_out + pix, l);
    *reinterpret_cast<unsigned long long*>(t8 + pix) = packed;
  }
  lsum = wave_sum(lsum);
  lcnt = wave_sum(lcnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = lsum; red[1][wave] = lcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int wv = 0; wv < NT / 64; ++wv) { a += red[0][wv]; c += red[1][wv]; }
    atomicAdd(acc, a);
    atomicAdd(acc + 1, c);
  }
}

// tmp[b][c][iy][ox] = sum_oy wy(oy, iy) * (softmax_c(oy, ox) - [t == c]) * valid / count * grad_out
template <typename T>
__global__ __launch_bounds__(NT) void upsample_ce_bwd_rows_kernel(const T* low, long ldl, const unsigned char* t8,
                                                                  const float* lse, const float* inv_count,
                                                                  const float* grad_out, float* tmp,
                                                                  int B, int C, int h, int w, int H, int W) {
  const int W8 = W / 8;
  const long total = (long)B * C * h * W8;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const float gs = (*inv_count) * (grad_out ? *grad_out : 1.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(i % W8);
    long p = i / W8;
    const int iy = (int)(p % h); p /= h;
    const int c = (int)(p % C);
    const long b = p / C;
    Tap tx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tx[j] = ac_tap(sx, xg * 8 + j, w);
    int lo, hi;
    ac_window(sy, iy, H, &lo, &hi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = lo; oy <= hi; ++oy) {
      const Tap ty = ac_tap(sy, oy, h);
      float wy = 0.f;
      if (ty.i0 == iy) wy += ty.l0;
      if (ty.i1 == iy) wy += ty.l1;
      if (wy == 0.f) continue;
      const T* r0 = low + ((b * h + ty.i0) * (long)w) * ldl + c;
      const T* r1 = low + ((b * h + ty.i1) * (long)w) * ldl + c;
      const long pix = (b * H + oy) * (long)W + xg * 8;
      float l[8];
      V8<float>::load(lse + pix, l);
      const unsigned long long tp = *reinterpret_cast<const unsigned long long*>(t8 + pix);
      int cached = -1;
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (tx[j].i0 != cached) {
          a0 = (float)r0[(long)tx[j].i0 * ldl]; a1 = (float)r1[(long)tx[j].i0 * ldl];
          b0 = (float)r0[(long)tx[j].i1 * ldl]; b1 = (float)r1[(long)tx[j].i1 * ldl];
          cached = tx[j].i0;
        }
        const float v = ty.l0 * (tx[j].l0 * a0 + tx[j].l1 * b0) + ty.l1 * (tx[j].l0 * a1 + tx[j].l1 * b1);
        const int tj = (int)((tp >> (8 * j)) & 0xffu);
        const float d = (tj == 255) ? 0.f : (__expf(v - l[j]) - (tj == c ? 1.f : 0.f));
        acc[j] += wy * d;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= gs;
    V8<float>::store(tmp + ((b * C + c) * (long)h + iy) * W + xg * 8, acc);
  }
}

// argmax over class planes (lowest index wins ties, like torch.argmax) + confusion matrix [C][C] (rows = truth)
template <typename T>
__global__ __launch_bounds__(NT) void argmax_confusion_kernel(const T* logits, const long long* target,
                                                              unsigned char* pred_out, unsigned long long* cm,
                                                              long B, int C, long HW, int ignore_index) {
  extern __shared__ unsigned int scm[];  // [C*C]
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) scm[i] = 0u;
  __syncthreads();
  const long groups = B * (HW / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    for (int c = 0; c < C; ++c) {
      float v[8];
      V8<T>::load(logits + (b * C + c) * HW + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[j] > best[j] || (c == 0)) { best[j] = v[j]; arg[j] = c; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (pred_out) pred_out[b * HW + off + j] = (unsigned char)arg[j];
      if (cm && target) {
        const long long t = target[b * HW + off + j];
        if (t != ignore_index && t >= 0 && t < C) atomicAdd(&scm[(int)t * C + arg[j]], 1u);
      }
    }
  }
  __syncthreads();
  if (cm)
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
      if (scm[i]) atomicAdd(cm + i, (unsigned long long)scm[i]);
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

extern "C" {

int tss_cross_entropy_fwd(const void* logits, const long long* target, float* lse, double* acc /*[2], zeroed*/,
                          float* loss, float* inv_count, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(lse), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  {
    tss::ProfScope prof(TSS_K_CE_FWD, (hipStream_t)stream, (double)B * HW * (C * esz(dtype) + 12.0), 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const bf16_t*)logits, target, lse, acc, B, C, HW, ignore_index);
    else
      hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const float*)logits, target, lse, acc, B, C, HW, ignore_index);
  }
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, loss, inv_count);
  return tss::check_last("cross_entropy_fwd");
}

int tss_cross_entropy_bwd(const void* logits, const long long* target, const float* lse, const float* inv_count,
                          const float* grad_out, void* dlogits, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(dlogits) && tss::aligned16(lse), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_CE_BWD, (hipStream_t)stream, (double)B * HW * (2.0 * C * esz(dtype) + 12.0), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)logits, target, lse, inv_count, grad_out, (bf16_t*)dlogits, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)logits, target, lse, inv_count, grad_out, (float*)dlogits, B, C, HW, ignore_index);
  return tss::check_last("cross_entropy_bwd");
}

int tss_argmax_confusion(const void* logits, const long long* target, unsigned char* pred,
                         unsigned long long* confusion /*[C*C] accumulated*/, long B, int C, long HW,
                         int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 64 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  long grid = grid_for(groups);
  if (grid > 1024) grid = 1024;
  tss::ProfScope prof(TSS_K_ARGMAX, (hipStream_t)stream, (double)B * HW * (C * esz(dtype) + 9.0), 0);
  const size_t sh = (size_t)C * C * sizeof(unsigned int);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(argmax_confusion_kernel<bf16_t>, dim3((int)grid), dim3(NT), sh, (hipStream_t)stream,
                       (const bf16_t*)logits, target, pred, confusion, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(argmax_confusion_kernel<float>, dim3((int)grid), dim3(NT), sh, (hipStream_t)stream,
                       (const float*)logits, target, pred, confusion, B, C, HW, ignore_index);
  return tss::check_last("argmax_confusion");
}

int tss_upsample_ce_fwd(const void* low, long ldl, const long long* target, float* lse, unsigned char* target_u8,
                        double* acc /*[2], zeroed*/, float* loss, float* inv_count,
                        int B, int C, int h, int w, int H, int W, int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C < 255 && ldl >= C && (W % 8) == 0 && h > 0 && w > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(lse) && (reinterpret_cast<uintptr_t>(target_u8) & 7u) == 0, TSS_ERR_ALIGN);
  const long groups = (long)B * H * (W / 8);
  if (groups == 0) return TSS_OK;
  {
    tss::ProfScope prof(TSS_K_UPSAMPLE_CE_FWD, (hipStream_t)stream,
                        (double)B * h * w * C * esz(dtype) + (double)B * H * W * 13.0, 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(upsample_ce_fwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const bf16_t*)low, ldl, target, lse, target_u8, acc, B, C, h, w, H, W, ignore_index);
    else
      hipLaunchKernelGGL(upsample_ce_fwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const float*)low, ldl, target, lse, target_u8, acc, B, C, h, w, H, W, ignore_index);
  }
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, loss, inv_count);
  return tss::check_last("upsample_ce_fwd");
}

int tss_upsample_ce_bwd_rows(const void* low, long ldl, const unsigned char* target_u8, const float* lse,
                             const float* inv_count, const float* grad_out, float* tmp /*[B*C*h*W] f32*/,
                             int B, int C, int h, int w, int H, int W, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C < 255 && ldl >= C && (W % 8) == 0 && h > 0 && w > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(lse) && tss::aligned16(tmp), TSS_ERR_ALIGN);
  const long total = (long)B * C * h * (W / 8);
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_UPSAMPLE_CE_BWD_ROWS, (hipStream_t)stream,
                      (double)B * H * W * 5.0 + (double)B * C * h * W * 4.0 + (double)B * h * w * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(upsample_ce_bwd_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)low, ldl, target_u8, lse, inv_count, grad_out, tmp, B, C, h, w, H, W);
  else
    hipLaunchKernelGGL(upsample_ce_bwd_rows_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)low, ldl, target_u8, lse, inv_count, grad_out, tmp, B, C, h, w, H, W);
  return tss::check_last("upsample_ce_bwd_rows");
}

}  // extern "C"

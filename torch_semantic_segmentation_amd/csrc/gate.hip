// Channel attention gates of BiSeNet:
//   AttentionRefinementModule  TSS/models/bisenet.py:134-148   out = sigmoid(a) * x
//   FeatureFusionModule        TSS/models/bisenet.py:112-131   out = x * (1 + sigmoid(a))
// with a = the output of a 1x1 convolution on the globally pooled map ([B][C], one value per image and channel) and x the
// [B][H][W][C] NHWC map.  Forward: elementwise, 16 bytes per lane.  Backward: dx = g * s (s = sigmoid(a) + add_one) and
// da[b][c] = sigmoid'(a) * sum over the pixels of image b of g * x -- a per-(image, channel) sum: every block owns one row slice
// of one image, leaves its partial sums in a workspace row (no atomics) and a second, tiny kernel adds the slices in order.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

template <typename T>
__global__ __launch_bounds__(NT) void gate_fwd_kernel(const T* x, long ldx, const T* a, long lda, T* out, long ldo, long HW, long P,
                                                      int C, float add_one) {
  const int CV = C >> 3;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    const long b = p / HW;
    float xv[8], av[8], o[8];
    V8<T>::load(x + p * ldx + cv * 8, xv);
    V8<T>::load(a + b * lda + cv * 8, av);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = xv[j] * (sigm(av[j]) + add_one);
    V8<T>::store(out + p * ldo + cv * 8, o);
  }
}

// grid = B * S blocks; block (b, sl) handles pixels [sl * per, (sl + 1) * per) of image b.  thread = (channel vector cv, pixel lane pl)
template <typename T>
__global__ __launch_bounds__(NT) void gate_bwd_kernel(const T* g, long ldg, const T* x, long ldx, const T* a, long lda, T* dx, long lddx,
                                                      float* ws, long HW, int C, int S, float add_one) {
  __shared__ float red[NT * 8];
  const int CV = C >> 3, NPL = NT / CV;
  const int tid = threadIdx.x, cv = tid % CV, pl = tid / CV;
  const long b = blockIdx.x / S;
  const int sl = blockIdx.x % S;
  const long per = (HW + S - 1) / S;
  const long q0 = sl * per, q1 = (q0 + per < HW) ? q0 + per : HW;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < NPL) {
    float av[8], s[8];
    V8<T>::load(a + b * lda + cv * 8, av);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = sigm(av[j]) + add_one;
    for (long q = q0 + pl; q < q1; q += NPL) {
      const long p = b * HW + q;
      float gv[8], xv[8], o[8];
      V8<T>::load(g + p * ldg + cv * 8, gv);
      V8<T>::load(x + p * ldx + cv * 8, xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] = gv[j] * s[j]; acc[j] += gv[j] * xv[j]; }
      V8<T>::store(dx + p * lddx + cv * 8, o);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl * C + cv * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    float t = 0.f;
    for (int r = 0; r < NPL; ++r) t += red[r * C + c];
    ws[((long)b * S + sl) * C + c] = t;
  }
}

// da[b][c] = sigmoid'(a[b][c]) * sum over the S slices (fixed order)
template <typename T>
__global__ __launch_bounds__(NT) void gate_bwd_reduce_kernel(const float* ws, const T* a, long lda, T* da, long ldda, int B, int C, int S) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += ws[((long)b * S + s) * C + c];
  const float sg = sigm((float)a[(long)b * lda + c]);
  da[(long)b * ldda + c] = (T)(t * sg * (1.f - sg));
}

inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

extern "C" {

int tss_gate_slices(int B, long HW) {       // row slices per image of the backward pass (workspace = B * slices * C floats)
  if (B <= 0 || HW <= 0) return 1;
  long s = (1024 + B - 1) / B;
  const long cap = (HW + 63) / 64;          // at least 64 pixels per slice
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : s);
}

int tss_gate_fwd(const void* x, long ldx, const void* a, long lda, void* out, long ldo, int B, long HW, int C, float add_one,
                 int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && C <= 2048 && (ldx % 8) == 0 && ldx >= C && (lda % 8) == 0 && lda >= C && (ldo % 8) == 0 && ldo >= C,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(a) && tss::aligned16(out), TSS_ERR_ALIGN);
  const long P = (long)B * HW;
  if (P == 0) return TSS_OK;
  const long total = P * (C / 8);
  long grid = (total + NT - 1) / NT;
  if (grid > 2048) grid = 2048;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, 2.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(gate_fwd_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)a, lda,
                       (bf16_t*)out, ldo, HW, P, C, add_one);
  else
    hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const float*)x, ldx, (const float*)a, lda,
                       (float*)out, ldo, HW, P, C, add_one);
  return tss::check_last("gate_fwd");
}

int tss_gate_bwd(const void* g, long ldg, const void* x, long ldx, const void* a, long lda, void* dx, long lddx, void* da, long ldda,
                 float* ws, int B, long HW, int C, float add_one, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && C <= 2048 && (C / 8) <= NT && (ldg % 8) == 0 && ldg >= C && (ldx % 8) == 0 && ldx >= C && (lda % 8) == 0
              && lda >= C && (lddx % 8) == 0 && lddx >= C && ldda >= C && ws, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(g) && tss::aligned16(x) && tss::aligned16(a) && tss::aligned16(dx), TSS_ERR_ALIGN);
  if ((long)B * HW == 0) return TSS_OK;
  const int S = tss_gate_slices(B, HW);
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, 3.0 * B * HW * C * esz(dtype), 0);
#define TSS_GATE_BWD(TT)                                                                                                       \
  hipLaunchKernelGGL(gate_bwd_kernel<TT>, dim3(B * S), dim3(NT), 0, (hipStream_t)stream, (const TT*)g, ldg, (const TT*)x, ldx,   \
                     (const TT*)a, lda, (TT*)dx, lddx, ws, HW, C, S, add_one);                                                 \
  hipLaunchKernelGGL(gate_bwd_reduce_kernel<TT>, dim3((B * C + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, ws, (const TT*)a, lda, \
                     (TT*)da, ldda, B, C, S)
  if (dtype == TSS_BF16) { TSS_GATE_BWD(bf16_t); } else { TSS_GATE_BWD(float); }
#undef TSS_GATE_BWD
  return tss::check_last("gate_bwd");
}

}  // extern "C"

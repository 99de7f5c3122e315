// Glue kernels of the model zoo beyond the benchmarked path: LEDNet (TSS/models/lednet.py) and ESNet
// (TSS/models/esnet.py).  None of these is on the FastSCNN / ContextNet step; they are small, HBM-bound, elementwise or
// reduction kernels written for coverage and exactness (fixed summation orders, no atomics), not tuned shape by shape.
//
//   tensor_stats      BatchNorm statistics of a MATERIALISED tensor (slab rows, as the conv epilogues write them): the
//                     BatchNorm2d that follows torch.cat([conv(x), pool(x)]) in DownsamplingBlock (lednet.py:126-144, esnet.py:47-68)
//   bn_bwd_apply      that BatchNorm's input gradient, dz = ga (e - ce) + gb (z - mean)
//   pool_concat       z = cat([y1 + bias, max_pool2d(x, 2)], channel axis) and its backward (arg-max recomputed from x)
//   mul_addrows       out = u * a + r[image]   (APNModule, lednet.py:86-90: x * level4(input) + level5(global pool))
//   scale_rows        out = x * m[image][channel]   (nn.Dropout2d, lednet.py:113 / esnet.py:117,163: the mask is drawn by the caller)
#include "common.h"

namespace {

constexpr int NT = 256;

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// grid = TSS_STAT_SLABS blocks, block i owns pixels [i * per, (i + 1) * per) and slab row i (zeros when it owns none)
template <typename T>
__global__ __launch_bounds__(NT) void tensor_stats_kernel(const T* z, long ldz, long P, int C, double* stats) {
  __shared__ double r0[NT], r1[NT];
  const int tid = threadIdx.x;
  const long per = (P + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < P) ? p0 + per : P;
  double* row = stats + (long)blockIdx.x * 2 * C;
  for (int c0 = 0; c0 < C; c0 += NT) {
    const int cw = (C - c0 < NT) ? C - c0 : NT;
    int cwp = 1;
    while (cwp < cw) cwp <<= 1;                      // lanes per pixel: a power of two <= 256
    const int npl = NT / cwp, c = tid & (cwp - 1), pl = tid / cwp;
    double s = 0.0, q = 0.0;
    if (c < cw)
      for (long p = p0 + pl; p < p1; p += npl) {
        const double v = (double)ldf(z + p * ldz + c0 + c);
        s += v; q += v * v;
      }
    r0[tid] = s; r1[tid] = q;
    __syncthreads();
    if (tid < cw) {
      double a = 0.0, b = 0.0;
      for (int l = 0; l < npl; ++l) { a += r0[l * cwp + tid]; b += r1[l * cwp + tid]; }
      row[c0 + tid] = a; row[C + c0 + tid] = b;
    }
    __syncthreads();
  }
}

// The same sums with 16-byte (bf16) / 32-byte (f32) vectors when the channel count is a multiple of 8: a lane owns 8 channels of every
// (256 / (C / 8))-th pixel of the block's range, four pixels requested per trip; per-lane partials in f32 for bf16 data (a lane sums a few
// hundred values), f64 across the lanes and for f32 data.  (The element-wise kernel above reads 2 bytes per load with one load in flight:
// 0.5 TB/s -- 1.6 ms for the bias gradient of ESNet's full-resolution classifier.)
template <typename T> struct SAcc { typedef float type; };
template <> struct SAcc<float> { typedef double type; };
template <typename T>
__global__ __launch_bounds__(NT) void tensor_stats_vec_kernel(const T* z, long ldz, long P, int C, double* stats) {
  typedef typename SAcc<T>::type A;
  __shared__ double red[2][NT * 8 / 2];              // [sum | squares][lane-row][C] for C <= 1024... sized below: npl * C <= NT * 8
  const int CV = C >> 3, npl = NT / CV;
  const int tid = threadIdx.x, cv = tid % CV, pl = tid / CV;
  const long per = (P + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = (p0 + per < P) ? p0 + per : P;
  A s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0; q[j] = 0; }
  if (pl < npl) {
    constexpr int JU = 4;
    for (long pa = p0 + pl; pa < p1; pa += (long)npl * JU) {
      typename V8<T>::Raw r[JU];
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const long p = pa + (long)u * npl;
        r[u] = V8<T>::load_raw(z + (p < p1 ? p : pa) * ldz + cv * 8);
      }
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        if (pa + (long)u * npl < p1) {
          float v[8];
          V8<T>::unpack(r[u], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[j] += (A)v[j]; q[j] += (A)v[j] * (A)v[j]; }
        }
      }
    }
  }
  // lanes of one channel vector meet in LDS: two passes (sums, squares) through the same [npl][C] array
  double* buf = &red[0][0];
  double* row = stats + (long)blockIdx.x * 2 * C;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
    if (pl < npl) {
#pragma unroll
      for (int j = 0; j < 8; ++j) buf[pl * C + cv * 8 + j] = (double)(which ? q[j] : s[j]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
      double a = 0.0;
      for (int l = 0; l < npl; ++l) a += buf[l * C + c];
      row[which * C + c] = a;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const T* e, long lde, const T* z, long ldz, const float* ga, const float* gb,
                                                          const float* gce, const float* gmu, T* dz, long lddz, long P, int C) {
  const int CV = C >> 3;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    float ev[8], zv[8], o[8];
    V8<T>::load(e + p * lde + cv * 8, ev);
    if (gb) V8<T>::load(z + p * ldz + cv * 8, zv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv * 8 + j;
      o[j] = gb ? ga[c] * (ev[j] - gce[c]) + gb[c] * (zv[j] - gmu[c]) : ga[c] * ev[j];
    }
    V8<T>::store(dz + p * lddz + cv * 8, o);
  }
}

struct PoolArgs {
  const void* x; int x_f32; long sxb, sxc, sxh, sxw;      // the block input, element strides (NHWC activation or NCHW image)
  int Cin, N1, B, Ho, Wo;
};

// torch's max_pool2d scan order (rows, then columns), strict '>' or NaN: the first maximum wins
template <typename T>
__device__ __forceinline__ float pool4(const PoolArgs& g, long b, int oy, int ox, int ci, int* arg) {
  float best = -TSS_INF;
  int at = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long off = b * g.sxb + (long)ci * g.sxc + (long)(2 * oy + (t >> 1)) * g.sxh + (long)(2 * ox + (t & 1)) * g.sxw;
    const float v = g.x_f32 ? reinterpret_cast<const float*>(g.x)[off] : (float)reinterpret_cast<const T*>(g.x)[off];
    if (t == 0 || v > best || v != v) { best = v; at = t; }
  }
  *arg = at;
  return best;
}

template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_fwd_kernel(const PoolArgs g, const T* y1, long ld1, const float* bias, T* z, long ldz) {
  const int Ct = g.N1 + g.Cin;
  const long total = (long)g.B * g.Ho * g.Wo * Ct;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int c = (int)(i % Ct);
    const long p = i / Ct;
    float v;
    if (c < g.N1) {
      v = (float)y1[p * ld1 + c] + (bias ? bias[c] : 0.f);
    } else {
      const int ox = (int)(p % g.Wo);
      const long t = p / g.Wo;
      const int oy = (int)(t % g.Ho);
      int arg;
      v = pool4<T>(g, t / g.Ho, oy, ox, c - g.N1, &arg);
    }
    z[p * ldz + c] = (T)v;
  }
}

// The same operator on an NHWC activation of the output's dtype with whole 8-channel vectors on both sides (N1, Cin multiples of 8): a lane
// owns one 16-byte (bf16) vector of an output pixel -- a copy of y1 (+ bias) or the channel-wise maximum of the four window vectors, first
// maximum wins as above.  (The element-wise kernel: 1.2 TB/s on the 470 MB of LEDNet's second downsampling block.)
template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_fwd_vec_kernel(const PoolArgs g, const T* y1, long ld1, const float* bias, T* z, long ldz) {
  const int V1 = g.N1 >> 3, VT = V1 + (g.Cin >> 3);
  const long total = (long)g.B * g.Ho * g.Wo * VT;
  const T* x = reinterpret_cast<const T*>(g.x);
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % VT);
    const long p = i / VT;
    float v[8];
    if (cv < V1) {
      V8<T>::load(y1 + p * ld1 + cv * 8, v);
      if (bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias[cv * 8 + j];
      }
    } else {
      const int ox = (int)(p % g.Wo);
      const long t = p / g.Wo;
      const int oy = (int)(t % g.Ho);
      const long base = (t / g.Ho) * g.sxb + (long)(2 * oy) * g.sxh + (long)(2 * ox) * g.sxw + (cv - V1) * 8;
      typename V8<T>::Raw r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = V8<T>::load_raw(x + base + (q >> 1) * g.sxh + (q & 1) * g.sxw);
      V8<T>::unpack(r[0], v);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        float u[8];
        V8<T>::unpack(r[q], u);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u[j] > v[j] || u[j] != u[j]) v[j] = u[j];
      }
    }
    V8<T>::store(z + p * ldz + cv * 8, v);
  }
}

// ... and on the f32 NCHW IMAGE (the first downsampling block: 3 input channels, N1 = 29 / 13 convolution channels, a ragged boundary inside
// an 8-channel vector): a lane owns one output vector; its convolution channels come from one 16-byte load of y1 (computed with channels
// padded to whole vectors), its pool channels from four scalar image loads each.
template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_fwd_img_kernel(const PoolArgs g, const T* y1, long ld1, const float* bias, T* z, long ldz) {
  const int Ct = g.N1 + g.Cin, VT = Ct >> 3;
  const long total = (long)g.B * g.Ho * g.Wo * VT;
  const float* x = reinterpret_cast<const float*>(g.x);
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % VT);
    const long p = i / VT;
    float v[8];
    if (cv * 8 < g.N1) {
      V8<T>::load(y1 + p * ld1 + cv * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) if (bias && cv * 8 + j < g.N1) v[j] += bias[cv * 8 + j];
    }
    if (cv * 8 + 8 > g.N1) {
      const int ox = (int)(p % g.Wo);
      const long t = p / g.Wo;
      const int oy = (int)(t % g.Ho);
      const long base = (t / g.Ho) * g.sxb + (long)(2 * oy) * g.sxh + (long)(2 * ox) * g.sxw;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = cv * 8 + j - g.N1;
        if (ci >= 0) {
          const float* px = x + base + (long)ci * g.sxc;
          float best = px[0];
          const float u1 = px[g.sxw], u2 = px[g.sxh], u3 = px[g.sxh + g.sxw];
          if (u1 > best || u1 != u1) best = u1;
          if (u2 > best || u2 != u2) best = u2;
          if (u3 > best || u3 != u3) best = u3;
          v[j] = best;
        }
      }
    }
    V8<T>::store(z + p * ldz + cv * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_bwd_vec_kernel(const PoolArgs g, const T* dz, long lddz, T* dx, long lddx) {
  const int VC = g.Cin >> 3;
  const long total = (long)g.B * g.Ho * g.Wo * VC;
  const T* x = reinterpret_cast<const T*>(g.x);
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % VC);
    const long p = i / VC;
    const int ox = (int)(p % g.Wo);
    const long t = p / g.Wo;
    const int oy = (int)(t % g.Ho);
    const long b = t / g.Ho;
    const long base = b * g.sxb + (long)(2 * oy) * g.sxh + (long)(2 * ox) * g.sxw + cv * 8;
    typename V8<T>::Raw r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = V8<T>::load_raw(x + base + (q >> 1) * g.sxh + (q & 1) * g.sxw);
    float gv[8], best[8];
    int at[8];
    V8<T>::load(dz + p * lddz + g.N1 + cv * 8, gv);
    V8<T>::unpack(r[0], best);
#pragma unroll
    for (int j = 0; j < 8; ++j) at[j] = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      float u[8];
      V8<T>::unpack(r[q], u);
#pragma unroll
      for (int j = 0; j < 8; ++j) if (u[j] > best[j] || u[j] != u[j]) { best[j] = u[j]; at[j] = q; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = at[j] == q ? gv[j] : 0.f;
      const long pin = (b * (2 * g.Ho) + 2 * oy + (q >> 1)) * (long)(2 * g.Wo) + 2 * ox + (q & 1);
      V8<T>::store(dx + pin * lddx + cv * 8, o);
    }
  }
}

// dx[b][2 oy + dy][2 ox + dx][ci] = dz[p][N1 + ci] at the window's arg-max, 0 at its three other pixels (H, W even: every input
// pixel lies in exactly one window)
template <typename T>
__global__ __launch_bounds__(NT) void pool_concat_bwd_kernel(const PoolArgs g, const T* dz, long lddz, T* dx, long lddx) {
  const long total = (long)g.B * g.Ho * g.Wo * g.Cin;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int ci = (int)(i % g.Cin);
    const long p = i / g.Cin;
    const int ox = (int)(p % g.Wo);
    const long t = p / g.Wo;
    const int oy = (int)(t % g.Ho);
    const long b = t / g.Ho;
    int arg;
    pool4<T>(g, b, oy, ox, ci, &arg);
    const T gv = dz[p * lddz + g.N1 + ci];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long pin = (b * (2 * g.Ho) + 2 * oy + (q >> 1)) * (long)(2 * g.Wo) + 2 * ox + (q & 1);
      dx[pin * lddx + ci] = (q == arg) ? gv : (T)0.f;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void mul_addrows_fwd_kernel(const T* u, long ldu, const T* a, long lda, const T* r, long ldr, T* out,
                                                             long ldo, long HW, long P, int C) {
  const int CV = C >> 3;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    float uv[8], av[8], rv[8], o[8];
    V8<T>::load(u + p * ldu + cv * 8, uv);
    V8<T>::load(a + p * lda + cv * 8, av);
    V8<T>::load(r + (p / HW) * ldr + cv * 8, rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = uv[j] * av[j] + rv[j];
    V8<T>::store(out + p * ldo + cv * 8, o);
  }
}

// grid = B * S blocks: block (b, sl) owns a pixel slice of image b; du = g a, da = g u, ws[b][sl][c] = sum over the slice of g
template <typename T>
__global__ __launch_bounds__(NT) void mul_addrows_bwd_kernel(const T* g, long ldg, const T* u, long ldu, const T* a, long lda, T* du,
                                                             long lddu, T* da, long ldda, float* ws, long HW, int C, int S) {
  __shared__ float red[NT * 8];
  const int CV = C >> 3, NPL = NT / CV;
  const int tid = threadIdx.x, cv = tid % CV, pl = tid / CV;
  const long b = blockIdx.x / S;
  const int sl = blockIdx.x % S;
  const long per = (HW + S - 1) / S;
  const long q0 = sl * per, q1 = (q0 + per < HW) ? q0 + per : HW;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < NPL) {
    for (long q = q0 + pl; q < q1; q += NPL) {
      const long p = b * HW + q;
      float gv[8], uv[8], av[8], o0[8], o1[8];
      V8<T>::load(g + p * ldg + cv * 8, gv);
      V8<T>::load(u + p * ldu + cv * 8, uv);
      V8<T>::load(a + p * lda + cv * 8, av);
#pragma unroll
      for (int j = 0; j < 8; ++j) { o0[j] = gv[j] * av[j]; o1[j] = gv[j] * uv[j]; acc[j] += gv[j]; }
      V8<T>::store(du + p * lddu + cv * 8, o0);
      V8<T>::store(da + p * ldda + cv * 8, o1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl * C + cv * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    float t = 0.f;
    for (int rr = 0; rr < NPL; ++rr) t += red[rr * C + c];
    ws[((long)b * S + sl) * C + c] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void rows_reduce_kernel(const float* ws, T* dr, long lddr, int B, int C, int S) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += ws[((long)b * S + s) * C + c];
  dr[(long)b * lddr + c] = (T)t;
}

template <typename T>
__global__ __launch_bounds__(NT) void scale_rows_kernel(const T* x, long ldx, const float* m, T* out, long ldo, long HW, long P, int C) {
  const int CV = C >> 3;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    const float* mr = m + (p / HW) * C + cv * 8;
    float xv[8], o[8];
    V8<T>::load(x + p * ldx + cv * 8, xv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = xv[j] * mr[j];
    V8<T>::store(out + p * ldo + cv * 8, o);
  }
}

// gradient of the input of a split unit: out[p] = cat(gl[p], gr[p]) + gs[p] (gs may be absent) -- the two halves the branches of an
// SS-nbt unit hand back and the gradient of the skip connection, in one pass (autograd's own route is two zero-filled full tensors,
// two slice copies and two adds: 9 C bytes per pixel instead of 2 - 3 C).
template <typename T>
__global__ __launch_bounds__(NT) void cat2_add_kernel(const T* gl, long ldl, const T* gr, long ldr, const T* gs, long lds, T* out, long ldo,
                                                      long P, int half) {
  const int HV = half >> 3, CV = 2 * HV;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    float a[8], b[8];
    if (cv < HV) V8<T>::load(gl + p * ldl + cv * 8, a);
    else V8<T>::load(gr + p * ldr + (cv - HV) * 8, a);
    if (gs) {
      V8<T>::load(gs + p * lds + cv * 8, b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
    }
    V8<T>::store(out + p * ldo + cv * 8, a);
  }
}

// gradient of a channel slice x[:, :C] of a tensor that was computed with CP >= C channels (a ragged class count padded to whole
// 8-channel vectors): out[p][c] = c < C ? g[p][c] : 0, one pass (autograd's own route: a zero-filled tensor + a strided copy).
// g has an arbitrary (possibly odd) pitch, so it is read element by element; out is written as whole vectors.
template <typename T>
__global__ __launch_bounds__(NT) void pad_channels_kernel(const T* g, long ldg, int C, T* out, long ldo, int CP, long P) {
  const int CV = CP >> 3;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int cv = (int)(i % CV);
    const long p = i / CV;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv * 8 + j;
      v[j] = c < C ? (float)g[p * ldg + c] : 0.f;
    }
    V8<T>::store(out + p * ldo + cv * 8, v);
  }
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}
inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

extern "C" {

int tss_tensor_stats(const void* z, long ldz, long P, int C, double* stats, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(P >= 0 && C > 0 && ldz >= C && z && stats, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, (double)P * C * esz(dtype), 0);
  const bool vec = (C % 8) == 0 && C <= NT * 8 / 2 && (ldz % 8) == 0 && tss::aligned16(z) && P >= 4 * TSS_STAT_SLABS;
  if (vec && dtype == TSS_BF16)
    hipLaunchKernelGGL(tensor_stats_vec_kernel<bf16_t>, dim3(TSS_STAT_SLABS), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)z, ldz, P, C, stats);
  else if (vec)
    hipLaunchKernelGGL(tensor_stats_vec_kernel<float>, dim3(TSS_STAT_SLABS), dim3(NT), 0, (hipStream_t)stream, (const float*)z, ldz, P, C, stats);
  else if (dtype == TSS_BF16)
    hipLaunchKernelGGL(tensor_stats_kernel<bf16_t>, dim3(TSS_STAT_SLABS), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)z, ldz, P, C, stats);
  else
    hipLaunchKernelGGL(tensor_stats_kernel<float>, dim3(TSS_STAT_SLABS), dim3(NT), 0, (hipStream_t)stream, (const float*)z, ldz, P, C, stats);
  return tss::check_last("tensor_stats");
}

int tss_bn_bwd_apply(const void* e, long lde, const void* z, long ldz, const float* ga, const float* gb, const float* gce,
                     const float* gmu, void* dz, long lddz, long P, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (lde % 8) == 0 && lde >= C && (lddz % 8) == 0 && lddz >= C && ga && e && dz, TSS_ERR_SHAPE);
  TSS_REQUIRE(!gb || (z && gce && gmu && (ldz % 8) == 0 && ldz >= C), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(dz) && (!gb || tss::aligned16(z)), TSS_ERR_ALIGN);
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)P * C * esz(dtype) * (gb ? 3 : 2), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)e, lde,
                       (const bf16_t*)z, ldz, ga, gb, gce, gmu, (bf16_t*)dz, lddz, P, C);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const float*)e, lde,
                       (const float*)z, ldz, ga, gb, gce, gmu, (float*)dz, lddz, P, C);
  return tss::check_last("bn_bwd_apply");
}

int tss_pool_concat_fwd(const void* y1, long ld1, const float* bias, int N1, const void* x, int x_f32, long sxb, long sxc, long sxh,
                        long sxw, int Cin, void* z, long ldz, int B, int Hin, int Win, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N1 >= 0 && Cin > 0 && B >= 0 && Hin > 0 && Win > 0 && (Hin % 2) == 0 && (Win % 2) == 0 && ldz >= N1 + Cin && x && z &&
              (N1 == 0 || (y1 && ld1 >= N1)), TSS_ERR_SHAPE);
  PoolArgs g = {x, x_f32, sxb, sxc, sxh, sxw, Cin, N1, B, Hin / 2, Win / 2};
  const long total = (long)B * g.Ho * g.Wo * (N1 + Cin);
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, (double)B * g.Ho * g.Wo * (2.0 * N1 + 5.0 * Cin) * esz(dtype), 0);
  // an NHWC activation of the output's dtype, whole vectors everywhere: the vectorised kernel
  const bool vec = (x_f32 != 0) == (dtype == TSS_F32) && sxc == 1 && (N1 % 8) == 0 && (Cin % 8) == 0 && (ld1 % 8) == 0 && (ldz % 8) == 0 &&
                   (sxb % 8) == 0 && (sxh % 8) == 0 && (sxw % 8) == 0 && tss::aligned16(x) && tss::aligned16(z) && (N1 == 0 || tss::aligned16(y1));
  if (vec) {
    const long tv = total / 8;
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(pool_concat_fwd_vec_kernel<bf16_t>, dim3(grid_for(tv)), dim3(NT), 0, (hipStream_t)stream, g, (const bf16_t*)y1, ld1, bias,
                         (bf16_t*)z, ldz);
    else
      hipLaunchKernelGGL(pool_concat_fwd_vec_kernel<float>, dim3(grid_for(tv)), dim3(NT), 0, (hipStream_t)stream, g, (const float*)y1, ld1, bias,
                         (float*)z, ldz);
    return tss::check_last("pool_concat_fwd");
  }
  const bool img = x_f32 != 0 && ((N1 + Cin) % 8) == 0 && (ldz % 8) == 0 && tss::aligned16(z) && N1 > 0 && (ld1 % 8) == 0 && ld1 >= (N1 + 7) / 8 * 8 &&
                   tss::aligned16(y1);
  if (img) {
    const long tv = total / 8;
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(pool_concat_fwd_img_kernel<bf16_t>, dim3(grid_for(tv)), dim3(NT), 0, (hipStream_t)stream, g, (const bf16_t*)y1, ld1, bias,
                         (bf16_t*)z, ldz);
    else
      hipLaunchKernelGGL(pool_concat_fwd_img_kernel<float>, dim3(grid_for(tv)), dim3(NT), 0, (hipStream_t)stream, g, (const float*)y1, ld1, bias,
                         (float*)z, ldz);
    return tss::check_last("pool_concat_fwd");
  }
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(pool_concat_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g, (const bf16_t*)y1, ld1, bias,
                       (bf16_t*)z, ldz);
  else
    hipLaunchKernelGGL(pool_concat_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g, (const float*)y1, ld1, bias,
                       (float*)z, ldz);
  return tss::check_last("pool_concat_fwd");
}

int tss_pool_concat_bwd(const void* dz, long lddz, int N1, const void* x, int x_f32, long sxb, long sxc, long sxh, long sxw, int Cin,
                        void* dx, long lddx, int B, int Hin, int Win, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N1 >= 0 && Cin > 0 && B >= 0 && Hin > 0 && Win > 0 && (Hin % 2) == 0 && (Win % 2) == 0 && lddz >= N1 + Cin && lddx >= Cin &&
              x && dz && dx, TSS_ERR_SHAPE);
  PoolArgs g = {x, x_f32, sxb, sxc, sxh, sxw, Cin, N1, B, Hin / 2, Win / 2};
  const long total = (long)B * g.Ho * g.Wo * Cin;
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)total * 9.0 * esz(dtype), 0);
  const bool vec = (x_f32 != 0) == (dtype == TSS_F32) && sxc == 1 && (N1 % 8) == 0 && (Cin % 8) == 0 && (lddz % 8) == 0 && (lddx % 8) == 0 &&
                   (sxb % 8) == 0 && (sxh % 8) == 0 && (sxw % 8) == 0 && tss::aligned16(x) && tss::aligned16(dz) && tss::aligned16(dx);
  if (vec) {
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(pool_concat_bwd_vec_kernel<bf16_t>, dim3(grid_for(total / 8)), dim3(NT), 0, (hipStream_t)stream, g, (const bf16_t*)dz, lddz,
                         (bf16_t*)dx, lddx);
    else
      hipLaunchKernelGGL(pool_concat_bwd_vec_kernel<float>, dim3(grid_for(total / 8)), dim3(NT), 0, (hipStream_t)stream, g, (const float*)dz, lddz,
                         (float*)dx, lddx);
    return tss::check_last("pool_concat_bwd");
  }
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(pool_concat_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g, (const bf16_t*)dz, lddz,
                       (bf16_t*)dx, lddx);
  else
    hipLaunchKernelGGL(pool_concat_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g, (const float*)dz, lddz,
                       (float*)dx, lddx);
  return tss::check_last("pool_concat_bwd");
}

int tss_rows_slices(int B, long HW) {       // pixel slices per image of tss_mul_addrows_bwd (workspace = B * slices * C floats)
  if (B <= 0 || HW <= 0) return 1;
  long s = (1024 + B - 1) / B;
  const long cap = (HW + 63) / 64;
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : s);
}

int tss_mul_addrows_fwd(const void* u, long ldu, const void* a, long lda, const void* r, long ldr, void* out, long ldo, int B, long HW,
                        int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldu % 8) == 0 && ldu >= C && (lda % 8) == 0 && lda >= C && (ldr % 8) == 0 && ldr >= C &&
              (ldo % 8) == 0 && ldo >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(u) && tss::aligned16(a) && tss::aligned16(r) && tss::aligned16(out), TSS_ERR_ALIGN);
  const long P = (long)B * HW;
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, 3.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(mul_addrows_fwd_kernel<bf16_t>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)u, ldu,
                       (const bf16_t*)a, lda, (const bf16_t*)r, ldr, (bf16_t*)out, ldo, HW, P, C);
  else
    hipLaunchKernelGGL(mul_addrows_fwd_kernel<float>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const float*)u, ldu,
                       (const float*)a, lda, (const float*)r, ldr, (float*)out, ldo, HW, P, C);
  return tss::check_last("mul_addrows_fwd");
}

int tss_mul_addrows_bwd(const void* g, long ldg, const void* u, long ldu, const void* a, long lda, void* du, long lddu, void* da,
                        long ldda, void* dr, long lddr, float* ws, int B, long HW, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (C / 8) <= NT && (ldg % 8) == 0 && ldg >= C && (ldu % 8) == 0 && ldu >= C && (lda % 8) == 0 && lda >= C &&
              (lddu % 8) == 0 && lddu >= C && (ldda % 8) == 0 && ldda >= C && lddr >= C && ws && dr, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(g) && tss::aligned16(u) && tss::aligned16(a) && tss::aligned16(du) && tss::aligned16(da), TSS_ERR_ALIGN);
  if ((long)B * HW == 0) return TSS_OK;
  const int S = tss_rows_slices(B, HW);
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, 5.0 * B * HW * C * esz(dtype), 0);
#define TSS_MAR_BWD(TT)                                                                                                              \
  hipLaunchKernelGGL(mul_addrows_bwd_kernel<TT>, dim3(B * S), dim3(NT), 0, (hipStream_t)stream, (const TT*)g, ldg, (const TT*)u, ldu,  \
                     (const TT*)a, lda, (TT*)du, lddu, (TT*)da, ldda, ws, HW, C, S);                                                 \
  hipLaunchKernelGGL(rows_reduce_kernel<TT>, dim3((B * C + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, ws, (TT*)dr, lddr, B, C, S)
  if (dtype == TSS_BF16) { TSS_MAR_BWD(bf16_t); } else { TSS_MAR_BWD(float); }
#undef TSS_MAR_BWD
  return tss::check_last("mul_addrows_bwd");
}

int tss_cat2_add(const void* gl, long ldl, const void* gr, long ldr, const void* gs, long lds, void* out, long ldo, long P, int half,
                 int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(half > 0 && (half % 8) == 0 && (ldl % 8) == 0 && ldl >= half && (ldr % 8) == 0 && ldr >= half && (ldo % 8) == 0 && ldo >= 2 * half &&
              gl && gr && out && (!gs || ((lds % 8) == 0 && lds >= 2 * half)), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(gl) && tss::aligned16(gr) && tss::aligned16(out) && (!gs || tss::aligned16(gs)), TSS_ERR_ALIGN);
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)P * half * (gs ? 6.0 : 4.0) * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(cat2_add_kernel<bf16_t>, dim3(grid_for(P * (half / 4))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)gl, ldl,
                       (const bf16_t*)gr, ldr, (const bf16_t*)gs, lds, (bf16_t*)out, ldo, P, half);
  else
    hipLaunchKernelGGL(cat2_add_kernel<float>, dim3(grid_for(P * (half / 4))), dim3(NT), 0, (hipStream_t)stream, (const float*)gl, ldl,
                       (const float*)gr, ldr, (const float*)gs, lds, (float*)out, ldo, P, half);
  return tss::check_last("cat2_add");
}

int tss_pad_channels(const void* g, long ldg, int C, void* out, long ldo, int CP, long P, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && CP >= C && (CP % 8) == 0 && ldg >= C && (ldo % 8) == 0 && ldo >= CP && g && out, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(out), TSS_ERR_ALIGN);
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)P * (C + CP) * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(pad_channels_kernel<bf16_t>, dim3(grid_for(P * (CP / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)g, ldg, C,
                       (bf16_t*)out, ldo, CP, P);
  else
    hipLaunchKernelGGL(pad_channels_kernel<float>, dim3(grid_for(P * (CP / 8))), dim3(NT), 0, (hipStream_t)stream, (const float*)g, ldg, C,
                       (float*)out, ldo, CP, P);
  return tss::check_last("pad_channels");
}

int tss_scale_rows(const void* x, long ldx, const float* m, void* out, long ldo, int B, long HW, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldx % 8) == 0 && ldx >= C && (ldo % 8) == 0 && ldo >= C && m, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(out), TSS_ERR_ALIGN);
  const long P = (long)B * HW;
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, 2.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, m,
                       (bf16_t*)out, ldo, HW, P, C);
  else
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(grid_for(P * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const float*)x, ldx, m,
                       (float*)out, ldo, HW, P, C);
  return tss::check_last("scale_rows");
}

}  // extern "C"

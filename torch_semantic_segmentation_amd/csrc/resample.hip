// Bilinear resampling (align_corners=True everywhere on the hot path) and adaptive average pooling.
//
// Index arithmetic follows torch's upsample_bilinear2d exactly: scale = (in-1)/(out-1) in f32 (0 when
// out == 1), src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
// Backward kernels are gathers (no atomics, deterministic): every input pixel re-evaluates the forward
// index function over the window of outputs that can touch it, so forward and backward agree bit for
// bit on which outputs a source pixel feeds.
#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------- NHWC <-> NHWC
template <typename T>
__global__ __launch_bounds__(NT) void bilinear_nhwc_fwd_kernel(const T* x, long ldx, T* y, long ldy, int B, int Hin,
                                                               int Win, int Hout, int Wout, int C) {
  const int CV = C / 8;
  const long total = (long)B * Hout * Wout * CV;
  const float sy = ac_scale(Hin, Hout), sx = ac_scale(Win, Wout);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ox = (int)(p % Wout); p /= Wout;
    const int oy = (int)(p % Hout);
    const long b = p / Hout;
    const Tap ty = ac_tap(sy, oy, Hin), tx = ac_tap(sx, ox, Win);
    const T* base = x + (b * Hin * (long)Win) * ldx + cv * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    V8<T>::load(base + ((long)ty.i0 * Win + tx.i0) * ldx, v00);
    V8<T>::load(base + ((long)ty.i0 * Win + tx.i1) * ldx, v01);
    V8<T>::load(base + ((long)ty.i1 * Win + tx.i0) * ldx, v10);
    V8<T>::load(base + ((long)ty.i1 * Win + tx.i1) * ldx, v11);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = ty.l0 * (tx.l0 * v00[j] + tx.l1 * v01[j]) + ty.l1 * (tx.l0 * v10[j] + tx.l1 * v11[j]);
    V8<T>::store(y + ((b * Hout + oy) * (long)Wout + ox) * ldy + cv * 8, o);
  }
}

// Backward is separable and done as two gathers through an f32 workspace [B][Hin][Wout][C]:
//   rows : tmp[b][iy][ox][:] = sum_oy wy(oy, iy) * dy[b][oy][ox][:]      (coalesced 16-byte loads down a column)
//   cols : dx [b][iy][ix][:] = sum_ox wx(ox, ix) * tmp[b][iy][ox][:]
// A direct 2-D gather would give a source pixel of a 1x1 pooled map (pyramid pooling) a serial loop over every
// output pixel; the two passes keep every loop at O(scale) or O(output extent) with ~Wout x more threads.
template <typename T>
__global__ __launch_bounds__(NT) void bilinear_nhwc_bwd_rows_kernel(const T* dy, long lddy, float* tmp, int B, int Hin,
                                                                    int Hout, int Wout, int C) {
  const int CV = C / 8;
  const long total = (long)B * Hin * Wout * CV;
  const float sy = ac_scale(Hin, Hout);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ox = (int)(p % Wout); p /= Wout;
    const int iy = (int)(p % Hin);
    const long b = p / Hin;
    int lo, hi;
    ac_window(sy, iy, Hout, &lo, &hi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // 4 window rows per trip, every load issued before the first use (a `continue` on a zero weight made each
    // load its own dependent round trip: 32-64 of them for the 1-bin pyramid level); weight 0 just multiplies through
    for (int o0 = lo; o0 <= hi; o0 += 4) {
      typename V8<T>::Raw raw[4];
      float wy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int oy = o0 + u <= hi ? o0 + u : hi;
        wy[u] = o0 + u <= hi ? ac_weight(sy, oy, Hin, iy) : 0.f;
        raw[u] = V8<T>::load_raw(dy + ((b * Hout + oy) * (long)Wout + ox) * lddy + cv * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        V8<T>::unpack(raw[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy[u] != 0.f ? wy[u] * v[j] : 0.f;
      }
    }
    V8<float>::store(tmp + ((b * Hin + iy) * (long)Wout + ox) * C + cv * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void bilinear_nhwc_bwd_cols_kernel(const float* tmp, T* dx, long lddx, int B, int Hin,
                                                                    int Win, int Wout, int C) {
  const int CV = C / 8;
  const long total = (long)B * Hin * Win * CV;
  const float sx = ac_scale(Win, Wout);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ix = (int)(p % Win); p /= Win;
    const int iy = (int)(p % Hin);
    const long b = p / Hin;
    int lo, hi;
    ac_window(sx, ix, Wout, &lo, &hi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int o0 = lo; o0 <= hi; o0 += 4) {
      float v[4][8], wx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ox = o0 + u <= hi ? o0 + u : hi;
        wx[u] = o0 + u <= hi ? ac_weight(sx, ox, Win, ix) : 0.f;
        V8<float>::load(tmp + ((b * Hin + iy) * (long)Wout + ox) * C + cv * 8, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wx[u] != 0.f ? wx[u] * v[u][j] : 0.f;
    }
    V8<T>::store(dx + ((b * Hin + iy) * (long)Win + ix) * lddx + cv * 8, acc);
  }
}

// ---------------------------------------------------------------------------- planar NCHW -> NCHW (image downscale)
template <typename TI, typename TO>
__global__ __launch_bounds__(NT) void bilinear_planar_fwd_kernel(const TI* x, TO* y, long planes, int Hin, int Win,
                                                                 int Hout, int Wout) {
  const long total = planes * Hout * Wout;
  const float sy = ac_scale(Hin, Hout), sx = ac_scale(Win, Wout);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wout);
    long p = i / Wout;
    const int oy = (int)(p % Hout);
    const long pl = p / Hout;
    const Tap ty = ac_tap(sy, oy, Hin), tx = ac_tap(sx, ox, Win);
    const TI* base = x + pl * Hin * (long)Win;
    const float v00 = (float)base[(long)ty.i0 * Win + tx.i0], v01 = (float)base[(long)ty.i0 * Win + tx.i1];
    const float v10 = (float)base[(long)ty.i1 * Win + tx.i0], v11 = (float)base[(long)ty.i1 * Win + tx.i1];
    y[i] = (TO)(ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11));
  }
}

// ---------------------------------------------------------------------------- logits head: NHWC(low) -> NCHW planes
// One thread = 8 consecutive output x of one output row, for a GROUP of 8 classes: the taps are computed once,
// the <= 3 source columns x 2 rows are read as 16-byte channel vectors (8 classes each), and every class plane
// gets one 16-byte store.  (Per-class threads re-did the tap arithmetic 19x and issued 2-byte gathers.)
template <typename T, int NCOLS>   // NCOLS = source columns the 8 outputs of a lane can touch: 3 (>= x8) or 4 (>= ~x3.5)
__global__ __launch_bounds__(NT) void upsample_head_fwd_kernel(const T* low, long ldl, T* y, int B, int N, int h, int w,
                                                               int H, int W) {
  const int W8 = W / 8;
  const int NG = (N + 7) / 8;                       // class groups; ldl >= NG*8 (host-checked)
  const long total = (long)B * NG * H * W8;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(i % W8);
    long p = i / W8;
    const int oy = (int)(p % H); p /= H;
    const int ng = (int)(p % NG);
    const long b = p / NG;
    const Tap ty = ac_tap(sy, oy, h);
    const Tap t0 = ac_tap(sx, xg * 8, w);
    const int c0 = t0.i0;                           // the 8 outputs touch source columns c0 .. c0+3 at most (host-checked)
    const T* r0 = low + ((b * h + ty.i0) * (long)w) * ldl + ng * 8;
    const T* r1 = low + ((b * h + ty.i1) * (long)w) * ldl + ng * 8;
    float col[NCOLS][8];                            // vertically blended source columns, 8 classes each
#pragma unroll
    for (int k = 0; k < NCOLS; ++k) {
      const int cx = (c0 + k < w) ? c0 + k : w - 1;
      float a[8], bb[8];
      V8<T>::load(r0 + (long)cx * ldl, a);
      V8<T>::load(r1 + (long)cx * ldl, bb);
#pragma unroll
      for (int q = 0; q < 8; ++q) col[k][q] = ty.l0 * a[q] + ty.l1 * bb[q];
    }
    float o[8][8];                                  // [class][pixel]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const Tap tx = ac_tap(sx, xg * 8 + j, w);
      const int d0 = tx.i0 - c0, d1 = tx.i1 - c0;   // 0..3
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v0 = d0 == 0 ? col[0][q] : (d0 == 1 ? col[1][q] : col[2][q]);
        float v1 = d1 == 0 ? col[0][q] : (d1 == 1 ? col[1][q] : col[2][q]);
        if (NCOLS == 4) { v0 = d0 == 3 ? col[NCOLS - 1][q] : v0; v1 = d1 == 3 ? col[NCOLS - 1][q] : v1; }
        o[q][j] = tx.l0 * v0 + tx.l1 * v1;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int n = ng * 8 + q;
      if (n < N) V8<T>::store(y + ((b * N + n) * (long)H + oy) * W + xg * 8, o[q]);
    }
  }
}

// General fallback (any scale): one thread = 8 consecutive output x of one (b, class, output row), scalar gathers.
template <typename T>
__global__ __launch_bounds__(NT) void upsample_head_fwd_generic_kernel(const T* low, long ldl, T* y, int B, int N, int h, int w,
                                                               int H, int W) {
  const int W8 = W / 8;
  const long total = (long)B * N * H * W8;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(i % W8);
    long p = i / W8;
    const int oy = (int)(p % H); p /= H;
    const int n = (int)(p % N);
    const long b = p / N;
    const Tap ty = ac_tap(sy, oy, h);
    const T* r0 = low + ((b * h + ty.i0) * (long)w) * ldl + n;
    const T* r1 = low + ((b * h + ty.i1) * (long)w) * ldl + n;
    float o[8];
    int cached = -1;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;  // column i0 (rows 0,1) and column i1 (rows 0,1)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const Tap tx = ac_tap(sx, xg * 8 + j, w);
      if (tx.i0 != cached) {
        a0 = (float)r0[(long)tx.i0 * ldl]; a1 = (float)r1[(long)tx.i0 * ldl];
        b0 = (float)r0[(long)tx.i1 * ldl]; b1 = (float)r1[(long)tx.i1 * ldl];
        cached = tx.i0;
      }
      o[j] = ty.l0 * (tx.l0 * a0 + tx.l1 * b0) + ty.l1 * (tx.l0 * a1 + tx.l1 * b1);
    }
    V8<T>::store(y + ((b * N + n) * (long)H + oy) * W + xg * 8, o);
  }
}

// backward pass 1 (rows): tmp[b][n][iy][ox] = sum_oy wy(oy, iy) * dy[b][n][oy][ox] * gscale
template <typename T>
__global__ __launch_bounds__(NT) void upsample_head_bwd_rows_kernel(const T* dy, float* tmp, const float* gscale,
                                                                    long planes, int h, int H, int W) {
  const int W8 = W / 8;
  const long total = planes * h * W8;
  const float sy = ac_scale(h, H);
  const float gs = gscale ? *gscale : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(i % W8);
    long p = i / W8;
    const int iy = (int)(p % h);
    const long pl = p / h;
    int lo, hi;
    ac_window(sy, iy, H, &lo, &hi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = lo; oy <= hi; ++oy) {
      const float wy = ac_weight(sy, oy, h, iy);
      if (wy == 0.f) continue;
      float v[8];
      V8<T>::load(dy + (pl * H + oy) * (long)W + xg * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wy * v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= gs;
    V8<float>::store(tmp + (pl * h + iy) * (long)W + xg * 8, acc);
  }
}
// backward pass 2 (columns): dlow[b][iy][ix][n] = sum_ox wx(ox, ix) * tmp[b][n][iy][ox]
template <typename T>
__global__ __launch_bounds__(NT) void upsample_head_bwd_cols_kernel(const float* tmp, T* dlow, long ldl, int B, int N,
                                                                    int h, int w, int W) {
  const long total = (long)B * N * h * w;
  const float sx = ac_scale(w, W);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % w);
    long p = i / w;
    const int iy = (int)(p % h); p /= h;
    const int n = (int)(p % N);
    const long b = p / N;
    int lo, hi;
    ac_window(sx, ix, W, &lo, &hi);
    const float* row = tmp + ((b * N + n) * (long)h + iy) * W;
    float acc = 0.f;
    for (int ox = lo; ox <= hi; ++ox) acc += ac_weight(sx, ox, w, ix) * row[ox];
    dlow[((b * h + iy) * (long)w + ix) * ldl + n] = (T)acc;
  }
}

// ---------------------------------------------------------------------------- adaptive average pool (NHWC)
__device__ __forceinline__ int pool_start(int i, int n, int bins) { return (int)(((long)i * n) / bins); }
__device__ __forceinline__ int pool_end(int i, int n, int bins) { return (int)((((long)i + 1) * n + bins - 1) / bins); }

template <typename T>
__global__ __launch_bounds__(NT) void adaptive_pool_fwd_kernel(const T* x, long ldx, T* y, long ldy, int H, int W, int C,
                                                               int bins, int CV, int NPL) {
  __shared__ float red[NT * 8];
  const int cell = blockIdx.x % (bins * bins);
  const long b = blockIdx.x / (bins * bins);
  const int bi = cell / bins, bj = cell % bins;
  const int y0 = pool_start(bi, H, bins), y1 = pool_end(bi, H, bins);
  const int x0 = pool_start(bj, W, bins), x1 = pool_end(bj, W, bins);
  const int tid = threadIdx.x, cg = tid % CV, pl = tid / CV;
  const bool active = pl < NPL;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int ww = x1 - x0, n = (y1 - y0) * ww;
  if (active) {
    // 8 independent loads in flight per lane (a one-bin pool walks 2048 pixels with 16 pixel lanes: 128 dependent
    // round trips otherwise)
    for (int k0 = pl; k0 < n; k0 += NPL * 8) {
      typename V8<T>::Raw raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * NPL;
        const int kk = k < n ? k : pl;
        const int yy = y0 + kk / ww, xx = x0 + kk % ww;
        raw[u] = V8<T>::load_raw(x + ((b * H + yy) * (long)W + xx) * ldx + cg * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float v[8];
        V8<T>::unpack(raw[u], v);
        const bool on = k0 + u * NPL < n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += on ? v[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl * C + cg * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < NPL; ++q) s += red[q * C + c];
    y[(b * bins * bins + cell) * ldy + c] = (T)(s / (float)n);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void adaptive_pool_bwd_kernel(const T* dy, long lddy, T* dx, long lddx, int B, int H,
                                                               int W, int C, int bins) {
  const int CV = C / 8;
  const long total = (long)B * H * W * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long b = p / H;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int bi = 0; bi < bins; ++bi) {
      const int y0 = pool_start(bi, H, bins), y1 = pool_end(bi, H, bins);
      if (yy < y0 || yy >= y1) continue;
      for (int bj = 0; bj < bins; ++bj) {
        const int x0 = pool_start(bj, W, bins), x1 = pool_end(bj, W, bins);
        if (xx < x0 || xx >= x1) continue;
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        float v[8];
        V8<T>::load(dy + ((b * bins + bi) * (long)bins + bj) * lddy + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j] * inv;
      }
    }
    V8<T>::store(dx + ((b * H + yy) * (long)W + xx) * lddx + cv * 8, acc);
  }
}

// ---------------------------------------------------------------------------- pyramid pooling, all arms per launch
// PyramidPoolingModule (TSS/models/fastscnn.py:101-123) is four arms of pool -> 1x1 conv -> BN -> ReLU -> upsample over a
// 32 x 64 map: 53 launches of 4-11 us per step when every arm runs the generic operators one by one (~0.4 ms for 0.5 GMAC).
// These four kernels handle the arms' element-wise stages for ALL arms at once (tables passed by value):
//   ppm_pool_fwd    : the adaptive average pools of every arm                                   (4 launches -> 1)
//   ppm_pool_bwd    : dx = sum over arms of the pool gradients                                  (4 + 3 adds -> 1)
//   ppm_concat_fwd  : out = cat(x, upsample(relu(bn_i(raw_i)))) -- BN + ReLU applied per tap    (4 + 4 + 1 -> 1)
//   ppm_concat_bwd  : e_i = relu'(.) * upsample^T(dout slice_i) + the BatchNorm-backward sums   (8 + 4 -> 1)
constexpr int PPM_MAX = 4;
struct PpmArgs {
  const void* x; long ldx; void* out; long ldo;           // concat: x -> out[:, :C];  pool: x is the pooled-from map
  const void* raw[PPM_MAX]; long ldr[PPM_MAX];            // arm tensors [B][bins][bins][ca] (pool outputs / conv raw outputs / grads)
  void* e[PPM_MAX]; long lde[PPM_MAX];                    // concat bwd: masked gradient per arm
  const float* mean[PPM_MAX]; const float* scale[PPM_MAX]; const float* beta[PPM_MAX];
  double* bstats[PPM_MAX];
  int bins[PPM_MAX], cell0[PPM_MAX + 1];                  // cell0: prefix sums of bins^2
  int relu[PPM_MAX];
  int narms, B, H, W, C, ca;
};

// S > 1 (few images: one 3 x 2048 x 4096 inference has 50 windows, the one-bin window alone 2 MB): every window is cut into S
// row slices whose f32 partial sums go to `ws`; ppm_pool_combine_kernel adds them in slice order and divides.
template <typename T>
__global__ __launch_bounds__(NT) void ppm_pool_fwd_kernel(const PpmArgs g, int CV, int NPL, int S, float* ws) {
  __shared__ float red[NT * 8];
  const int cells = g.cell0[g.narms];
  const int sl = blockIdx.x % S;
  const int wid = blockIdx.x / S;
  const int cid = wid % cells;
  const long b = wid / cells;
  int arm = 0;
#pragma unroll
  for (int a = 1; a < PPM_MAX; ++a) if (a < g.narms && cid >= g.cell0[a]) arm = a;
  const int bins = g.bins[arm], cell = cid - g.cell0[arm];
  const T* x = reinterpret_cast<const T*>(g.x);
  T* y = reinterpret_cast<T*>(g.e[arm]);
  const int H = g.H, W = g.W, C = g.C;
  const int bi = cell / bins, bj = cell % bins;
  const int wy0 = pool_start(bi, H, bins), wy1 = pool_end(bi, H, bins);
  const int x0 = pool_start(bj, W, bins), x1 = pool_end(bj, W, bins);
  const int y0 = wy0 + (int)((long)(wy1 - wy0) * sl / S), y1 = wy0 + (int)((long)(wy1 - wy0) * (sl + 1) / S);
  const int tid = threadIdx.x, cg = tid % CV, pl = tid / CV;
  const bool active = pl < NPL;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int ww = x1 - x0, n = (y1 - y0) * ww, nwin = (wy1 - wy0) * ww;
  if (active) {
    for (int k0 = pl; k0 < n; k0 += NPL * 8) {
      typename V8<T>::Raw raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * NPL;
        const int kk = k < n ? k : 0;
        const int yy = y0 + kk / ww, xx = x0 + kk % ww;
        raw[u] = V8<T>::load_raw(x + ((b * H + yy) * (long)W + xx) * g.ldx + cg * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float v[8];
        V8<T>::unpack(raw[u], v);
        const bool on = k0 + u * NPL < n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += on ? v[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl * C + cg * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int q = 0; q < NPL; ++q) t += red[q * C + c];
    if (S == 1) y[(b * bins * bins + cell) * g.lde[arm] + c] = (T)(t / (float)nwin);
    else ws[((long)wid * S + sl) * C + c] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(128) void ppm_pool_combine_kernel(const PpmArgs g, int S, const float* ws) {
  const int cells = g.cell0[g.narms];
  const int wid = blockIdx.x, cid = wid % cells;
  const long b = wid / cells;
  int arm = 0;
#pragma unroll
  for (int a = 1; a < PPM_MAX; ++a) if (a < g.narms && cid >= g.cell0[a]) arm = a;
  const int bins = g.bins[arm], cell = cid - g.cell0[arm];
  const int bi = cell / bins, bj = cell % bins;
  const int nwin = (pool_end(bi, g.H, bins) - pool_start(bi, g.H, bins)) * (pool_end(bj, g.W, bins) - pool_start(bj, g.W, bins));
  T* y = reinterpret_cast<T*>(g.e[arm]);
  for (int c = threadIdx.x; c < g.C; c += blockDim.x) {
    // slice order, sixteen loads in flight at a time (left as a plain loop the compiler issues one load per round trip: 62 us for
    // the 256 slices of the one-window pool over the 256 x 512 map of BASELINE config 5)
    float t = 0.f;
    const float* col = ws + (long)wid * S * g.C + c;
    for (int q0 = 0; q0 < S; q0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = col[(long)(q0 + u < S ? q0 + u : q0) * g.C];
#pragma unroll
      for (int u = 0; u < 16; ++u) t += (q0 + u < S) ? v[u] : 0.f;
    }
    y[(b * bins * bins + cell) * g.lde[arm] + c] = (T)(t / (float)nwin);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void ppm_pool_bwd_kernel(const PpmArgs g) {
  const int CV = g.C / 8, H = g.H, W = g.W;
  const long total = (long)g.B * H * W * CV;
  T* dx = reinterpret_cast<T*>(g.out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long b = p / H;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int a = 0; a < g.narms; ++a) {
      const int bins = g.bins[a];
      const T* dy = reinterpret_cast<const T*>(g.raw[a]);
      for (int bi = 0; bi < bins; ++bi) {
        const int y0 = pool_start(bi, H, bins), y1 = pool_end(bi, H, bins);
        if (yy < y0 || yy >= y1) continue;
        for (int bj = 0; bj < bins; ++bj) {
          const int x0 = pool_start(bj, W, bins), x1 = pool_end(bj, W, bins);
          if (xx < x0 || xx >= x1) continue;
          const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
          float v[8];
          V8<T>::load(dy + ((b * bins + bi) * (long)bins + bj) * g.ldr[a] + cv * 8, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[j] * inv;
        }
      }
    }
    if (g.x) {      // the other consumer of the pooled-from map (the concat's own copy of x): its gradient is added here, not by a launch of its own
      float r[8];
      V8<T>::load(reinterpret_cast<const T*>(g.x) + ((b * H + yy) * (long)W + xx) * g.ldx + cv * 8, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += r[j];
    }
    V8<T>::store(dx + ((b * H + yy) * (long)W + xx) * g.ldo + cv * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void ppm_concat_fwd_kernel(const PpmArgs g) {
  const int CVx = g.C / 8, CVa = g.ca / 8, CVt = CVx + g.narms * CVa;
  const int H = g.H, W = g.W;
  const long total = (long)g.B * H * W * CVt;
  const T* x = reinterpret_cast<const T*>(g.x);
  T* out = reinterpret_cast<T*>(g.out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CVt);
    long p = i / CVt;
    const long pix = p;
    const int ox = (int)(p % W); p /= W;
    const int oy = (int)(p % H);
    const long b = p / H;
    float o[8];
    if (cv < CVx) {
      V8<T>::load(x + pix * g.ldx + cv * 8, o);
    } else {
      const int arm = (cv - CVx) / CVa, c0 = ((cv - CVx) % CVa) * 8;
      const int bins = g.bins[arm];
      const float sy = ac_scale(bins, H), sx = ac_scale(bins, W);
      const Tap ty = ac_tap(sy, oy, bins), tx = ac_tap(sx, ox, bins);
      const T* base = reinterpret_cast<const T*>(g.raw[arm]) + (b * bins * (long)bins) * g.ldr[arm] + c0;
      const long ld = g.ldr[arm];
      float v00[8], v01[8], v10[8], v11[8], mu[8], sc[8], be[8];
      V8<T>::load(base + ((long)ty.i0 * bins + tx.i0) * ld, v00);
      V8<T>::load(base + ((long)ty.i0 * bins + tx.i1) * ld, v01);
      V8<T>::load(base + ((long)ty.i1 * bins + tx.i0) * ld, v10);
      V8<T>::load(base + ((long)ty.i1 * bins + tx.i1) * ld, v11);
      const bool aff = g.scale[arm] != nullptr;
      {   // six 16-byte loads through null-safe pointers (24 scalar loads behind a branch otherwise)
        const float* safe = reinterpret_cast<const float*>(base);
        const float* pm = aff ? g.mean[arm] + c0 : safe; const float* ps = aff ? g.scale[arm] + c0 : safe;
        const float* pb = aff ? g.beta[arm] + c0 : safe;
#pragma unroll
        for (int h = 0; h < 8; h += 4) { V4<float>::load(pm + h, mu + h); V4<float>::load(ps + h, sc + h); V4<float>::load(pb + h, be + h); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = aff ? mu[j] : 0.f; sc[j] = aff ? sc[j] : 1.f; be[j] = aff ? be[j] : 0.f; }
      }
      const float lo = g.relu[arm] ? 0.f : -TSS_INF;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // the arm's activation is rounded to T where the unfused path materialised it, then interpolated
        const float a00 = V8<T>::round(fmaxf((v00[j] - mu[j]) * sc[j] + be[j], lo)), a01 = V8<T>::round(fmaxf((v01[j] - mu[j]) * sc[j] + be[j], lo));
        const float a10 = V8<T>::round(fmaxf((v10[j] - mu[j]) * sc[j] + be[j], lo)), a11 = V8<T>::round(fmaxf((v11[j] - mu[j]) * sc[j] + be[j], lo));
        o[j] = ty.l0 * (tx.l0 * a00 + tx.l1 * a01) + ty.l1 * (tx.l0 * a10 + tx.l1 * a11);
      }
    }
    V8<T>::store(out + pix * g.ldo + cv * 8, o);
  }
}

// one block per (image, arm, cell): gathers the cell's gradient from the window of output pixels that read it,
// applies the ReLU mask of the arm's activation and leaves (e, e * (raw - mean)) in the arm's slab row b * bins^2 + cell
template <typename T>
__global__ __launch_bounds__(NT) void ppm_concat_bwd_kernel(const PpmArgs g) {
  __shared__ float red[NT * 8];
  const int cells = g.cell0[g.narms];
  const int cid = blockIdx.x % cells;
  const long b = blockIdx.x / cells;
  int arm = 0;
#pragma unroll
  for (int a = 1; a < PPM_MAX; ++a) if (a < g.narms && cid >= g.cell0[a]) arm = a;
  const int bins = g.bins[arm], cell = cid - g.cell0[arm];
  const int ci = cell / bins, cj = cell % bins;
  const int H = g.H, W = g.W, ca = g.ca, CVa = ca / 8;
  const int tid = threadIdx.x, cg = tid % CVa, pl = tid / CVa, NPL = NT / CVa;
  const float sy = ac_scale(bins, H), sx = ac_scale(bins, W);
  int ylo, yhi, xlo, xhi;
  ac_window(sy, ci, H, &ylo, &yhi);
  ac_window(sx, cj, W, &xlo, &xhi);
  const int ww = xhi - xlo + 1, n = (yhi - ylo + 1) * ww;
  const T* dout = reinterpret_cast<const T*>(g.x) + g.C + arm * ca + cg * 8;   // x = the gradient of the concat buffer
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = pl; k0 < n; k0 += NPL * 4) {
    typename V8<T>::Raw raw[4];
    float wgt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * NPL;
      const int kk = k < n ? k : pl;
      const int oy = ylo + kk / ww, ox = xlo + kk % ww;
      wgt[u] = k < n ? ac_weight(sy, oy, bins, ci) * ac_weight(sx, ox, bins, cj) : 0.f;
      raw[u] = V8<T>::load_raw(dout + ((b * H + oy) * (long)W + ox) * g.ldx);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      V8<T>::unpack(raw[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wgt[u] != 0.f ? wgt[u] * v[j] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pl * ca + cg * 8 + j] = acc[j];
  __syncthreads();
  const long q = b * bins * bins + cell;
  const int rows_used = g.B * bins * bins;
  for (int c = tid; c < ca; c += blockDim.x) {
    float t = 0.f;
    for (int r = 0; r < NPL; ++r) t += red[r * ca + c];
    const float v = (float)reinterpret_cast<const T*>(g.raw[arm])[q * g.ldr[arm] + c];
    const bool aff = g.scale[arm] != nullptr;
    const float xc = v - (aff ? g.mean[arm][c] : 0.f);
    const float act = aff ? xc * g.scale[arm][c] + g.beta[arm][c] : xc;
    if (g.relu[arm] && !(act > 0.f)) t = 0.f;
    t = V8<T>::round(t);
    reinterpret_cast<T*>(g.e[arm])[q * g.lde[arm] + c] = (T)t;
    double* st = g.bstats[arm];
    if (st) {
      st[q * 2 * ca + c] = (double)t;
      st[q * 2 * ca + ca + c] = (double)t * (double)xc;
      for (long rr = q + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) { st[rr * 2 * ca + c] = 0.0; st[rr * 2 * ca + ca + c] = 0.0; }
    }
  }
}

// copy a [P][C] NHWC tensor into a channel slice of another (concat without torch.cat)
template <typename T>
__global__ __launch_bounds__(NT) void copy_nhwc_kernel(const T* x, long ldx, T* y, long ldy, long P, int C) {
  const int CV = C / 8;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / CV;
    const int cv = (int)(i - p * CV);
    float v[8];
    V8<T>::load(x + p * ldx + cv * 8, v);
    V8<T>::store(y + p * ldy + cv * 8, v);
  }
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

#define TSS_DISPATCH(dtype, KERNEL, grid, stream, ...)                                                    \
  do {                                                                                                    \
    if ((dtype) == TSS_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, __VA_ARGS__);     \
  } while (0)

extern "C" {

int tss_bilinear_nhwc_fwd(const void* x, long ldx, void* y, long ldy, int B, int Hin, int Win, int Hout, int Wout,
                          int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0 && ldx >= C && ldy >= C && Hin > 0 && Win > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  const long total = (long)B * Hout * Wout * (C / 8);
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_BILINEAR_FWD, (hipStream_t)stream, ((double)B * Hin * Win + (double)B * Hout * Wout) * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(bilinear_nhwc_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (bf16_t*)y, ldy, B, Hin, Win, Hout, Wout, C);
  else
    hipLaunchKernelGGL(bilinear_nhwc_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)x, ldx, (float*)y, ldy, B, Hin, Win, Hout, Wout, C);
  return tss::check_last("bilinear_nhwc_fwd");
}

int tss_bilinear_nhwc_bwd(const void* dy, long lddy, void* dx, long lddx, float* tmp, int B, int Hin, int Win,
                          int Hout, int Wout, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (lddy % 8) == 0 && (lddx % 8) == 0 && lddy >= C && lddx >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dy) && tss::aligned16(dx) && tss::aligned16(tmp) && tmp, TSS_ERR_ALIGN);
  const long t1 = (long)B * Hin * Wout * (C / 8), t2 = (long)B * Hin * Win * (C / 8);
  if (t1 == 0) return TSS_OK;
  hipStream_t st = (hipStream_t)stream;
  {
    tss::ProfScope prof(TSS_K_BILINEAR_BWD, st, (double)B * Hout * Wout * C * esz(dtype) + (double)B * Hin * Wout * C * 4, 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(bilinear_nhwc_bwd_rows_kernel<bf16_t>, dim3(grid_for(t1)), dim3(NT), 0, st, (const bf16_t*)dy, lddy, tmp, B, Hin, Hout, Wout, C);
    else
      hipLaunchKernelGGL(bilinear_nhwc_bwd_rows_kernel<float>, dim3(grid_for(t1)), dim3(NT), 0, st, (const float*)dy, lddy, tmp, B, Hin, Hout, Wout, C);
  }
  {
    tss::ProfScope prof(TSS_K_BILINEAR_BWD_COLS, st, (double)B * Hin * Wout * C * 4 + (double)B * Hin * Win * C * esz(dtype), 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(bilinear_nhwc_bwd_cols_kernel<bf16_t>, dim3(grid_for(t2)), dim3(NT), 0, st, tmp, (bf16_t*)dx, lddx, B, Hin, Win, Wout, C);
    else
      hipLaunchKernelGGL(bilinear_nhwc_bwd_cols_kernel<float>, dim3(grid_for(t2)), dim3(NT), 0, st, tmp, (float*)dx, lddx, B, Hin, Win, Wout, C);
  }
  return tss::check_last("bilinear_nhwc_bwd");
}

int tss_bilinear_planar_fwd(const void* x, int x_dtype, void* y, int y_dtype, long planes, int Hin, int Win,
                            int Hout, int Wout, void* stream) {
  TSS_REQUIRE((x_dtype == TSS_F32 || x_dtype == TSS_BF16) && (y_dtype == TSS_F32 || y_dtype == TSS_BF16), TSS_ERR_DTYPE);
  TSS_REQUIRE(Hin > 0 && Win > 0, TSS_ERR_SHAPE);
  const long total = planes * Hout * Wout;
  if (total == 0) return TSS_OK;
  const int grid = grid_for(total);
  tss::ProfScope prof(TSS_K_BILINEAR_PLANAR_FWD, (hipStream_t)stream,
                      (double)planes * ((double)Hin * Win * esz(x_dtype) + (double)Hout * Wout * esz(y_dtype)), 0);
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == TSS_F32 && y_dtype == TSS_F32)
    hipLaunchKernelGGL((bilinear_planar_fwd_kernel<float, float>), dim3(grid), dim3(NT), 0, s, (const float*)x, (float*)y, planes, Hin, Win, Hout, Wout);
  else if (x_dtype == TSS_F32)
    hipLaunchKernelGGL((bilinear_planar_fwd_kernel<float, bf16_t>), dim3(grid), dim3(NT), 0, s, (const float*)x, (bf16_t*)y, planes, Hin, Win, Hout, Wout);
  else if (y_dtype == TSS_F32)
    hipLaunchKernelGGL((bilinear_planar_fwd_kernel<bf16_t, float>), dim3(grid), dim3(NT), 0, s, (const bf16_t*)x, (float*)y, planes, Hin, Win, Hout, Wout);
  else
    hipLaunchKernelGGL((bilinear_planar_fwd_kernel<bf16_t, bf16_t>), dim3(grid), dim3(NT), 0, s, (const bf16_t*)x, (bf16_t*)y, planes, Hin, Win, Hout, Wout);
  return tss::check_last("bilinear_planar_fwd");
}

int tss_upsample_head_fwd(const void* low, long ldl, void* y, int B, int N, int h, int w, int H, int W,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N > 0 && ldl >= N && (W % 8) == 0 && h > 0 && w > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(y), TSS_ERR_ALIGN);
  if ((long)B * N * H * W == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_UPSAMPLE_HEAD_FWD, (hipStream_t)stream, ((double)B * N * h * w + (double)B * N * H * W) * esz(dtype), 0);
  // class-vector kernel: needs the 8 outputs of a lane within 4 source columns (7*scale_x < 2, i.e. >= ~x3.5
  // upsampling), 16-byte channel vectors (pitch covers whole groups of 8 classes) and an aligned source
  const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const bool fast = 7.f * sx < 1.99f && (ldl % 8) == 0 && ldl >= (N + 7) / 8 * 8 && tss::aligned16(low);
  if (fast) {
    const long total = (long)B * ((N + 7) / 8) * H * (W / 8);
    const bool three = 7.f * sx < 0.99f;   // x8 and up: i0 advances at most once over a lane's 8 outputs
    if (dtype == TSS_BF16) {
      if (three) hipLaunchKernelGGL((upsample_head_fwd_kernel<bf16_t, 3>), dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                                    (const bf16_t*)low, ldl, (bf16_t*)y, B, N, h, w, H, W);
      else hipLaunchKernelGGL((upsample_head_fwd_kernel<bf16_t, 4>), dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                              (const bf16_t*)low, ldl, (bf16_t*)y, B, N, h, w, H, W);
    } else {
      if (three) hipLaunchKernelGGL((upsample_head_fwd_kernel<float, 3>), dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                                    (const float*)low, ldl, (float*)y, B, N, h, w, H, W);
      else hipLaunchKernelGGL((upsample_head_fwd_kernel<float, 4>), dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                              (const float*)low, ldl, (float*)y, B, N, h, w, H, W);
    }
  } else {
    const long total = (long)B * N * H * (W / 8);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(upsample_head_fwd_generic_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                         (const bf16_t*)low, ldl, (bf16_t*)y, B, N, h, w, H, W);
    else
      hipLaunchKernelGGL(upsample_head_fwd_generic_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                         (const float*)low, ldl, (float*)y, B, N, h, w, H, W);
  }
  return tss::check_last("upsample_head_fwd");
}

int tss_upsample_head_bwd(const void* dy, const float* gscale, float* tmp /*[B*N*h*W] f32 workspace*/,
                          void* dlow, long ldl, int B, int N, int h, int w, int H, int W, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N > 0 && ldl >= N && (W % 8) == 0 && h > 0 && w > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dy) && tss::aligned16(tmp), TSS_ERR_ALIGN);
  const long planes = (long)B * N;
  const long t1 = planes * h * (W / 8), t2 = planes * h * w;
  if (t1 == 0) return TSS_OK;
  {
    tss::ProfScope prof(TSS_K_UPSAMPLE_HEAD_BWD_ROWS, (hipStream_t)stream, (double)planes * H * W * esz(dtype) + (double)planes * h * W * 4, 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(upsample_head_bwd_rows_kernel<bf16_t>, dim3(grid_for(t1)), dim3(NT), 0, (hipStream_t)stream,
                         (const bf16_t*)dy, tmp, gscale, planes, h, H, W);
    else
      hipLaunchKernelGGL(upsample_head_bwd_rows_kernel<float>, dim3(grid_for(t1)), dim3(NT), 0, (hipStream_t)stream,
                         (const float*)dy, tmp, gscale, planes, h, H, W);
  }
  {
    tss::ProfScope prof(TSS_K_UPSAMPLE_HEAD_BWD_COLS, (hipStream_t)stream, (double)planes * h * W * 4 + (double)planes * h * w * esz(dtype), 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(upsample_head_bwd_cols_kernel<bf16_t>, dim3(grid_for(t2)), dim3(NT), 0, (hipStream_t)stream,
                         tmp, (bf16_t*)dlow, ldl, B, N, h, w, W);
    else
      hipLaunchKernelGGL(upsample_head_bwd_cols_kernel<float>, dim3(grid_for(t2)), dim3(NT), 0, (hipStream_t)stream,
                         tmp, (float*)dlow, ldl, B, N, h, w, W);
  }
  return tss::check_last("upsample_head_bwd");
}

int tss_adaptive_pool_fwd(const void* x, long ldx, void* y, long ldy, int B, int H, int W, int C, int bins,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && C <= NT * 8 && (ldx % 8) == 0 && ldx >= C && ldy >= C && bins >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x), TSS_ERR_ALIGN);
  if (B == 0) return TSS_OK;
  const int CV = C / 8, NPL = NT / CV;
  const int threads = (CV * NPL + 63) / 64 * 64;
  const int grid = B * bins * bins;
  tss::ProfScope prof(TSS_K_POOL_FWD, (hipStream_t)stream, (double)B * H * W * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (bf16_t*)y, ldy, H, W, C, bins, CV, NPL);
  else
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream,
                       (const float*)x, ldx, (float*)y, ldy, H, W, C, bins, CV, NPL);
  return tss::check_last("adaptive_pool_fwd");
}

int tss_adaptive_pool_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int H, int W, int C, int bins,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (lddy % 8) == 0 && lddy >= C && (lddx % 8) == 0 && lddx >= C && bins >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dy) && tss::aligned16(dx), TSS_ERR_ALIGN);
  const long total = (long)B * H * W * (C / 8);
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_POOL_BWD, (hipStream_t)stream, (double)B * H * W * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, B, H, W, C, bins);
  else
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)dy, lddy, (float*)dx, lddx, B, H, W, C, bins);
  return tss::check_last("adaptive_pool_bwd");
}

namespace {
int ppm_fill(PpmArgs& g, int narms, const int* bins, int B, int H, int W, int C, int ca) {
  if (narms < 1 || narms > PPM_MAX || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0) return TSS_ERR_SHAPE;
  g.narms = narms; g.B = B; g.H = H; g.W = W; g.C = C; g.ca = ca;
  g.cell0[0] = 0;
  for (int a = 0; a < narms; ++a) {
    if (bins[a] < 1) return TSS_ERR_SHAPE;   // bins > H or W is legal (one-pixel windows repeat), as in torch
    g.bins[a] = bins[a];
    g.cell0[a + 1] = g.cell0[a] + bins[a] * bins[a];
  }
  for (int a = narms; a < PPM_MAX; ++a) { g.bins[a] = 1; g.cell0[a + 1] = g.cell0[narms]; }
  return TSS_OK;
}
}  // namespace

int tss_ppm_pool_slices(int B, int ncells) {   // row slices per window: > 1 only when the windows alone cannot fill the chip
  const long w = (long)B * ncells;
  if (w <= 0 || w > 128) return 1;
  long S = (1024 + w - 1) / w;
  // (one window of one image -- global average pooling of a 256 x 512 map, the image-pooling branch of an ASPP head at 2048 x 4096 --
  //  was 1.05 ms through 16 slices of 2 MB each: up to 256 slices, the kernel clamps a slice to at least one row)
  return (int)(S > 256 ? 256 : S);
}

int tss_ppm_pool_fwd(const void* x, long ldx, void* const* y, const long* ldy, const int* bins, int narms, float* ws,
                     int B, int H, int W, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C <= NT * 8 && (ldx % 8) == 0 && ldx >= C && tss::aligned16(x), TSS_ERR_SHAPE);
  PpmArgs g = {};
  const int rc = ppm_fill(g, narms, bins, B, H, W, C, C);
  if (rc) return rc;
  g.x = x; g.ldx = ldx;
  for (int a = 0; a < narms; ++a) { TSS_REQUIRE(y[a] && ldy[a] >= C, TSS_ERR_SHAPE); g.e[a] = y[a]; g.lde[a] = ldy[a]; }
  const int CV = C / 8, NPL = NT / CV;
  const int threads = (CV * NPL + 63) / 64 * 64;
  const int S = ws ? tss_ppm_pool_slices(B, g.cell0[narms]) : 1;   // ws: B * cells * S * C floats
  const int grid = B * g.cell0[narms] * S;
  tss::ProfScope prof(TSS_K_POOL_FWD, (hipStream_t)stream, (double)narms * B * H * W * C * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ppm_pool_fwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g, CV, NPL, S, ws);
  else hipLaunchKernelGGL(ppm_pool_fwd_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g, CV, NPL, S, ws);
  if (S > 1) {
    if (dtype == TSS_BF16) hipLaunchKernelGGL(ppm_pool_combine_kernel<bf16_t>, dim3(B * g.cell0[narms]), dim3(128), 0, (hipStream_t)stream, g, S, ws);
    else hipLaunchKernelGGL(ppm_pool_combine_kernel<float>, dim3(B * g.cell0[narms]), dim3(128), 0, (hipStream_t)stream, g, S, ws);
  }
  return tss::check_last("ppm_pool_fwd");
}

int tss_ppm_pool_bwd(const void* const* dy, const long* lddy, const int* bins, int narms, void* dx, long lddx,
                     const void* radd, long ldr, int B, int H, int W, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((lddx % 8) == 0 && lddx >= C && tss::aligned16(dx), TSS_ERR_SHAPE);
  TSS_REQUIRE(!radd || ((ldr % 8) == 0 && ldr >= C && tss::aligned16(radd)), TSS_ERR_SHAPE);
  PpmArgs g = {};
  const int rc = ppm_fill(g, narms, bins, B, H, W, C, C);
  if (rc) return rc;
  g.out = dx; g.ldo = lddx; g.x = radd; g.ldx = ldr;
  for (int a = 0; a < narms; ++a) {
    TSS_REQUIRE(dy[a] && (lddy[a] % 8) == 0 && lddy[a] >= C && tss::aligned16(dy[a]), TSS_ERR_SHAPE);
    g.raw[a] = dy[a]; g.ldr[a] = lddy[a];
  }
  const long total = (long)B * H * W * (C / 8);
  tss::ProfScope prof(TSS_K_POOL_BWD, (hipStream_t)stream, (double)B * H * W * C * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ppm_pool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(ppm_pool_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g);
  return tss::check_last("ppm_pool_bwd");
}

int tss_ppm_concat_fwd(const void* x, long ldx, const void* const* raw, const long* ldr, const int* bins,
                       const float* const* mean, const float* const* scale, const float* const* beta, const int* relu,
                       int narms, void* out, long ldo, int B, int H, int W, int C, int ca, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(ca > 0 && (ca % 8) == 0 && (ldx % 8) == 0 && ldx >= C && (ldo % 8) == 0 && ldo >= C + narms * ca, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(out), TSS_ERR_ALIGN);
  PpmArgs g = {};
  const int rc = ppm_fill(g, narms, bins, B, H, W, C, ca);
  if (rc) return rc;
  g.x = x; g.ldx = ldx; g.out = out; g.ldo = ldo;
  for (int a = 0; a < narms; ++a) {
    TSS_REQUIRE(raw[a] && (ldr[a] % 8) == 0 && ldr[a] >= ca && tss::aligned16(raw[a]), TSS_ERR_SHAPE);
    g.raw[a] = raw[a]; g.ldr[a] = ldr[a]; g.mean[a] = mean[a]; g.scale[a] = scale[a]; g.beta[a] = beta[a]; g.relu[a] = relu[a];
    TSS_REQUIRE((mean[a] != nullptr) == (scale[a] != nullptr) && (beta[a] != nullptr) == (scale[a] != nullptr), TSS_ERR_SHAPE);
  }
  const long total = (long)B * H * W * ((C + narms * ca) / 8);
  tss::ProfScope prof(TSS_K_BILINEAR_FWD, (hipStream_t)stream, (double)B * H * W * (2 * C + narms * ca) * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ppm_concat_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(ppm_concat_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, g);
  return tss::check_last("ppm_concat_fwd");
}

int tss_ppm_concat_bwd(const void* dout, long lddo, const void* const* raw, const long* ldr, const int* bins,
                       const float* const* mean, const float* const* scale, const float* const* beta, const int* relu,
                       double* const* bstats, void* const* e, const long* lde, int narms,
                       int B, int H, int W, int C, int ca, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(ca > 0 && (ca % 8) == 0 && ca <= 256 && (NT % (ca / 8)) == 0 && (lddo % 8) == 0 && lddo >= C + narms * ca, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dout), TSS_ERR_ALIGN);
  PpmArgs g = {};
  const int rc = ppm_fill(g, narms, bins, B, H, W, C, ca);
  if (rc) return rc;
  g.x = dout; g.ldx = lddo;
  for (int a = 0; a < narms; ++a) {
    TSS_REQUIRE(raw[a] && e[a] && ldr[a] >= ca && lde[a] >= ca, TSS_ERR_SHAPE);
    TSS_REQUIRE((long)B * bins[a] * bins[a] <= TSS_STAT_SLABS || !bstats[a], TSS_ERR_SHAPE);   // one slab row per (image, cell)
    g.raw[a] = raw[a]; g.ldr[a] = ldr[a]; g.mean[a] = mean[a]; g.scale[a] = scale[a]; g.beta[a] = beta[a]; g.relu[a] = relu[a];
    g.bstats[a] = bstats[a]; g.e[a] = e[a]; g.lde[a] = lde[a];
  }
  const int grid = B * g.cell0[narms];
  tss::ProfScope prof(TSS_K_BILINEAR_BWD, (hipStream_t)stream, (double)B * H * W * narms * ca * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ppm_concat_bwd_kernel<bf16_t>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(ppm_concat_bwd_kernel<float>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g);
  return tss::check_last("ppm_concat_bwd");
}

int tss_copy_nhwc(const void* x, long ldx, void* y, long ldy, long P, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0 && ldx >= C && ldy >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  const long total = P * (C / 8);
  if (total == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_COPY, (hipStream_t)stream, 2.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(copy_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, P, C);
  else
    hipLaunchKernelGGL(copy_nhwc_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, P, C);
  return tss::check_last("copy_nhwc");
}

int tss_upsample_head_bwd_cols(const float* tmp, void* dlow, long ldl, int B, int N, int h, int w, int W,
                               int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N > 0 && ldl >= N && h > 0 && w > 0, TSS_ERR_SHAPE);
  const long t2 = (long)B * N * h * w;
  if (t2 == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_UPSAMPLE_HEAD_BWD_COLS, (hipStream_t)stream, (double)B * N * h * W * 4 + (double)t2 * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(upsample_head_bwd_cols_kernel<bf16_t>, dim3(grid_for(t2)), dim3(NT), 0, (hipStream_t)stream,
                       tmp, (bf16_t*)dlow, ldl, B, N, h, w, W);
  else
    hipLaunchKernelGGL(upsample_head_bwd_cols_kernel<float>, dim3(grid_for(t2)), dim3(NT), 0, (hipStream_t)stream,
                       tmp, (float*)dlow, ldl, B, N, h, w, W);
  return tss::check_last("upsample_head_bwd_cols");
}

}  // extern "C"

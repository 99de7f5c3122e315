// Stem: 3x3 stride-s convolution of the NCHW image (Cin*9 <= 32 taps) into 32/any-multiple-of-8 NHWC channels.
// K = 27 is too short to be worth an MFMA tile (the im2col gather into LDS dominated): this is a direct VALU kernel.
// One lane = one output pixel: its 27 input taps are loaded once into registers (27 independent loads in flight),
// the weights sit in LDS and are read as wave-uniform broadcasts, the N outputs of the pixel leave as contiguous
// 16-byte NHWC stores, and per-channel sum / sum-of-squares go to the block's statistics slab row.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXK = 32;   // Cin * 9 padded
constexpr int MAXN = 64;

struct StemArgs {
  const void* x; int x_f32;
  const float* w; void* y; long ldy; double* stats;
  int B, Cin, Hin, Win, Hout, Wout, N, stride;
};

template <typename T, typename TX, int NCH8>  // TX = image element type, NCH8 = N / 8 output channel vectors per pixel
__global__ __launch_bounds__(NT) void stem_fwd_kernel(const StemArgs g) {
  constexpr int N = NCH8 * 8;
  __shared__ __align__(16) float wl[MAXK * N];      // [k][n]: one tap's N weights are contiguous (broadcast reads)
  __shared__ float red[NT / 64][2][N];
  const int tid = threadIdx.x;
  const int K = g.Cin * 9;
  for (int i = tid; i < MAXK * N; i += NT) {
    const int k = i / N, n = i - k * N;
    wl[i] = k < K ? g.w[n * K + k] : 0.f;
  }
  __syncthreads();
  const TX* x = reinterpret_cast<const TX*>(g.x);
  T* y = reinterpret_cast<T*>(g.y);
  float s1[N], s2[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { s1[n] = 0.f; s2[n] = 0.f; }
  const long P = (long)g.B * g.Hout * g.Wout;
  const long HWo = (long)g.Hout * g.Wout;
  const long plane = (long)g.Hin * g.Win;
  for (long p = (long)blockIdx.x * NT + tid; p < P; p += (long)gridDim.x * NT) {
    const long b = p / HWo; const long rem = p - b * HWo;
    const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
    // clamped tap offsets inside one plane + validity (shared by the channels)
    int off[9]; bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * g.stride + ky - 1;
      const bool vy = iy >= 0 && iy < g.Hin;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * g.stride + kx - 1;
        ok[ky * 3 + kx] = vy && ix >= 0 && ix < g.Win;
        off[ky * 3 + kx] = (vy ? iy : 0) * g.Win + (ix < 0 ? 0 : (ix >= g.Win ? g.Win - 1 : ix));
      }
    }
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int c = 0; c < g.Cin; ++c) {
      const TX* xp = x + (b * g.Cin + c) * plane;
      float xin[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) { const float v = (float)xp[off[t]]; xin[t] = ok[t] ? v : 0.f; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float* wk = wl + (c * 9 + t) * N;
#pragma unroll
        for (int n = 0; n < N; n += 4) {
          const float4 wv = *reinterpret_cast<const float4*>(wk + n);
          acc[n] += xin[t] * wv.x; acc[n + 1] += xin[t] * wv.y; acc[n + 2] += xin[t] * wv.z; acc[n + 3] += xin[t] * wv.w;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NCH8; ++v) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = V8<T>::round(acc[v * 8 + j]);
        s1[v * 8 + j] += o[j];
        s2[v * 8 + j] += o[j] * o[j];
      }
      V8<T>::store(y + p * g.ldy + v * 8, o);
    }
  }
  if (g.stats) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float a = wave_sum(s1[n]), b = wave_sum(s2[n]);
      if (lane == 0) { red[wave][0][n] = a; red[wave][1][n] = b; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * N; i += NT) {
      const int which = i / N, n = i - which * N;
      double s = 0.0;
      for (int wv = 0; wv < NT / 64; ++wv) s += (double)red[wv][which][n];
      g.stats[(long)blockIdx.x * 2 * N + i] = s;
      for (int r = blockIdx.x + gridDim.x; r < TSS_STAT_SLABS; r += gridDim.x) g.stats[(long)r * 2 * N + i] = 0.0;
    }
  }
}

// Weight gradient of the stem: dW[n][j] = sum_p g[p][n] * patch[p][j], g = ga*(e-ce) + gb*(y-mu).
// Tiles of 256 pixels: lane = pixel for staging (its 27 taps + its 32 gradient channels go to LDS, one HBM round
// trip per tile), then lane = (n, 4 taps) for the contraction over the tile's pixels (2 LDS reads per 4 FMAs).
// Partial [32][27] sums leave the block through its workspace row (summed by stem_wgrad_reduce_kernel).
struct StemWgradArgs {
  const void* e; long lde; const void* yraw; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  const void* x; float* ws; float* dw;
  int B, Cin, Hin, Win, Hout, Wout, stride;
};

template <typename T, typename TX>
__global__ __launch_bounds__(NT, 2) void stem_wgrad_kernel(const StemWgradArgs g) {
  constexpr int N = 32, TP = 256, KP = 28;
  __shared__ __align__(16) float Gs[TP * N];
  __shared__ __align__(16) float Xs[TP * KP];
  const int tid = threadIdx.x;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const TX* x = reinterpret_cast<const TX*>(g.x);
  const int K = g.Cin * 9;
  const long P = (long)g.B * g.Hout * g.Wout;
  const long HWo = (long)g.Hout * g.Wout;
  const long plane = (long)g.Hin * g.Win;
  const long ntiles = (P + TP - 1) / TP;
  // G staging role: 4 channel vectors per pixel, 4 pixels per thread
  const int gcv = tid & 3;
  float ca[8], cb[8], cc[8], cm[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ca[j] = g.ga ? g.ga[gcv * 8 + j] : 1.f;
    cb[j] = g.yraw ? g.gb[gcv * 8 + j] : 0.f;
    cc[j] = g.yraw ? g.gce[gcv * 8 + j] : 0.f;
    cm[j] = g.yraw ? g.gmu[gcv * 8 + j] : 0.f;
  }
  // contraction role
  const int n = tid & 31, jg = tid >> 5;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p0 = tile * TP;
    __syncthreads();
    {  // taps of pixel p0 + tid
      const long p = p0 + tid;
      const bool in = p < P;
      const long pc = in ? p : 0;
      const long b = pc / HWo; const long rem = pc - b * HWo;
      const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
      float* xr = Xs + tid * KP;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * g.stride + ky - 1;
          const bool vy = iy >= 0 && iy < g.Hin;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * g.stride + kx - 1;
            const bool ok = in && c < g.Cin && vy && ix >= 0 && ix < g.Win;
            const long off = (b * g.Cin + (c < g.Cin ? c : 0)) * plane + (vy ? iy : 0) * (long)g.Win +
                             (ix < 0 ? 0 : (ix >= g.Win ? g.Win - 1 : ix));
            const float v = (float)x[off];
            xr[c * 9 + ky * 3 + kx] = ok ? v : 0.f;
          }
        }
      }
      xr[27] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // gradient vectors: pixel (tid>>2) + 64*i, channels gcv*8..+7
      const int px = (tid >> 2) + 64 * i;
      const long p = p0 + px;
      const bool in = p < P;
      const long pc = in ? p : 0;
      float ev[8], gv[8];
      V8<T>::load(e + pc * g.lde + gcv * 8, ev);
      if (yr) {
        float yv[8];
        V8<T>::load(yr + pc * g.ldyr + gcv * 8, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = in ? ca[j] * (ev[j] - cc[j]) + cb[j] * (yv[j] - cm[j]) : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = in ? ca[j] * ev[j] : 0.f;
      }
      V8<float>::store(Gs + px * N + gcv * 8, gv);
    }
    __syncthreads();
#pragma unroll 8
    for (int p = 0; p < TP; ++p) {
      const float gp = Gs[p * N + n];
      const float4 xv = *reinterpret_cast<const float4*>(Xs + p * KP + jg * 4);
      acc[0] += gp * xv.x; acc[1] += gp * xv.y; acc[2] += gp * xv.z; acc[3] += gp * xv.w;
    }
  }
  // this block's partial dW[n][jg*4 + k]
  float* row = g.ws + (long)blockIdx.x * N * KP;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = jg * 4 + k;
    if (j < KP) {
      row[n * KP + j] = acc[k];
    }
  }
  (void)K;
}

// dW[n][j] += sum over the `rows` workspace rows (16 columns x 16 row-groups per block, 8 loads in flight per lane)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* ws, float* dw, int K, int rows) {
  constexpr int N = 32, KP = 28;
  __shared__ float part[16][17];
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);  // column over N * KP
  const int rg = threadIdx.x >> 4;
  float s = 0.f;
  if (i < N * KP) {
    for (int r0 = rg; r0 < rows; r0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = r0 + 16 * u; v[u] = r < rows ? ws[(long)r * N * KP + i] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  part[rg][threadIdx.x & 15] = s;
  __syncthreads();
  if (threadIdx.x < 16 && i < N * KP) {
    const int n = i / KP, j = i - n * KP;
    if (j < K) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += part[q][threadIdx.x];
      dw[n * K + j] += t;
    }
  }
}

}  // namespace

// returns false when the shape is outside what the direct kernel covers (caller falls back to the MFMA path)
bool tss_stem_direct_fwd(const void* x_nchw, int x_is_f32, const float* w, void* y, long ldy, double* stats,
                         int B, int Cin, int Hin, int Win, int N, int stride, int dtype, hipStream_t stream) {
  if (dtype != TSS_BF16 || Cin > 3 || N != 32) return false;
  StemArgs g = {};
  g.x = x_nchw; g.x_f32 = x_is_f32; g.w = w; g.y = y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.N = N; g.stride = stride;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  const long P = (long)B * g.Hout * g.Wout;
  const int grid = tss::persistent_blocks((P + NT - 1) / NT, TSS_STAT_SLABS);
  if (x_is_f32) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, float, 4>), dim3(grid), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, bf16_t, 4>), dim3(grid), dim3(NT), 0, stream, g);
  return true;
}

// direct stem weight gradient (bf16, N == 32, Cin <= 3); ws: [tss_stat_slabs()][32*28] f32
bool tss_stem_direct_wgrad(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                           const float* gce, const float* gmu, const void* x_nchw, int x_is_f32, float* dw, float* ws,
                           int B, int Cin, int Hin, int Win, int N, int stride, int dtype, hipStream_t stream) {
  if (dtype != TSS_BF16 || Cin > 3 || N != 32 || !ws) return false;
  StemWgradArgs g = {};
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = x_nchw; g.ws = ws; g.dw = dw;
  g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.stride = stride;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  const long P = (long)B * g.Hout * g.Wout;
  long grid = (P + 255) / 256;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;
  if (x_is_f32) hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, float>), dim3((int)grid), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, bf16_t>), dim3((int)grid), dim3(NT), 0, stream, g);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((32 * 28 + 15) / 16), dim3(256), 0, stream, ws, dw, Cin * 9, (int)grid);
  return true;
}

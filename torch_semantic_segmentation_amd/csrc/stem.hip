// Stem: 3x3 stride-2 convolution of the NCHW f32 image (Cin*9 = 27 taps) into 32 NHWC bf16 channels, forward and weight
// gradient.  One lane gathers the 27 taps of one output pixel straight from the image planes (9 dword-aligned 16-byte
// loads), both contractions run on v_mfma_f32_16x16x32_bf16 with the taps padded to one 32-deep k-step.
// History (profiles/README.md): VALU kernels first (864 FMAs per pixel forward: 196 us; an LDS-bound per-pixel outer
// product for the gradient: 368 us), then these: 120 us and 158 us.
#include "common.h"

namespace {

constexpr int NT = 256;

struct StemArgs {
  const void* x; int x_f32;
  const float* w; void* y; long ldy; double* stats;
  int B, Cin, Hin, Win, Hout, Wout, N, stride;
};

// The 27 taps (3 channels x 3 x 3, padding 1) of output pixel (b, oy, ox) of the stride-`stride` stem, zero where the
// window leaves the image.  f32 image, stride 2: the three kx taps of a (channel, row) are adjacent floats, so ONE
// 16-byte load (dword-aligned; gfx950 runs global memory in unaligned-access mode) replaces three scalar loads --
// 9 load instructions per pixel instead of 27, which is what bounds these kernels (64 lanes x 4 B scattered over
// 512 B per instruction).  The load window is clamped into the row, the taps are picked from it by position.
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
template <typename TX>
__device__ __forceinline__ void stem_taps(const TX* x, long b, int oy, int ox, bool in, int Cin, int Hin, int Win,
                                          int stride, long plane, float v[27]) {
  if (sizeof(TX) == 4 && stride == 2 && Win >= 4) {
    const int ix0 = 2 * ox - 1;
    const int base = ix0 < 0 ? 0 : (ix0 > Win - 4 ? Win - 4 : ix0);
    const int sh = ix0 - base;                      // -1 (left edge), 0, or +1 (right edge)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        const bool vy = in && c < Cin && iy >= 0 && iy < Hin;
        const float* row = reinterpret_cast<const float*>(x) + (b * Cin + (c < Cin ? c : 0)) * plane + (long)(vy ? iy : 0) * Win + base;
        const F4u f = *reinterpret_cast<const F4u*>(row);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int idx = kx + sh;                  // position of tap kx inside the window; -1 = left padding
          const float t = idx <= 0 ? f.x : (idx == 1 ? f.y : (idx == 2 ? f.z : f.w));
          v[c * 9 + ky * 3 + kx] = (vy && idx >= 0 && ix0 + kx < Win) ? t : 0.f;
        }
      }
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride + ky - 1;
      const bool vy = iy >= 0 && iy < Hin;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride + kx - 1;
        const bool ok = in && c < Cin && vy && ix >= 0 && ix < Win;
        const long off = (b * Cin + (c < Cin ? c : 0)) * plane + (vy ? iy : 0) * (long)Win + (ix < 0 ? 0 : (ix >= Win ? Win - 1 : ix));
        const float t = (float)x[off];
        v[c * 9 + ky * 3 + kx] = ok ? t : 0.f;
      }
    }
}

// MFMA form of the stem forward (the performance path).  The VALU kernel above spends 864 FMAs per pixel (VALU-bound,
// ~190 us); here the lane that owns a pixel packs its 27 taps as one bf16 row [pixel][32] in LDS (80-byte pitch:
// conflict-free 16-byte stores and fragment reads), the 32x27 weights sit in registers as two MFMA A-fragments per
// lane, and a wave needs 8 v_mfma_f32_16x16x32_bf16 for its 64 pixels.  D = W x patches^T: a lane holds 4 consecutive
// channels of one pixel (8-byte NHWC stores, DPP row sums for the statistics), exactly like pwfast_kernel.
template <typename TX>
__global__ __launch_bounds__(NT, 2) void stem_fwd_mfma_kernel(const StemArgs g) {
  typedef bf16_t T;
  constexpr int N = 32, TP = 256, PITCH = 40;
  __shared__ __align__(16) T Xp[TP * PITCH];
  __shared__ float red[NT / 64][2][N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int K = g.Cin * 9;
  bf16x8 wf[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = fq * 8 + j;
      wf[i][j] = (T)(g.w[(i * 16 + fr) * K + (k < K ? k : 0)] * (k < K ? 1.f : 0.f));
    }
  const TX* x = reinterpret_cast<const TX*>(g.x);
  T* y = reinterpret_cast<T*>(g.y);
  float st1[2][4], st2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }
  const long P = (long)g.B * g.Hout * g.Wout;
  const long HWo = (long)g.Hout * g.Wout;
  const long plane = (long)g.Hin * g.Win;
  const long ntiles = (P + TP - 1) / TP;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p0 = tile * TP;
    {
      const long p = p0 + tid;
      const bool in = p < P;
      const long pc = in ? p : 0;
      const long b = pc / HWo; const long rem = pc - b * HWo;
      const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
      float v[32];
      stem_taps<TX>(x, b, oy, ox, in, g.Cin, g.Hin, g.Win, g.stride, plane, v);
#pragma unroll
      for (int k = 27; k < 32; ++k) v[k] = 0.f;
      __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
      for (int h = 0; h < 4; ++h) V8<T>::store(Xp + tid * PITCH + h * 8, v + h * 8);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int prow = wave * 64 + m * 16 + fr;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xp + prow * PITCH + fq * 8);
      const long p = p0 + prow;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (p < P) {
          bf16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (T)d[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
          *reinterpret_cast<bf16x4*>(y + p * g.ldy + i * 16 + fq * 4) = o;
        }
      }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = row16_sum(st1[i][q]), b = row16_sum(st2[i][q]);
        if (fr == 0) { red[wave][0][i * 16 + fq * 4 + q] = a; red[wave][1][i * 16 + fq * 4 + q] = b; }
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

// MFMA form of the stem weight gradient (the performance path).  The VALU kernel above reads LDS twice per four FMAs
// (512 LDS instructions per lane per 256-pixel tile: LDS-bound, 360 us); here the contraction over the pixels is
// 8 v_mfma_f32_16x16x32_bf16 per wave per tile on operands kept pixel-major in LDS:
//   Xt[k][p] = tap k of pixel p (bf16, written by the lane that owns pixel p: conflict-free 2-byte stores)
//   Gt[n][p] = BatchNorm-backward of (e, y) for channel n of pixel p
// D = Gt x Xt^T accumulates in registers over all tiles of the block; the four waves (a 64-pixel quarter each) meet in
// LDS at the end and the block's [32][28] partial goes to its workspace row (summed by stem_wgrad_reduce_kernel).
template <typename TX>
__global__ __launch_bounds__(NT, 2) void stem_wgrad_mfma_kernel(const StemWgradArgs g) {
  typedef bf16_t T;
  constexpr int N = 32, TP = 256, KP = 28, ROW = TP + 8;   // bf16 elements per LDS row (16-byte aligned rows)
  __shared__ __align__(16) T Gt[N * ROW];
  __shared__ __align__(16) T Xt[32 * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const TX* x = reinterpret_cast<const TX*>(g.x);
  const long P = (long)g.B * g.Hout * g.Wout;
  const long HWo = (long)g.Hout * g.Wout;
  const long plane = (long)g.Hin * g.Win;
  const long ntiles = (P + TP - 1) / TP;
  for (int i = tid; i < 5 * ROW; i += NT) Xt[27 * ROW + i] = (T)0.f;   // taps 27..31 pad the MFMA k-fragment
  const int gcv = tid & 3;
  float ca[8], cb[8], cc[8];
  {
    float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(g.ga + gcv * 8 + h, v0 + h); V4<float>::load(g.gb + gcv * 8 + h, v1 + h);
      V4<float>::load(g.gce + gcv * 8 + h, v2 + h); V4<float>::load(g.gmu + gcv * 8 + h, v3 + h);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = v0[j]; cb[j] = v1[j]; cc[j] = -(v0[j] * v2[j]) - v1[j] * v3[j]; }   // g = ca*e + cb*y + cc
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p0 = tile * TP;
    __syncthreads();
    // gradient vectors first (two 16-byte loads x 4), then the 27 strided tap loads of this lane's pixel
    uint4 re[4], ry[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long p = p0 + (tid >> 2) + 64 * i;
      const long pc = p < P ? p : 0;
      re[i] = *reinterpret_cast<const uint4*>(e + pc * g.lde + gcv * 8);
      ry[i] = *reinterpret_cast<const uint4*>(yr + pc * g.ldyr + gcv * 8);
    }
    {
      const long p = p0 + tid;
      const bool in = p < P;
      const long pc = in ? p : 0;
      const long b = pc / HWo; const long rem = pc - b * HWo;
      const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
      float v[27];
      stem_taps<TX>(x, b, oy, ox, in, g.Cin, g.Hin, g.Win, g.stride, plane, v);
#pragma unroll
      for (int k = 0; k < 27; ++k) Xt[k * ROW + tid] = (T)v[k];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (tid >> 2) + 64 * i;
      const bool in = p0 + px < P;
      const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re[i]);
      const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry[i]);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float lo = ca[2 * h] * __uint_as_float(ue[h] << 16) + (cb[2 * h] * __uint_as_float(uy[h] << 16) + cc[2 * h]);
        const float hi = ca[2 * h + 1] * __uint_as_float(ue[h] & 0xffff0000u) +
                         (cb[2 * h + 1] * __uint_as_float(uy[h] & 0xffff0000u) + cc[2 * h + 1]);
        Gt[(gcv * 8 + 2 * h) * ROW + px] = (T)(in ? lo : 0.f);
        Gt[(gcv * 8 + 2 * h + 1) * ROW + px] = (T)(in ? hi : 0.f);
      }
    }
    __syncthreads();
    // this wave's 64 pixels: two k-steps of 32
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int col = wave * 64 + ks * 32 + fq * 8;
      bf16x8 gf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gf[i] = *reinterpret_cast<const bf16x8*>(Gt + (i * 16 + fr) * ROW + col);
        xf[i] = *reinterpret_cast<const bf16x8*>(Xt + (i * 16 + fr) * ROW + col);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  // the four waves' partials meet in LDS (Gt is dead): red[wave][n][k], then one thread per (n, k < 28)
  __syncthreads();
  float* red = reinterpret_cast<float*>(Gt);   // 4 * 32 * 32 floats = 16 KB <= sizeof(Gt)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 32 + i * 16 + fq * 4 + r) * 32 + j * 16 + fr] = acc[i][j][r];
  __syncthreads();
  float* row = g.ws + (long)blockIdx.x * N * KP;
  for (int i = tid; i < N * KP; i += NT) {
    const int n = i / KP, k = i - n * KP;
    row[i] = red[n * 32 + k] + red[(32 + n) * 32 + k] + red[(64 + n) * 32 + k] + red[(96 + n) * 32 + k];
  }
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
  if (x_is_f32) hipLaunchKernelGGL((stem_fwd_mfma_kernel<float>), dim3(grid), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL((stem_fwd_mfma_kernel<bf16_t>), dim3(grid), dim3(NT), 0, stream, g);
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
  if (yraw && ga && gb && gce && gmu) {   // MFMA form (train-mode BatchNorm behind the stem, the usual case)
    if (x_is_f32) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<float>), dim3((int)grid), dim3(NT), 0, stream, g);
    else hipLaunchKernelGGL((stem_wgrad_mfma_kernel<bf16_t>), dim3((int)grid), dim3(NT), 0, stream, g);
  } else if (x_is_f32) hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, float>), dim3((int)grid), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, bf16_t>), dim3((int)grid), dim3(NT), 0, stream, g);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((32 * 28 + 15) / 16), dim3(256), 0, stream, ws, dw, Cin * 9, (int)grid);
  return true;
}

// Lean bf16 kernels of the stride-2 dense 3x3 convolution (padding 1, square 32 / 64 channels): the convolution arm of
// DownsamplingBlock, TSS/models/lednet.py:126-144 and TSS/models/esnet.py:47-68 -- forward, backward-data, weight gradient.
//
// The generic implicit-GEMM kernel walks the nine taps as dependent chunks through LDS and pads every tile to 128 output channels; its
// backward-data form (a transposed gather) visits all nine taps for every input pixel although only 1, 2 or 4 of them exist for a
// given pixel parity: 3.0 ms for the 32 -> 32 layer at 8 x 512 x 1024 (forward 0.74 ms, weight gradient 1.5 ms), a quarter of LEDNet's
// train step for two layers (profiles/r04_step_kernels_lednet.txt).  These kernels use the scheme of fc1d.hip: the B operand of
// v_mfma_f32_16x16x32_bf16 is one 16-byte vector of an NHWC row, loaded straight from global memory by the lane that feeds it.
//   forward : a wave owns 16 * MT consecutive OUTPUT pixels of one output row; the nine taps are three groups (kernel rows) of three;
//             per group the lane loads pixel 2 x + dx - 1 of input row 2 y + dy - 1 (every input pixel is used by some tap: the rows
//             are fully consumed through L1), converts, and the next group's loads fly under this group's MFMAs;
//   backward: a wave owns 16 * MT INPUT pixels of one parity of one input row, so that all of them see the same 1 / 2 / 4 taps
//             (dy = y + 1 mod 2, dx likewise) and their sources are consecutive output pixels; all taps of the NEXT tile are
//             requested while this tile's MFMAs run;
//   weights : all nine taps in LDS in fragment order (18 / 72 KB), written once per block from the f32 tensor;
//   dW      : contraction over output pixels -- g and the nine strided copies of x are written once per stage as [pixel][channel]
//             rows of the dual-use LDS image (pwsweep.hip) and read back transposed; per-block rows, summed by tss_dw_reduce_many.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256;

struct ScArgs {
  int B, Hi, Wi, Ho, Wo;                              // input / output map of the FORWARD convolution
  const T* a0; long lda0; const T* a1; long lda1;     // fwd: x (a1 unused)   bwd: e, yraw (both on the OUTPUT grid)
  const float* c0; const float* c1; const float* c2; const float* c3; int a_relu;
  const float* w; long w_os, w_ks, w_t9;              // f32 weights: element (tap dy*3+dx, output o, contraction k) at w[o*w_os + k*w_ks + tap*w_t9]
  const float* bias;
  T* y; long ldy; double* stats;
  const T* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// fold the prologue constants of one 8-channel vector: MODE 0: a = relu?((x - c1) * c0 + c2); 1: g = c0 * e; 2: g = c0 (e - c2) + c1 (y - c3)
template <int MODE>
__device__ __forceinline__ void fold8(const ScArgs& g, int ch, float (&k0)[8], float (&k1)[8], float (&kadd)[8]) {
  const float* safe = g.w;
  float v0[8], v1[8], v2[8], v3[8];
  const float* p0 = g.c0 ? g.c0 + ch : safe; const float* p1 = g.c1 ? g.c1 + ch : safe;
  const float* p2 = g.c2 ? g.c2 + ch : safe; const float* p3 = (MODE == 2 && g.c3) ? g.c3 + ch : safe;
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    V4<float>::load(p0 + h, v0 + h); V4<float>::load(p1 + h, v1 + h); V4<float>::load(p2 + h, v2 + h); V4<float>::load(p3 + h, v3 + h);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float c0v = g.c0 ? v0[j] : 1.f, c1v = g.c1 ? v1[j] : 0.f, c2v = g.c2 ? v2[j] : 0.f, c3v = (MODE == 2 && g.c3) ? v3[j] : 0.f;
    k0[j] = c0v;
    if (MODE == 0) { k1[j] = 0.f; kadd[j] = c2v - c1v * c0v; }
    else if (MODE == 1) { k1[j] = 0.f; kadd[j] = 0.f; }
    else { k1[j] = c1v; kadd[j] = -(c0v * c2v) - c1v * c3v; }
  }
}

template <int MODE>
__device__ __forceinline__ uint4 convert8(const uint4& ra, const uint4& rb, bool plain, bool ok, const float (&k0)[8], const float (&k1)[8],
                                          const float (&kadd)[8], float relu_lo) {
  uint4 r = ra;
  if (!plain) {
    const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra);
    const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb);
    bf16x8 o;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float lo = blo(ua[h]) * k0[2 * h] + kadd[2 * h];
      float hi = bhi(ua[h]) * k0[2 * h + 1] + kadd[2 * h + 1];
      if (MODE == 2) { lo += blo(ub[h]) * k1[2 * h]; hi += bhi(ub[h]) * k1[2 * h + 1]; }
      if (MODE == 0) { lo = fmaxf(lo, relu_lo); hi = fmaxf(hi, relu_lo); }
      o[2 * h] = (T)lo; o[2 * h + 1] = (T)hi;
    }
    r = *reinterpret_cast<const uint4*>(&o);
  }
  if (!ok) r = make_uint4(0u, 0u, 0u, 0u);            // zero padding applies to the ACTIVATED tensor
  return r;
}

// per-block statistics -> slab row (as conv3x3.hip / fc1d.hip)
template <int C, int NF>
__device__ __forceinline__ void flush_stats(const float (&st1)[NF][4], const float (&st2)[NF][4], double* stats, float (*red)[2][C], int CR = C) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
      if (fr == 0) { red[wave][0][i * 16 + fq * 4 + q] = u; red[wave][1][i * 16 + fq * 4 + q] = w2; }
    }
  __syncthreads();
  if (tid < CR) {                                      // CR real channels (slab rows are [2 * CR] wide), C = padded fragment width
    const double a = ((double)red[0][0][tid] + (double)red[1][0][tid]) + ((double)red[2][0][tid] + (double)red[3][0][tid]);
    const double b = ((double)red[0][1][tid] + (double)red[1][1][tid]) + ((double)red[2][1][tid] + (double)red[3][1][tid]);
    const int row = blockIdx.x, rows_used = gridDim.x;
    stats[(long)row * 2 * CR + tid] = a;
    stats[(long)row * 2 * CR + CR + tid] = b;
    for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
      stats[(long)rr * 2 * CR + tid] = 0.0;
      stats[(long)rr * 2 * CR + CR + tid] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------- forward
// CK contraction channels per tap (multiples of 8, zero-padded to whole 32-wide k-steps), CN outputs (multiples of 8, padded to whole fragments):
// rectangular instances serve the INPUT gradient of nn.ConvTranspose2d (UpsamplingBlock, TSS/models/esnet.py:71-80), which is this convolution.
template <int CK, int CN, int MT>
__global__ __launch_bounds__(NT, 2) void sc2_fwd_kernel(const ScArgs g) {
  constexpr int NF = (CN + 15) / 16, C = NF * 16, KPT = (CK + 31) / 32, NKS = 3 * KPT, NCS = KPT, TWV = 16 * MT;
  extern __shared__ __align__(16) unsigned char smem[];
  uint4* Wl = reinterpret_cast<uint4*>(smem);         // [3 kernel rows][NF][NKS][64 lanes]
  __shared__ __align__(16) float Ec[C];
  __shared__ float red[4][2][C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;

  for (int e = tid; e < 3 * NF * NKS * 64; e += NT) {
    const int f = e >> 6, l = e & 63;
    const int ks = f % NKS, i = (f / NKS) % NF, dy = f / (NKS * NF);
    const int n = i * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8;
    const int dx = k / (KPT * 32), c = k - dx * (KPT * 32);
    const bool in = n < CN && c < CK;
    const float* src = g.w + (in ? (long)n * g.w_os + (long)c * g.w_ks + (long)(dy * 3 + dx) * g.w_t9 : 0);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = in ? (T)src[(long)j * g.w_ks] : (T)0.f;
    Wl[e] = *reinterpret_cast<const uint4*>(&o);
  }
  if (tid < C) Ec[tid] = (g.bias && tid < CN) ? g.bias[tid] : 0.f;

  int cch[NCS];
  float k0[NCS][8], k1[NCS][8], kadd[NCS][8];
#pragma unroll
  for (int s = 0; s < NCS; ++s) {
    cch[s] = (s * 32 + fq * 8 < CK) ? s * 32 + fq * 8 : 0;
    fold8<0>(g, cch[s], k0[s], k1[s], kadd[s]);
  }
  const bool plain = !g.c0 && !g.c1 && !g.c2 && !g.a_relu;
  const float relu_lo = g.a_relu ? 0.f : -TSS_INF;

  float st1[NF][4], st2[NF][4];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  const int tpr = (g.Wo + TWV - 1) / TWV;
  const long rows = (long)g.B * g.Ho;
  const long nblk = ((rows + 3) >> 2) * tpr;

  uint4 ra[MT][NKS];
  uint32_t okb = 0;
  // geometry of the tile whose loads are in flight (l) and of the tile being accumulated (c)
  long pin_l = 0, pco_l = 0; int x0_l = 0, yo_l = 0; bool live_l = false; long brow_l = 0;
  long pco_c = 0; int x0_c = 0; bool live_c = false;

#define SC_GEOM(BT)                                              \
  {                                                                \
    const long grp = (BT) / tpr;                                   \
    const int tx = (int)((BT) - grp * tpr);                        \
    const long by = grp * 4 + wave;                                \
    live_l = by < rows;                                            \
    const long byc = live_l ? by : rows - 1;                       \
    yo_l = (int)(byc % g.Ho);                                      \
    brow_l = (byc / g.Ho) * g.Hi;                                  \
    x0_l = tx * TWV;                                               \
    pco_l = byc * g.Wo + x0_l;                                     \
  }
#define SC_ISSUE(DY)                                                                                           \
  {                                                                                                              \
    const int yi = 2 * yo_l + (DY) - 1;                                                                          \
    const bool rowok = yi >= 0 && yi < g.Hi;                                                                     \
    pin_l = (brow_l + (rowok ? yi : 0)) * g.Wi;                                                                  \
    okb = 0;                                                                                                     \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                            \
      const int xo = x0_l + m * 16 + fr;                                                                         \
      _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                      \
        const int dx = ks / KPT;                                                                                 \
        const int xi = 2 * xo + dx - 1;                                                                          \
        const bool ok = rowok && xo < g.Wo && xi >= 0 && xi < g.Wi && (CK % 32 == 0 || (ks % KPT) * 32 + fq * 8 < CK); \
        okb |= ok ? (1u << (m * NKS + ks)) : 0u;                                                                 \
        ra[m][ks] = *reinterpret_cast<const uint4*>(g.a0 + (pin_l + (ok ? xi : 0)) * g.lda0 + cch[ks % KPT]); \
      }                                                                                                          \
    }                                                                                                            \
  }

  long bt = blockIdx.x;
  int dy = 0;
  if (bt < nblk) { SC_GEOM(bt); SC_ISSUE(0); }
  __syncthreads();

  f32x4 acc[MT][NF];
  while (bt < nblk) {
    if (dy == 0) {
      pco_c = pco_l; x0_c = x0_l; live_c = live_l;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NF; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 op[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int s = ks % KPT;
        const uint4 r = convert8<0>(ra[m][ks], ra[m][ks], plain, (okb >> (m * NKS + ks)) & 1u, k0[s], k1[s], kadd[s], relu_lo);
        op[m][ks] = *reinterpret_cast<const bf16x8*>(&r);
      }
    // the next group's (or the next tile's first group's) loads go out now
    int ndy = dy + 1; long nbt = bt;
    if (ndy == 3) { ndy = 0; nbt = bt + gridDim.x; }
    if (nbt < nblk) {
      if (ndy == 0) SC_GEOM(nbt);
      SC_ISSUE(ndy);
    }
    asm volatile("" ::: "memory");                    // weight fragments are re-read from LDS, not hoisted
    const uint4* wg = Wl + dy * (NF * NKS * 64);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const uint4 wr = wg[(i * NKS + ks) * 64 + lane];
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wr);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, op[m][ks], acc[m][i], 0, 0, 0);
      }
    if (dy == 2 && live_c) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int nl = i * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(&Ec[nl]);
        const float cbias[4] = {e0.x, e0.y, e0.z, e0.w};
        if (CN % 16 != 0 && nl >= CN) continue;          // padded output channels are never stored
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int px = m * 16 + fr;
          if (x0_c + px < g.Wo) {
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (T)(acc[m][i][q] + cbias[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
            *reinterpret_cast<bf16x4*>(g.y + (pco_c + px) * g.ldy + nl) = o;
          }
        }
      }
    }
    dy = ndy; bt = nbt;
  }
#undef SC_GEOM
#undef SC_ISSUE
  if (g.stats) flush_stats<C, NF>(st1, st2, g.stats, red, CN);
}

// ------------------------------------------------------------------------------------------------------------------- backward-data
// MODE 1: g = c0 * e   2: g = c0 (e - c2) + c1 (y - c3).  CK contraction channels (the layer's N; multiples of 8, zero-padded to whole 32-wide
// k-steps), CN outputs (the layer's input channels; multiples of 8, padded to whole 16-wide fragments): the same kernel is the FORWARD of
// nn.ConvTranspose2d(CK, CN, 3, stride=2, padding=1, output_padding=1) (UpsamplingBlock, TSS/models/esnet.py:71-80), with its bias.
template <int CK, int CN, int MODE, int MT>
__global__ __launch_bounds__(NT, 2) void sc2_bwd_kernel(const ScArgs g) {
  constexpr int NF = (CN + 15) / 16, C = NF * 16, NKT = (CK + 31) / 32, NCS = NKT, TWV = 16 * MT;
  extern __shared__ __align__(16) unsigned char smem[];
  uint4* Wl = reinterpret_cast<uint4*>(smem);         // [9 taps][NF][NKT][64 lanes]; output = input channel, contraction = n
  __shared__ __align__(16) float Ec[4][C];
  __shared__ float red[4][2][C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;

  for (int e = tid; e < 9 * NF * NKT * 64; e += NT) {
    const int f = e >> 6, l = e & 63;
    const int ks = f % NKT, i = (f / NKT) % NF, tap = f / (NKT * NF);
    const int o_ = i * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8;
    const bool in = o_ < CN && k < CK;
    const float* src = g.w + (in ? (long)o_ * g.w_os + (long)k * g.w_ks + (long)tap * g.w_t9 : 0);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = in ? (T)src[(long)j * g.w_ks] : (T)0.f;
    Wl[e] = *reinterpret_cast<const uint4*>(&o);
  }
  if (tid < C) {
    float em = 0.f, es = 1.f, eh = 0.f, eb = 0.f;
    if (tid < CN) {
      if (g.xm) { if (g.mm) em = g.mm[tid]; if (g.ms) es = g.ms[tid]; if (g.mb) eh = g.mb[tid]; }
      if (g.bias) eb = g.bias[tid];
    }
    Ec[0][tid] = em; Ec[1][tid] = es; Ec[2][tid] = eh; Ec[3][tid] = eb;
  }
  float k0[NCS][8], k1[NCS][8], kadd[NCS][8];
#pragma unroll
  for (int s = 0; s < NCS; ++s) fold8<MODE>(g, (s * 32 + fq * 8 < CK) ? s * 32 + fq * 8 : 0, k0[s], k1[s], kadd[s]);
  const bool plain = MODE == 1 && !g.c0;

  float st1[NF][4], st2[NF][4];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  const int Wh = (g.Wi + 1) >> 1;                     // pixels of one parity in an input row (parity 1 may have one less)
  const int tpr = (Wh + TWV - 1) / TWV;
  const long rows = (long)g.B * g.Hi;
  const long nblk = ((rows + 3) >> 2) * 2 * tpr;      // (row group, parity, column range)

  uint4 ra[4][MT][NKT], rb[MODE == 2 ? 4 : 1][MODE == 2 ? MT : 1][MODE == 2 ? NKT : 1];
  uint2 rxm[MT][NF];
  uint32_t okb = 0;                                   // bit (slot * MT + m)
  int nty_l = 1, ntx_l = 1, par_l = 0, j0_l = 0, yi_l = 0; long pei_l = 0; bool live_l = false;

#define SB_GEOM(BT)                                                          \
  {                                                                            \
    const long grp = (BT) / (2 * tpr);                                         \
    const int rem = (int)((BT) - grp * (2 * tpr));                             \
    par_l = rem / tpr;                                                         \
    j0_l = (rem - par_l * tpr) * TWV;                                          \
    const long by = grp * 4 + wave;                                            \
    live_l = by < rows;                                                        \
    const long byc = live_l ? by : rows - 1;                                   \
    yi_l = (int)(byc % g.Hi);                                                  \
    pei_l = byc * g.Wi;                                                        \
    nty_l = (yi_l & 1) ? 2 : 1;                                                \
    ntx_l = par_l ? 2 : 1;                                                     \
  }
  // slot (ty, tx): source output pixel (yo, xo) and tap (dy, dx)
#define SB_ISSUE()                                                                                              \
  {                                                                                                               \
    okb = 0;                                                                                                      \
    const long bimg = (pei_l / g.Wi - yi_l) / g.Hi;                                                               \
    _Pragma("unroll") for (int ty = 0; ty < 2; ++ty) {                                                           \
      if (ty < nty_l) {                                                                                           \
        const int yo = (yi_l & 1) ? (ty == 0 ? (yi_l + 1) >> 1 : (yi_l - 1) >> 1) : (yi_l >> 1);                  \
        const bool rowok = yo < g.Ho;                                                                             \
        const long prow = (bimg * g.Ho + (rowok ? yo : 0)) * g.Wo;                                                \
        _Pragma("unroll") for (int tx = 0; tx < 2; ++tx) {                                                       \
          if (tx < ntx_l) {                                                                                       \
            const int slot = ty * 2 + tx;                                                                         \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                     \
              const int j = j0_l + m * 16 + fr;                                                                   \
              const int xo = j + ((par_l && tx == 0) ? 1 : 0);                                                    \
              const bool ok = rowok && xo < g.Wo && 2 * j + par_l < g.Wi;                                         \
              okb |= ok ? (1u << (slot * MT + m)) : 0u;                                                           \
              const long q = prow + (ok ? xo : 0);                                                                \
              _Pragma("unroll") for (int ks = 0; ks < NKT; ++ks) {                                               \
                const int kc = (ks * 32 + fq * 8 < CK) ? ks * 32 + fq * 8 : 0;                                    \
                ra[slot][m][ks] = *reinterpret_cast<const uint4*>(g.a0 + q * g.lda0 + kc);                        \
                if (MODE == 2) rb[slot][m][ks] = *reinterpret_cast<const uint4*>(g.a1 + q * g.lda1 + kc);         \
              }                                                                                                   \
            }                                                                                                     \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }
#define SB_ISSUE_XM()                                                                                           \
  if (g.xm) {                                                                                                     \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                             \
      const int xi = 2 * (j0_l + m * 16 + fr) + par_l;                                                            \
      const long p = pei_l + (xi < g.Wi ? xi : 0);                                                                \
      _Pragma("unroll") for (int i = 0; i < NF; ++i)                                                             \
        rxm[m][i] = *reinterpret_cast<const uint2*>(g.xm + p * g.ldxm + ((i * 16 + fq * 4 < CN) ? i * 16 + fq * 4 : 0)); \
    }                                                                                                             \
  }

  long bt = blockIdx.x;
  if (bt < nblk) { SB_GEOM(bt); SB_ISSUE(); SB_ISSUE_XM(); }
  __syncthreads();

  for (; bt < nblk; bt += gridDim.x) {
    // ---- all taps of this tile: registers -> MFMA layout
    const int nty = nty_l, ntx = ntx_l, par = par_l, j0 = j0_l, yi = yi_l; const long pei = pei_l; const bool live = live_l;
    bf16x8 op[4][MT][NKT];
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
      if ((slot >> 1) < nty && (slot & 1) < ntx) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int ks = 0; ks < NKT; ++ks) {
            const uint4 r = convert8<MODE>(ra[slot][m][ks], rb[MODE == 2 ? slot : 0][MODE == 2 ? m : 0][MODE == 2 ? ks : 0], plain,
                                           ((okb >> (slot * MT + m)) & 1u) && (CK % 32 == 0 || ks * 32 + fq * 8 < CK), k0[ks], k1[ks], kadd[ks], 0.f);
            op[slot][m][ks] = *reinterpret_cast<const bf16x8*>(&r);
          }
      }
    }
    const long btn = bt + gridDim.x;
    if (btn < nblk) { SB_GEOM(btn); SB_ISSUE(); }

    f32x4 acc[MT][NF];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
      if ((slot >> 1) < nty && (slot & 1) < ntx) {
        const int ty = slot >> 1, tx = slot & 1;
        const int dy = (yi & 1) ? (ty == 0 ? 0 : 2) : 1;
        const int dx = par ? (tx == 0 ? 0 : 2) : 1;
        const uint4* wt = Wl + (dy * 3 + dx) * (NF * NKT * 64);
#pragma unroll
        for (int ks = 0; ks < NKT; ++ks)
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            const uint4 wr = wt[(i * NKT + ks) * 64 + lane];
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wr);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, op[slot][m][ks], acc[m][i], 0, 0, 0);
          }
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int nl = i * 16 + fq * 4;
        const float4 e1 = *reinterpret_cast<const float4*>(&Ec[0][nl]);
        const float4 e2 = *reinterpret_cast<const float4*>(&Ec[1][nl]);
        const float4 e3 = *reinterpret_cast<const float4*>(&Ec[2][nl]);
        const float4 e4 = *reinterpret_cast<const float4*>(&Ec[3][nl]);
        const float cmm[4] = {e1.x, e1.y, e1.z, e1.w}, cms[4] = {e2.x, e2.y, e2.z, e2.w}, cmb[4] = {e3.x, e3.y, e3.z, e3.w};
        const float cbi[4] = {e4.x, e4.y, e4.z, e4.w};
        if (CN % 16 != 0 && nl >= CN) continue;          // padded output channels are never stored
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int xi = 2 * (j0 + m * 16 + fr) + par;
          if (xi < g.Wi) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q] + cbi[q];
            bf16x4 o;
            if (g.xm) {
              const uint2 xr = rxm[m][i];
              const float xc[4] = {blo(xr.x) - cmm[0], bhi(xr.x) - cmm[1], blo(xr.y) - cmm[2], bhi(xr.y) - cmm[3]};
              if (g.m_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(xc[q] * cms[q] + cmb[q] > 0.f)) v[q] = 0.f;
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * xc[q]; }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
            }
            *reinterpret_cast<bf16x4*>(g.y + (pei + xi) * g.ldy + nl) = o;
          }
        }
      }
    }
    if (btn < nblk) { SB_ISSUE_XM(); }
  }
#undef SB_GEOM
#undef SB_ISSUE
#undef SB_ISSUE_XM
  if (g.stats) flush_stats<C, NF>(st1, st2, g.stats, red, CN);
}

// ------------------------------------------------------------------------------------------------------------------- weight gradient
typedef __attribute__((ext_vector_type(4))) short v4s;
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
  union { v4s h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lo);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)hi);
  return u.v;
}

struct SwArgs {
  long P; int B, Hi, Wi, Ho, Wo;                       // P = B * Ho * Wo output pixels
  const T* e; long lde; const T* y; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  const T* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu;
  float* ws;                                           // [gridDim.x][C * C * 9], torch's [N][C][3][3] order
};

template <int C, bool HASY>
__global__ __launch_bounds__(NT, C == 64 ? 1 : 2) void sc2_wgrad_kernel(const SwArgs g) {      // 64 channels: 36 accumulator fragments per wave
  constexpr int NF = C / 16, NV = C / 8, PT = C == 64 ? 32 : 64, RPP = NT / NV, NP = PT / RPP, NIMG = (10 * NV + 15) / 16;
  constexpr int BUF = NIMG * PT * 256, NKS = PT / 32, NJ = C == 64 ? 4 : 1;
  static_assert(NP == 1, "one pass per stage");
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int cv = tid % NV, r = tid / NV;

  float ca[8], cb[HASY ? 8 : 1], cc[HASY ? 8 : 1], as[8], ab[8];
  const bool gplain = !HASY && !g.ga, aplain = !g.xs && !g.xm && !g.xb && !g.x_relu;
  {
    const float* safe = reinterpret_cast<const float*>(g.e);
    float v0[8], v1[8], v2[8], v3[8], w0[8], w1[8], w2[8];
    const float* p0 = g.ga ? g.ga + cv * 8 : safe; const float* p1 = (HASY && g.gb) ? g.gb + cv * 8 : safe;
    const float* p2 = (HASY && g.gce) ? g.gce + cv * 8 : safe; const float* p3 = (HASY && g.gmu) ? g.gmu + cv * 8 : safe;
    const float* q0 = g.xs ? g.xs + cv * 8 : safe; const float* q1 = g.xm ? g.xm + cv * 8 : safe; const float* q2 = g.xb ? g.xb + cv * 8 : safe;
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(p0 + h, v0 + h); V4<float>::load(p1 + h, v1 + h); V4<float>::load(p2 + h, v2 + h); V4<float>::load(p3 + h, v3 + h);
      V4<float>::load(q0 + h, w0 + h); V4<float>::load(q1 + h, w1 + h); V4<float>::load(q2 + h, w2 + h);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gav = g.ga ? v0[j] : 1.f;
      ca[j] = gav;
      if (HASY) { cb[j] = v1[j]; cc[j] = -(gav * v2[j]) - v1[j] * v3[j]; }
      const float sc = g.xs ? w0[j] : 1.f;
      as[j] = sc; ab[j] = (g.xb ? w2[j] : 0.f) - (g.xm ? w1[j] : 0.f) * sc;
    }
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;

  const int fi = C == 64 ? wave : (wave & 1);
  const int fj0 = C == 64 ? 0 : (wave >> 1);
  int troffG[2], troffA[9][NJ][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = fq * 8 + 4 * h + (fr >> 2);
    troffG[h] = img_off(row, (fi & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int F = (1 + t) * NF + fj0 + j;
        troffA[t][j][h] = (F >> 3) * PT * 256 + img_off(row, (F & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
      }
  }
  const int stG = img_off(r, cv);
  int stA[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { const int gc = (1 + t) * NV + cv; stA[t] = (gc >> 4) * PT * 256 + img_off(r, gc & 15); }

  f32x4 acc[9][NJ];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + gridDim.x - 1) / gridDim.x;
  const long s_begin = (long)blockIdx.x * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;

  uint4 re, ry, rx[9];
  uint32_t okb = 0;                                    // bits 0..8: tap inside the image; bit 9: the output pixel exists
#define SW_ISSUE(S)                                                                                    \
  {                                                                                                      \
    const long p = (S) * PT + r;                                                                         \
    const bool in = p < g.P;                                                                             \
    const long pcl = in ? p : g.P - 1;                                                                   \
    const int xo = (int)(pcl % g.Wo);                                                                    \
    const long byo = pcl / g.Wo;                                                                         \
    const int yo = (int)(byo % g.Ho);                                                                    \
    const long bi = (byo / g.Ho) * g.Hi;                                                                 \
    okb = in ? 512u : 0u;                                                                                \
    re = *reinterpret_cast<const uint4*>(g.e + pcl * g.lde + cv * 8);                                    \
    if (HASY) ry = *reinterpret_cast<const uint4*>(g.y + pcl * g.ldyr + cv * 8);                         \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                     \
      const int yi = 2 * yo + t / 3 - 1, xi = 2 * xo + t % 3 - 1;                                        \
      const bool ok = in && yi >= 0 && yi < g.Hi && xi >= 0 && xi < g.Wi;                                \
      okb |= ok ? (1u << t) : 0u;                                                                        \
      const long q = ok ? (bi + yi) * g.Wi + xi : (bi + 2 * yo) * g.Wi + 2 * xo;                         \
      rx[t] = *reinterpret_cast<const uint4*>(g.x + q * g.ldx + cv * 8);                                 \
    }                                                                                                    \
  }

  if (s_begin < s_end) SW_ISSUE(s_begin);
  int b = 0;
  for (long s = s_begin; s < s_end; ++s) {
    unsigned char* img = smem + b * BUF;
    {
      uint4 og = re;
      if (!gplain) {
        const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re);
        const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry);
        bf16x8 o;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          float lo = ca[2 * h] * blo(ue[h]), hi = ca[2 * h + 1] * bhi(ue[h]);
          if (HASY) { lo += cb[2 * h] * blo(uy[h]) + cc[2 * h]; hi += cb[2 * h + 1] * bhi(uy[h]) + cc[2 * h + 1]; }
          o[2 * h] = (T)lo; o[2 * h + 1] = (T)hi;
        }
        og = *reinterpret_cast<const uint4*>(&o);
      }
      if (!(okb & 512u)) og = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(img + stG) = og;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        uint4 oa = rx[t];
        if (!aplain) {
          const uint32_t* ux = reinterpret_cast<const uint32_t*>(&rx[t]);
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            o[2 * h] = (T)fmaxf(blo(ux[h]) * as[2 * h] + ab[2 * h], relu_lo);
            o[2 * h + 1] = (T)fmaxf(bhi(ux[h]) * as[2 * h + 1] + ab[2 * h + 1], relu_lo);
          }
          oa = *reinterpret_cast<const uint4*>(&o);
        }
        if (!((okb >> t) & 1u)) oa = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(img + stA[t]) = oa;
      }
    }
    if (s + 1 < s_end) SW_ISSUE(s + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const unsigned char* base = img + ks * 32 * 256;
      const bf16x8 gA = tr_pair(base + troffG[0], base + troffG[1]);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bf16x8 aB = tr_pair(base + troffA[t][j][0], base + troffA[t][j][1]);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gA, aB, acc[t][j], 0, 0, 0);
        }
    }
    b ^= 1;
  }
#undef SW_ISSUE
  float* row = g.ws + (long)blockIdx.x * (9 * C * C);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) row[((long)(16 * fi + 4 * fq + q) * C + 16 * (fj0 + j) + fr) * 9 + t] = acc[t][j][q];
}

// Rectangular form: NE channels on the OUTPUT grid ("e"), NX channels on the input grid ("x", padded to whole fragments in LDS).  Serves the
// weight gradient of nn.ConvTranspose2d(NE, NX, 3, stride 2) (UpsamplingBlock, TSS/models/esnet.py:71-80: e := the layer's input on the
// low-resolution grid, x := the gradient of its output), whose [NE][NX][3][3] layout this is.  64-pixel stages; wave w owns fragment
// w % NFE of the NE channels and the taps t = w / NFE (mod 4 / NFE).
template <int NE, int NX, bool HASY>
__global__ __launch_bounds__(NT, 2) void sc2_wgrad_rect_kernel(const SwArgs g) {
  constexpr int NFE = NE / 16, NFX = (NX + 15) / 16, NXP = NFX * 16, NVE = NE / 8, NVX = NXP / 8, PT = 64;
  constexpr int RPE = NT / NVE, NPE = (PT + RPE - 1) / RPE, RPX = NT / NVX, NPX = (PT + RPX - 1) / RPX;
  constexpr int NCH = NVE + 9 * NVX, NIMG = (NCH + 15) / 16, BUF = NIMG * PT * 256, NKS = PT / 32;
  constexpr int TSTEP = 4 / NFE, NTW = (9 + TSTEP - 1) / TSTEP;          // taps per wave
  static_assert(NFE == 1 || NFE == 2 || NFE == 4, "output fragments over the four waves");
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int cve = tid % NVE, re_ = tid / NVE, cvx = tid % NVX, rx_ = tid / NVX;

  float ca[8], cb[HASY ? 8 : 1], cc[HASY ? 8 : 1], as[8], ab[8];
  const bool gplain = !HASY && !g.ga, aplain = !g.xs && !g.xm && !g.xb && !g.x_relu;
  {
    const float* safe = reinterpret_cast<const float*>(g.e);
    const int cx = cvx * 8 < NX ? cvx * 8 : 0;
    float v0[8], v1[8], v2[8], v3[8], w0[8], w1[8], w2[8];
    const float* p0 = g.ga ? g.ga + cve * 8 : safe; const float* p1 = (HASY && g.gb) ? g.gb + cve * 8 : safe;
    const float* p2 = (HASY && g.gce) ? g.gce + cve * 8 : safe; const float* p3 = (HASY && g.gmu) ? g.gmu + cve * 8 : safe;
    const float* q0 = g.xs ? g.xs + cx : safe; const float* q1 = g.xm ? g.xm + cx : safe; const float* q2 = g.xb ? g.xb + cx : safe;
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(p0 + h, v0 + h); V4<float>::load(p1 + h, v1 + h); V4<float>::load(p2 + h, v2 + h); V4<float>::load(p3 + h, v3 + h);
      V4<float>::load(q0 + h, w0 + h); V4<float>::load(q1 + h, w1 + h); V4<float>::load(q2 + h, w2 + h);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gav = g.ga ? v0[j] : 1.f;
      ca[j] = gav;
      if (HASY) { cb[j] = v1[j]; cc[j] = -(gav * v2[j]) - v1[j] * v3[j]; }
      const float sc = g.xs ? w0[j] : 1.f;
      as[j] = sc; ab[j] = (g.xb ? w2[j] : 0.f) - (g.xm ? w1[j] : 0.f) * sc;
    }
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;

  const int fi = wave % NFE, t0 = wave / NFE;
  int troffG[2], troffA[NTW][NFX][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = fq * 8 + 4 * h + (fr >> 2);
    troffG[h] = img_off(row, fi * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
#pragma unroll
    for (int k = 0; k < NTW; ++k)
#pragma unroll
      for (int j = 0; j < NFX; ++j) {
        const int t = t0 + k * TSTEP;
        const int gc = NVE + (t < 9 ? t : 0) * NVX + j * 2 + ((fr & 3) >> 1);      // 16-byte column chunk of (tap, fragment j)
        troffA[k][j][h] = (gc >> 4) * PT * 256 + img_off(row, gc & 15) + 8 * (fr & 1);
      }
  }
  f32x4 acc[NTW][NFX];
#pragma unroll
  for (int k = 0; k < NTW; ++k)
#pragma unroll
    for (int j = 0; j < NFX; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + gridDim.x - 1) / gridDim.x;
  const long s_begin = (long)blockIdx.x * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;

  uint4 re[NPE], ry[HASY ? NPE : 1], rx[NPX][9];
  uint32_t oke = 0, okx[NPX];
#define SR_ISSUE(S)                                                                                      \
  {                                                                                                        \
    oke = 0;                                                                                               \
    _Pragma("unroll") for (int u = 0; u < NPE; ++u) {                                                     \
      const int rr = re_ + u * RPE;                                                                        \
      const long p = (S) * PT + rr;                                                                        \
      const bool in = rr < PT && p < g.P;                                                                  \
      const long pcl = in ? p : 0;                                                                         \
      oke |= in ? (1u << u) : 0u;                                                                          \
      re[u] = *reinterpret_cast<const uint4*>(g.e + pcl * g.lde + cve * 8);                                \
      if (HASY) ry[u] = *reinterpret_cast<const uint4*>(g.y + pcl * g.ldyr + cve * 8);                     \
    }                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < NPX; ++u) {                                                     \
      const int rr = rx_ + u * RPX;                                                                        \
      const long p = (S) * PT + rr;                                                                        \
      const bool in = rr < PT && p < g.P && cvx * 8 < NX;                                                  \
      const long pcl = (rr < PT && p < g.P) ? p : 0;                                                       \
      const int xo = (int)(pcl % g.Wo);                                                                    \
      const long byo = pcl / g.Wo;                                                                         \
      const int yo = (int)(byo % g.Ho);                                                                    \
      const long bi = (byo / g.Ho) * g.Hi;                                                                 \
      okx[u] = 0;                                                                                          \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                     \
        const int yi = 2 * yo + t / 3 - 1, xi = 2 * xo + t % 3 - 1;                                        \
        const bool ok = in && yi >= 0 && yi < g.Hi && xi >= 0 && xi < g.Wi;                                \
        okx[u] |= ok ? (1u << t) : 0u;                                                                     \
        const long q = ok ? (bi + yi) * g.Wi + xi : (bi + 2 * yo) * g.Wi + 2 * xo;                         \
        rx[u][t] = *reinterpret_cast<const uint4*>(g.x + q * g.ldx + (cvx * 8 < NX ? cvx * 8 : 0));       \
      }                                                                                                    \
    }                                                                                                      \
  }

  if (s_begin < s_end) SR_ISSUE(s_begin);
  int b = 0;
  for (long s = s_begin; s < s_end; ++s) {
    unsigned char* img = smem + b * BUF;
#pragma unroll
    for (int u = 0; u < NPE; ++u) {
      const int rr = re_ + u * RPE;
      if (rr < PT) {
        uint4 og = re[u];
        if (!gplain) {
          const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re[u]);
          const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry[HASY ? u : 0]);
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float lo = ca[2 * h] * blo(ue[h]), hi = ca[2 * h + 1] * bhi(ue[h]);
            if (HASY) { lo += cb[2 * h] * blo(uy[h]) + cc[2 * h]; hi += cb[2 * h + 1] * bhi(uy[h]) + cc[2 * h + 1]; }
            o[2 * h] = (T)lo; o[2 * h + 1] = (T)hi;
          }
          og = *reinterpret_cast<const uint4*>(&o);
        }
        if (!((oke >> u) & 1u)) og = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(img + img_off(rr, cve)) = og;
      }
    }
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int rr = rx_ + u * RPX;
      if (rr < PT) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          uint4 oa = rx[u][t];
          if (!aplain) {
            const uint32_t* ux = reinterpret_cast<const uint32_t*>(&rx[u][t]);
            bf16x8 o;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              o[2 * h] = (T)fmaxf(blo(ux[h]) * as[2 * h] + ab[2 * h], relu_lo);
              o[2 * h + 1] = (T)fmaxf(bhi(ux[h]) * as[2 * h + 1] + ab[2 * h + 1], relu_lo);
            }
            oa = *reinterpret_cast<const uint4*>(&o);
          }
          if (!((okx[u] >> t) & 1u)) oa = make_uint4(0u, 0u, 0u, 0u);
          const int gc = NVE + t * NVX + cvx;
          *reinterpret_cast<uint4*>(img + (gc >> 4) * PT * 256 + img_off(rr, gc & 15)) = oa;
        }
      }
    }
    if (s + 1 < s_end) SR_ISSUE(s + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const unsigned char* base = img + ks * 32 * 256;
      const bf16x8 gA = tr_pair(base + troffG[0], base + troffG[1]);
#pragma unroll
      for (int k = 0; k < NTW; ++k) {
        if (t0 + k * TSTEP < 9) {
#pragma unroll
          for (int j = 0; j < NFX; ++j) {
            const bf16x8 aB = tr_pair(base + troffA[k][j][0], base + troffA[k][j][1]);
            acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gA, aB, acc[k][j], 0, 0, 0);
          }
        }
      }
    }
    b ^= 1;
  }
#undef SR_ISSUE
  float* row = g.ws + (long)blockIdx.x * (9 * NE * NX);
#pragma unroll
  for (int k = 0; k < NTW; ++k) {
    const int t = t0 + k * TSTEP;
    if (t < 9) {
#pragma unroll
      for (int j = 0; j < NFX; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 16 * j + fr;
          if (c < NX) row[((long)(16 * fi + 4 * fq + q) * NX + c) * 9 + t] = acc[k][j][q];
        }
    }
  }
}

template <int NE, int NX> constexpr int swr_smem() { return 2 * (((NE / 8 + 9 * (((NX + 15) / 16) * 2) + 15) / 16) * 64 * 256); }

template <int NE, int NX, bool HASY>
void launch_swr(const SwArgs& g, int grid, hipStream_t stream) {
  constexpr int smem = swr_smem<NE, NX>();
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc2_wgrad_rect_kernel<NE, NX, HASY>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL((sc2_wgrad_rect_kernel<NE, NX, HASY>), dim3(grid), dim3(NT), smem, stream, g);
}

template <int C> constexpr int sw_smem() { return 2 * (((10 * (C / 8) + 15) / 16) * (C == 64 ? 32 : 64) * 256); }

int sw_rows(long P, int C) {
  const int PT = C == 64 ? 32 : 64;
  const long nstage = (P + PT - 1) / PT;
  long grid = C == 64 ? 256 : 512;                    // a row of partial sums is 36 C^2 bytes: 147 KB at 64 channels
  if (grid > (nstage + 7) / 8) grid = (nstage + 7) / 8;
  return (int)(grid < 1 ? 1 : grid);
}

bool sc_enabled() {
  static int v = -1;
  if (v < 0) { const char* s = getenv("TSS_SCONV"); v = (s && s[0] == '0') ? 0 : 1; }
  return v != 0;
}

template <class K>
int blocks_per_cu(K kernel, int smem) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NT, smem) != hipSuccess || nb < 1) nb = 1;
  return nb > 2 ? 2 : nb;
}

template <int CK, int CN, int MT>
void launch_fwd(const ScArgs& g, hipStream_t stream) {
  constexpr int smem = 3 * ((CN + 15) / 16) * (3 * ((CK + 31) / 32)) * 64 * 16;
  static tss::DevOnce attr;
  static int per_cu = 0;
  if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc2_fwd_kernel<CK, CN, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (per_cu == 0) per_cu = blocks_per_cu(sc2_fwd_kernel<CK, CN, MT>, smem);
  const long tpr = (g.Wo + 16 * MT - 1) / (16 * MT);
  const long nblk = (((long)g.B * g.Ho + 3) >> 2) * tpr;
  long grid = 256L * per_cu;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;
  if (grid > nblk) grid = nblk;
  hipLaunchKernelGGL((sc2_fwd_kernel<CK, CN, MT>), dim3((int)grid), dim3(NT), smem, stream, g);
}

template <int CK, int CN, int MODE, int MT>
void launch_bwd(const ScArgs& g, hipStream_t stream) {
  constexpr int smem = 9 * ((CN + 15) / 16) * ((CK + 31) / 32) * 64 * 16;
  static tss::DevOnce attr;
  static int per_cu = 0;
  if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc2_bwd_kernel<CK, CN, MODE, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (per_cu == 0) per_cu = blocks_per_cu(sc2_bwd_kernel<CK, CN, MODE, MT>, smem);
  const long tpr = (((g.Wi + 1) >> 1) + 16 * MT - 1) / (16 * MT);
  const long nblk = (((long)g.B * g.Hi + 3) >> 2) * 2 * tpr;
  long grid = 256L * per_cu;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;
  if (grid > nblk) grid = nblk;
  hipLaunchKernelGGL((sc2_bwd_kernel<CK, CN, MODE, MT>), dim3((int)grid), dim3(NT), smem, stream, g);
}

template <int C, bool HASY>
void launch_sw(const SwArgs& g, int grid, hipStream_t stream) {
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc2_wgrad_kernel<C, HASY>), hipFuncAttributeMaxDynamicSharedMemorySize, sw_smem<C>());
  hipLaunchKernelGGL((sc2_wgrad_kernel<C, HASY>), dim3(grid), dim3(NT), sw_smem<C>(), stream, g);
}

bool covered(int Hin, int Win, int Cin, int N) { return sc_enabled() && Cin == N && (N == 32 || N == 64) && Hin >= 2 && Win >= 2; }

}  // namespace

// forward on [9][N][Cin] f32 weights (tss_permute_w3x3); false: shape not covered
bool tss_sconv_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                   const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                   int B, int Hin, int Win, int Cin, int N, hipStream_t stream) {
  if (!sc_enabled() || Hin < 2 || Win < 2 || (ldx % 8) || (ldy % 4) || !tss::aligned16(x) || (reinterpret_cast<uintptr_t>(y) & 7u) || !w_tnc || B <= 0)
    return false;
  ScArgs g = {};
  g.B = B; g.Hi = Hin; g.Wi = Win; g.Ho = (Hin - 1) / 2 + 1; g.Wo = (Win - 1) / 2 + 1;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.w_os = Cin; g.w_ks = 1; g.w_t9 = (long)N * Cin;
  g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  if (N == 64 && Cin == 64) launch_fwd<64, 64, 1>(g, stream);
  else if (N == 32 && Cin == 32) launch_fwd<32, 32, 2>(g, stream);
  else if (N == 64 && Cin == 16) launch_fwd<16, 64, 2>(g, stream);      // input gradient of ConvTranspose2d(64, 16)
  else if (N == 16 && Cin == 24) launch_fwd<24, 16, 4>(g, stream);      // ... of ConvTranspose2d(16, 19 padded to 24)
  else if (N == 16 && Cin == 16) launch_fwd<16, 16, 4>(g, stream);
  else if (N == 48 && Cin == 16) launch_fwd<16, 48, 2>(g, stream);      // DownsamplingBlock(16, 64) of ESNet
  else if (N == 128 && Cin == 64) launch_fwd<64, 128, 1>(g, stream);    // input gradient of ConvTranspose2d(128, 64): 147 KB of weights, one block per CU
  else return false;
  return true;
}

// backward-data on [9][Cin][N] f32 weights (tss_permute_wtaps / tss_permute_w3x3)
bool tss_sconv_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                        const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                        const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                        void* e_in, long ldei, double* bstats, int B, int Hin, int Win, int Cin, int N, hipStream_t stream) {
  if (!(covered(Hin, Win, Cin, N) || (sc_enabled() && Hin >= 2 && Win >= 2 && Cin == 16 && N == 48)) || (lde % 8) || (ldei % 4) || !tss::aligned16(e) ||
      (reinterpret_cast<uintptr_t>(e_in) & 7u) || !w_tcn || B <= 0)
    return false;
  if (yraw) return false;      // a BatchNorm directly behind the layer (two operands per tap): not instantiated (the form with all four taps of a
                               // tile in registers spills; DownsamplingBlock's BatchNorm sits behind the concat, not behind the convolution)
  if (xraw && ((ldx % 4) || (reinterpret_cast<uintptr_t>(xraw) & 7u))) return false;
  ScArgs g = {};
  g.B = B; g.Hi = Hin; g.Wi = Win; g.Ho = (Hin - 1) / 2 + 1; g.Wo = (Win - 1) / 2 + 1;
  g.a0 = (const T*)e; g.lda0 = lde;
  g.c0 = ga;
  g.w = w_tcn; g.w_os = N; g.w_ks = 1; g.w_t9 = (long)Cin * N;
  g.y = (T*)e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = (const T*)xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  if (N == 48) launch_bwd<48, 16, 1, 2>(g, stream);                     // DownsamplingBlock(16, 64) of ESNet: 48 gradient channels -> 16
  else if (N == 64) launch_bwd<64, 64, 1, 2>(g, stream);
  else launch_bwd<32, 32, 1, 4>(g, stream);
  return true;
}

bool rect_covered(int Hin, int Win, int Cin, int N) {      // (N channels on the output grid, Cin on the input grid): ESNet's transposed layers
  return sc_enabled() && Hin >= 2 && Win >= 2 && ((N == 64 && Cin == 16) || (N == 16 && Cin == 24) || (N == 16 && Cin == 16));
}

extern "C" int tss_sconv_bwd_weight_rows(int B, int Hin, int Win, int Cin, int N, int dtype) {
  extern int g_tss_disable_fast;
  if (dtype != TSS_BF16 || g_tss_disable_fast || B <= 0) return 0;
  const long P = (long)B * ((Hin - 1) / 2 + 1) * ((Win - 1) / 2 + 1);
  if (covered(Hin, Win, Cin, N)) return sw_rows(P, N);
  if (rect_covered(Hin, Win, Cin, N)) {
    const long nstage = (P + 63) / 64;
    long grid = 512;
    if (grid > (nstage + 7) / 8) grid = (nstage + 7) / 8;
    return (int)(grid < 1 ? 1 : grid);
  }
  return 0;
}

extern "C" int tss_sconv_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                          const float* ga, const float* gb, const float* gce, const float* gmu,
                                          const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                                          int in_relu, float* ws, int B, int Hin, int Win, int Cin, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((covered(Hin, Win, Cin, N) || rect_covered(Hin, Win, Cin, N)) && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin && e &&
              xraw && ws && B > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(xraw) && (!yraw || tss::aligned16(yraw)), TSS_ERR_ALIGN);
  SwArgs g = {};
  g.B = B; g.Hi = Hin; g.Wi = Win; g.Ho = (Hin - 1) / 2 + 1; g.Wo = (Win - 1) / 2 + 1; g.P = (long)B * g.Ho * g.Wo;
  g.e = (const T*)e; g.lde = lde; g.y = (const T*)yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = (const T*)xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.ws = ws;
  const int grid = tss_sconv_bwd_weight_rows(B, Hin, Win, Cin, N, dtype);
  TSS_REQUIRE(grid > 0, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream, ((double)g.P * N * (yraw ? 2 : 1) + (double)B * Hin * Win * Cin) * 2.0,
                      2.0 * g.P * 9.0 * N * Cin);
  if (Cin != N) {
    if (N == 64 && Cin == 16) { if (yraw) launch_swr<64, 16, true>(g, grid, (hipStream_t)stream); else launch_swr<64, 16, false>(g, grid, (hipStream_t)stream); }
    else if (N == 16 && Cin == 24) { if (yraw) launch_swr<16, 24, true>(g, grid, (hipStream_t)stream); else launch_swr<16, 24, false>(g, grid, (hipStream_t)stream); }
    return tss::check_last("sconv_wgrad_rect");
  }
  if (N == 16) { if (yraw) launch_swr<16, 16, true>(g, grid, (hipStream_t)stream); else launch_swr<16, 16, false>(g, grid, (hipStream_t)stream); return tss::check_last("sconv_wgrad_rect"); }
  if (N == 64) { if (yraw) launch_sw<64, true>(g, grid, (hipStream_t)stream); else launch_sw<64, false>(g, grid, (hipStream_t)stream); }
  else { if (yraw) launch_sw<32, true>(g, grid, (hipStream_t)stream); else launch_sw<32, false>(g, grid, (hipStream_t)stream); }
  return tss::check_last("sconv_wgrad");
}

// forward of nn.ConvTranspose2d(Cin_t, Cout, 3, stride=2, padding=1, output_padding=1) + bias on [9][Cout][Cin_t] f32 weights (tss_permute_wtaps of
// the layer's [Cin_t][Cout][3][3] tensor): the parity-class gather above with x as the source grid.  false: shape not covered.
bool tss_sconv_transposed_fwd(const void* x, long ldx, const float* w_tcn, const float* bias, void* y, long ldy,
                              int B, int Hout, int Wout, int Cout, int Cin_t, hipStream_t stream) {
  if (!sc_enabled() || (Hout & 1) || (Wout & 1) || Hout < 2 || Wout < 2 || (ldx % 8) || (ldy % 4) || !tss::aligned16(x) ||
      (reinterpret_cast<uintptr_t>(y) & 7u) || !w_tcn || B <= 0)
    return false;
  ScArgs g = {};
  g.B = B; g.Hi = Hout; g.Wi = Wout; g.Ho = Hout / 2; g.Wo = Wout / 2;
  g.a0 = (const T*)x; g.lda0 = ldx;
  g.w = w_tcn; g.w_os = Cin_t; g.w_ks = 1; g.w_t9 = (long)Cout * Cin_t;
  g.bias = bias; g.y = (T*)y; g.ldy = ldy;
  if (Cin_t == 64 && Cout == 16) launch_bwd<64, 16, 1, 2>(g, stream);
  else if (Cin_t == 16 && Cout == 24) launch_bwd<16, 24, 1, 4>(g, stream);
  else if (Cin_t == 16 && Cout == 16) launch_bwd<16, 16, 1, 4>(g, stream);
  else if (Cin_t == 128 && Cout == 64) launch_bwd<128, 64, 1, 1>(g, stream);      // 147 KB of weights: one block per CU
  else return false;
  return true;
}

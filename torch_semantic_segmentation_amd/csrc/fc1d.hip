// Lean bf16 kernels of the factorized (1-D, three taps, dilated) convolutions: nn.Conv2d(C, C, (1,3) / (3,1), padding = dilation) of
// FactorizedConvBlock, TSS/models/lednet.py:157-180 (SS-nbt unit :95-124) and of ESNet's FCU / PFCU units, TSS/models/esnet.py:83-166.
// Forward and backward-data (the same three-tap convolution with the transposed weights and mirrored taps).
//
// LEDNet at 8 x 3 x 1024 x 2048 runs 104 of these per pass on 4.2 M / 1 M / 262 k pixel maps with C = 16 / 32 / 64 channels: 24 to 96
// flop per byte moved, i.e. HBM-bound by a factor of 3 - 13 on MI355X.  The generic implicit-GEMM kernel (convgemm.hip, A_TAPS) walks
// the taps as dependent chunks through LDS with three barriers each and pads every tile to 128 output channels: 370 - 400 GB/s
// (profiles/r03_bench_lednet.json).  Here there is no activation staging at all:
//   * the B operand of v_mfma_f32_16x16x32_bf16 is "pixel fr, 8 consecutive contraction channels fq*8.." per lane -- exactly one
//     16-byte vector of an NHWC row.  A lane loads its operand vectors straight from global memory (one tap = one shifted pixel),
//     applies the pending BatchNorm / ReLU (or the BatchNorm-backward combination of e and y) in registers, and feeds the MFMA;
//     the three taps are the contraction axis (K = 3 C; for C = 16 two taps share one 32-wide k-step);
//   * a WAVE owns a tile of 16 * MT consecutive pixels of one image row and all C output channels: no barrier in the loop, the
//     next tile's loads are issued as soon as the current tile's operands are converted and land under its MFMAs and epilogue;
//   * the weights sit in LDS in fragment order (<= 24 KB, conflict-free 16-byte reads), written once per block;
//   * the four waves of a block take four vertically adjacent rows of the same column range, so the rows a vertical (3x1) layer
//     reads three times are shared through the CU's L1 / the XCD's L2;
//   * epilogue as in conv3x3.hip: bias, ReLU mask of the layer's input (backward), rounding, statistics from the stored bits,
//     one slab row per block.
// C in {16, 32, 64}, square layers; everything else stays on convgemm_kernel.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256;

struct FcArgs {
  int B, H, W, D, axis, tap_sign;                     // axis 0: taps along W (1x3), 1: along H (3x1); source = p + tap_sign * (tap - 1) * D
  const T* a0; long lda0; const T* a1; long lda1;     // fwd: x (a1 unused)   bwd: e, yraw
  const float* c0; const float* c1; const float* c2; const float* c3; int a_relu;
  const float* w; int w_os, w_ks, w_ts;               // f32 weights: element (tap, output o, contraction k) at w[o * w_os + k * w_ks + tap * w_ts]
  const float* bias;
  T* y; long ldy; double* stats;
  const T* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// MODE 0: forward (a = relu?((x - c1) * c0 + c2));  1: backward, g = c0 * e;  2: backward, g = c0 * (e - c2) + c1 * (y - c3)
template <int C, int MODE, int MT>
__global__ __launch_bounds__(NT, 2) void fc1d_kernel(const FcArgs g) {
  constexpr int NF = C / 16, KT = 3 * C, NKS = (KT + 31) / 32, NCS = C == 64 ? 2 : 1, TWV = 16 * MT;
  constexpr bool BWD = MODE != 0;
  __shared__ uint4 Wl[NF * NKS * 64];                 // fragment (i, ks), lane l: W[i*16 + (l&15)][ks*32 + (l>>4)*8 .. +8] as bf16
  __shared__ __align__(16) float Ec[4][C];            // bias, and the mask constants of the backward epilogue: mean, scale, shift
  __shared__ float red[4][2][C];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  for (int e = tid; e < NF * NKS * 64; e += NT) {
    const int f = e >> 6, l = e & 63;
    const int i = f / NKS, ks = f - i * NKS;
    const int n = i * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8;
    const int tap = k / C, c = k - tap * C;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)0.f;
    if (tap < 3) {
      const float* src = g.w + (long)n * g.w_os + (long)c * g.w_ks + (long)tap * g.w_ts;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)src[(long)j * g.w_ks];
    }
    Wl[e] = *reinterpret_cast<const uint4*>(&o);
  }
  if (tid < C) {
    float eb = 0.f, em = 0.f, es = 1.f, eh = 0.f;
    if (g.bias) eb = g.bias[tid];
    if (BWD && g.xm) {
      if (g.mm) em = g.mm[tid];
      if (g.ms) es = g.ms[tid];
      if (g.mb) eh = g.mb[tid];
    }
    Ec[0][tid] = eb; Ec[1][tid] = em; Ec[2][tid] = es; Ec[3][tid] = eh;
  }

  // ---- this lane's operand channels and their folded prologue constants (fixed for the whole kernel)
  int cch[NCS];
#pragma unroll
  for (int s = 0; s < NCS; ++s) cch[s] = C == 64 ? s * 32 + fq * 8 : (C == 32 ? fq * 8 : (fq & 1) * 8);
  float k0[NCS][8], k1[MODE == 2 ? NCS : 1][8], kadd[NCS][8];
  bool plain;
  if (MODE == 0) plain = !g.c0 && !g.c1 && !g.c2 && !g.a_relu;
  else if (MODE == 1) plain = !g.c0;
  else plain = false;
  {
    const float* safe = g.w;
#pragma unroll
    for (int s = 0; s < NCS; ++s) {
      float v0[8], v1[8], v2[8], v3[8];
      const float* p0 = g.c0 ? g.c0 + cch[s] : safe;
      const float* p1 = g.c1 ? g.c1 + cch[s] : safe;
      const float* p2 = g.c2 ? g.c2 + cch[s] : safe;
      const float* p3 = (MODE == 2 && g.c3) ? g.c3 + cch[s] : safe;
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        V4<float>::load(p0 + h, v0 + h); V4<float>::load(p1 + h, v1 + h); V4<float>::load(p2 + h, v2 + h); V4<float>::load(p3 + h, v3 + h);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c0v = g.c0 ? v0[j] : 1.f, c1v = g.c1 ? v1[j] : 0.f, c2v = g.c2 ? v2[j] : 0.f, c3v = (MODE == 2 && g.c3) ? v3[j] : 0.f;
        k0[s][j] = c0v;
        if (MODE == 0) kadd[s][j] = c2v - c1v * c0v;
        else if (MODE == 1) kadd[s][j] = 0.f;
        else { k1[s][j] = c1v; kadd[s][j] = -(c0v * c2v) - c1v * c3v; }
      }
    }
  }
  const float relu_lo = (MODE == 0 && g.a_relu) ? 0.f : -TSS_INF;

  float st1[NF][4], st2[NF][4];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  // ---- tiles: groups of four image rows, column range by column range; wave w of a block takes row 4 * group + w
  const int tpr = (g.W + TWV - 1) / TWV;
  const long rows = (long)g.B * g.H;
  const long ngroups = (rows + 3) >> 2;
  const long nblk = ngroups * tpr;                      // block-tiles: (row group, column range)
  const int lim = g.axis ? g.H : g.W;
  const long pstep = g.axis ? g.W : 1;

  uint4 ra[MT][NKS], rb[MODE == 2 ? MT : 1][MODE == 2 ? NKS : 1];
  uint2 rxm[BWD ? MT : 1][BWD ? NF : 1];
  uint32_t okb = 0;                                     // validity bits of the loads in flight: bit m * NKS + ks
  long pc = 0; int x0 = 0; bool live = false;           // geometry of the tile whose loads are in flight

#define FC_GEOM(BT, PC, X0, YY, LIVE)                                                   \
  {                                                                                       \
    const long grp = (BT) / tpr;                                                          \
    const int tx = (int)((BT) - grp * tpr);                                               \
    const long by = grp * 4 + wave;                                                       \
    LIVE = by < rows;                                                                     \
    const long byc = LIVE ? by : rows - 1;                                                \
    YY = (int)(byc % g.H);                                                                \
    X0 = tx * TWV;                                                                        \
    PC = byc * g.W + X0;                                                                  \
  }
#define FC_ISSUE(PC, X0, YY)                                                                                       \
  {                                                                                                                  \
    okb = 0;                                                                                                         \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                                \
      const int px = m * 16 + fr;                                                                                    \
      const bool inw = (X0) + px < g.W;                                                                              \
      _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                          \
        const int tap = C >= 32 ? (ks * 32) / C : ks * 2 + (fq >> 1);                                               \
        const int sh = g.tap_sign * (tap - 1) * g.D;                                                                 \
        const int t = (g.axis ? (YY) : (X0) + px) + sh;                                                              \
        const bool ok = inw && tap < 3 && t >= 0 && t < lim;                                                         \
        const long q = ok ? (PC) + px + sh * pstep : (PC);                                                           \
        okb |= ok ? (1u << (m * NKS + ks)) : 0u;                                                                     \
        ra[m][ks] = *reinterpret_cast<const uint4*>(g.a0 + q * g.lda0 + cch[C == 64 ? (ks & 1) : 0]);              \
        if (MODE == 2) rb[m][ks] = *reinterpret_cast<const uint4*>(g.a1 + q * g.lda1 + cch[C == 64 ? (ks & 1) : 0]); \
      }                                                                                                              \
    }                                                                                                                \
  }
#define FC_ISSUE_XM(PC, X0)                                                                                         \
  if (BWD && g.xm) {                                                                                                 \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                                \
      const int px = m * 16 + fr;                                                                                    \
      const long p = (PC) + (((X0) + px < g.W) ? px : 0);                                                            \
      _Pragma("unroll") for (int i = 0; i < NF; ++i)                                                                \
        rxm[m][i] = *reinterpret_cast<const uint2*>(g.xm + p * g.ldxm + i * 16 + fq * 4);                            \
    }                                                                                                                \
  }

  long bt = blockIdx.x;
  int yy = 0;
  if (bt < nblk) {
    FC_GEOM(bt, pc, x0, yy, live);
    FC_ISSUE(pc, x0, yy);
    FC_ISSUE_XM(pc, x0);
  }
  __syncthreads();                                      // weights and epilogue constants are in LDS

  for (; bt < nblk; bt += gridDim.x) {
    // ---- operands of this tile: registers -> MFMA layout (the loads were issued one iteration ago)
    bf16x8 op[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bool ok = (okb >> (m * NKS + ks)) & 1u;
        uint4 r = ra[m][ks];
        if (!plain) {
          constexpr int sidx = 0;
          const int s = C == 64 ? (ks & 1) : sidx;
          const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra[m][ks]);
          const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb[MODE == 2 ? m : 0][MODE == 2 ? ks : 0]);
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float lo = blo(ua[h]) * k0[s][2 * h] + kadd[s][2 * h];
            float hi = bhi(ua[h]) * k0[s][2 * h + 1] + kadd[s][2 * h + 1];
            if (MODE == 2) { lo += blo(ub[h]) * k1[s][2 * h]; hi += bhi(ub[h]) * k1[s][2 * h + 1]; }
            if (MODE == 0) { lo = fmaxf(lo, relu_lo); hi = fmaxf(hi, relu_lo); }
            o[2 * h] = (T)lo; o[2 * h + 1] = (T)hi;
          }
          r = *reinterpret_cast<const uint4*>(&o);
        }
        if (!ok) r = make_uint4(0u, 0u, 0u, 0u);        // zero padding applies to the ACTIVATED tensor
        op[m][ks] = *reinterpret_cast<const bf16x8*>(&r);
      }
    const long pc_cur = pc; const int x0_cur = x0; const bool live_cur = live;

    // ---- the next tile's loads go out now and land under this tile's MFMAs and epilogue
    const long btn = bt + gridDim.x;
    if (btn < nblk) {
      FC_GEOM(btn, pc, x0, yy, live);
      FC_ISSUE(pc, x0, yy);
    }

    f32x4 acc[MT][NF];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");                    // the weight fragments are re-read from LDS per tile, not hoisted into 96 registers
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const uint4 wr = Wl[(i * NKS + ks) * 64 + lane];
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wr);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, op[m][ks], acc[m][i], 0, 0, 0);
      }

    // ---- epilogue: lane owns pixel m*16 + fr x channels i*16 + fq*4 .. +3
    if (live_cur) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int nl = i * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(&Ec[0][nl]);
        const float cbias[4] = {e0.x, e0.y, e0.z, e0.w};
        float cmm[4] = {0.f, 0.f, 0.f, 0.f}, cms[4] = {1.f, 1.f, 1.f, 1.f}, cmb[4] = {0.f, 0.f, 0.f, 0.f};
        if (BWD) {
          const float4 e1 = *reinterpret_cast<const float4*>(&Ec[1][nl]);
          const float4 e2 = *reinterpret_cast<const float4*>(&Ec[2][nl]);
          const float4 e3 = *reinterpret_cast<const float4*>(&Ec[3][nl]);
          cmm[0] = e1.x; cmm[1] = e1.y; cmm[2] = e1.z; cmm[3] = e1.w;
          cms[0] = e2.x; cms[1] = e2.y; cms[2] = e2.z; cms[3] = e2.w;
          cmb[0] = e3.x; cmb[1] = e3.y; cmb[2] = e3.z; cmb[3] = e3.w;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int px = m * 16 + fr;
          if (x0_cur + px < g.W) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q] + cbias[q];
            bf16x4 o;
            if (BWD && g.xm) {
              const uint2 xr = rxm[BWD ? m : 0][BWD ? i : 0];
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
            *reinterpret_cast<bf16x4*>(g.y + (pc_cur + px) * g.ldy + nl) = o;
          }
        }
      }
    }
    if (btn < nblk) { FC_ISSUE_XM(pc, x0); }
  }
#undef FC_GEOM
#undef FC_ISSUE
#undef FC_ISSUE_XM

  // ---- statistics: one slab row per block; rows no block owns are zeroed here (the caller never clears the buffer)
  if (g.stats) {
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
        if (fr == 0) { red[wave][0][i * 16 + fq * 4 + q] = u; red[wave][1][i * 16 + fq * 4 + q] = w2; }
      }
    __syncthreads();
    if (tid < C) {
      const double a = ((double)red[0][0][tid] + (double)red[1][0][tid]) + ((double)red[2][0][tid] + (double)red[3][0][tid]);
      const double b = ((double)red[0][1][tid] + (double)red[1][1][tid]) + ((double)red[2][1][tid] + (double)red[3][1][tid]);
      const int row = blockIdx.x, rows_used = gridDim.x;
      g.stats[(long)row * 2 * C + tid] = a;
      g.stats[(long)row * 2 * C + C + tid] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * C + tid] = 0.0;
        g.stats[(long)rr * 2 * C + C + tid] = 0.0;
      }
    }
  }
}

template <int C, int MODE, int MT>
void launch_fc(const FcArgs& g, hipStream_t stream) {
  constexpr int TWV = 16 * MT;
  const long tpr = (g.W + TWV - 1) / TWV;
  const long nblk = (((long)g.B * g.H + 3) >> 2) * tpr;
  static int per_cu = 0;
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fc1d_kernel<C, MODE, MT>, NT, 0) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb > 2 ? 2 : nb;
  }
  long grid = 256L * per_cu;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;
  if (grid > nblk) grid = nblk;
  hipLaunchKernelGGL((fc1d_kernel<C, MODE, MT>), dim3((int)grid), dim3(NT), 0, stream, g);
}

template <int MODE>
bool dispatch_fc(const FcArgs& g, int C, hipStream_t stream) {
  switch (C) {
    case 16: launch_fc<16, MODE, 4>(g, stream); return true;
    case 32: launch_fc<32, MODE, MODE == 2 ? 2 : 4>(g, stream); return true;
    case 64: launch_fc<64, MODE, 1>(g, stream); return true;
    default: return false;
  }
}

// unfold for the weight gradient: col[p][c*3 + tap] = act(x[p + off(tap)][c]) (0 outside the image), so that
// dW[n][c][tap] = sum_p g[p][n] * col[p][c*3 + tap] is a pointwise weight gradient with K = 3 C whose output layout is torch's
// [N][C][1][3] / [N][C][3][1].  One lane = one pixel x 8 channels: three 16-byte loads, three 16-byte stores.
__global__ __launch_bounds__(256) void im2col1d3_kernel(const T* x, long ldx, const float* mean, const float* scale, const float* bias,
                                                        int relu, T* col, int B, int H, int W, int C, int axis, int dil) {
  const int nv = C >> 3;
  const long total = (long)B * H * W * nv;
  const float lo = relu ? 0.f : -TSS_INF;
  const long pstep = axis ? W : 1;
  const int lim = axis ? H : W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % nv);
    const long pix = i / nv;
    const int xx = (int)(pix % W);
    const int yy = (int)((pix / W) % H);
    uint4 r[3];
    bool ok[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int sh = (t - 1) * dil;
      const int u = (axis ? yy : xx) + sh;
      ok[t] = u >= 0 && u < lim;
      const long q = ok[t] ? pix + sh * pstep : pix;
      r[t] = *reinterpret_cast<const uint4*>(x + q * ldx + cv * 8);
    }
    float sc[8], ad[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = scale ? scale[cv * 8 + j] : 1.f;
      sc[j] = s;
      ad[j] = (bias ? bias[cv * 8 + j] : 0.f) - (mean ? mean[cv * 8 + j] : 0.f) * s;
    }
    bf16x8 o[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const uint32_t* u = reinterpret_cast<const uint32_t*>(&r[t]);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float a = ok[t] ? fmaxf(blo(u[h]) * sc[2 * h] + ad[2 * h], lo) : 0.f;
        const float b = ok[t] ? fmaxf(bhi(u[h]) * sc[2 * h + 1] + ad[2 * h + 1], lo) : 0.f;
        const int ja = (2 * h) * 3 + t, jb = (2 * h + 1) * 3 + t;      // column c*3 + tap within this lane's 24
        o[ja >> 3][ja & 7] = (T)a;
        o[jb >> 3][jb & 7] = (T)b;
      }
    }
    T* dst = col + pix * ((long)C * 3) + (long)cv * 24;
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<bf16x8*>(dst + t * 8) = o[t];
  }
}

// ---- weight gradient of a three-tap layer in one sweep: dW[n][c][tap] = sum_p g[p][n] * a[p + off(tap)][c], g the BatchNorm-backward
// combination of e and y (or e alone), a the activated input.  The contraction runs over PIXELS, so both MFMA operands are needed
// pixel-major: a stage of PT pixels of g and of the three shifted copies of a is normalised in registers, written once to LDS as
// [pixel][channel] rows of the dual-use image of pwsweep.hip (16-byte row writes, ds_read_b64_tr_b16 transposed reads: no bank
// conflicts either way) and read back as k-major fragments.  One barrier per stage (two buffers); the next stage's loads are issued
// when the current stage's registers are free and land under its MFMAs.  A block owns a contiguous range of stages and leaves one
// row of partial sums in the workspace (torch's [N][C][taps] order); the rows are added by tss_dw_reduce_many.
// (unfold + pointwise kernel, round 4 first version: 9 C bytes per pixel instead of 3 C -- 428 us on the 1/2-resolution layers.)
typedef __attribute__((ext_vector_type(4))) short v4s;
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
  union { v4s h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lo);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)hi);
  return u.v;
}

struct FwArgs {
  long P; int B, H, W, D, axis;
  const T* e; long lde; const T* y; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  const T* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu;
  float* ws;                                           // [gridDim.x][C * C * 3]
};

template <int C, bool HASY>
__global__ __launch_bounds__(NT, 2) void fc1d_wgrad_kernel(const FwArgs g) {
  constexpr int NF = C / 16, NV = C / 8, PT = C == 64 ? 64 : 128, RPP = NT / NV, NP = PT / RPP, NIMG = (4 * NV + 15) / 16;
  constexpr int BUF = NIMG * PT * 256, NKS = PT / 32, NJ = C == 64 ? 4 : 1;
  extern __shared__ __align__(16) unsigned char smem[];          // two stage buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int cv = tid % NV, r = tid / NV;

  // ---- folded constants of this thread's channel vector
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
      if (HASY) { cb[j] = v1[j]; cc[j] = -(gav * v2[j]) - v1[j] * v3[j]; }      // g = ga*e + gb*y + cc
      const float sc = g.xs ? w0[j] : 1.f;
      as[j] = sc; ab[j] = (g.xb ? w2[j] : 0.f) - (g.xm ? w1[j] : 0.f) * sc;      // a = relu?(x*as + ab)
    }
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;

  // ---- this wave's part of the output and its transposed-read offsets
  const int fi = C == 64 ? wave : (C == 32 ? (wave & 1) : 0);          // fragment of output channels n
  const int fj0 = C == 32 ? (wave >> 1) : 0;                           // first fragment of input channels c
  int troffG[2], troffA[3][NJ][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = fq * 8 + 4 * h + (fr >> 2);
    troffG[h] = img_off(row, (fi & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int F = (1 + t) * NF + fj0 + j;
        troffA[t][j][h] = (F >> 3) * PT * 256 + img_off(row, (F & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
      }
  }
  // staging offsets of this thread's chunk: g at column chunk cv, tap t of a at (1 + t) * NV + cv
  int stG[NP], stA[NP][3];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int row = r + u * RPP;
    stG[u] = img_off(row, cv);
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int gc = (1 + t) * NV + cv; stA[u][t] = (gc >> 4) * PT * 256 + img_off(row, gc & 15); }
  }

  f32x4 acc[3][NJ];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + gridDim.x - 1) / gridDim.x;
  const long s_begin = (long)blockIdx.x * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;
  const int lim = g.axis ? g.H : g.W;
  const long pstep = g.axis ? g.W : 1;

  uint4 re[NP], ry[HASY ? NP : 1], rx[NP][3];
  uint32_t okb = 0;                                    // bits u*4 + t: tap t of pass u is inside the image; u*4 + 3: the pixel exists
#define FW_ISSUE(S)                                                                                      \
  {                                                                                                        \
    okb = 0;                                                                                               \
    _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                      \
      const long p = (S) * PT + r + u * RPP;                                                               \
      const bool in = p < g.P;                                                                             \
      const long pcl = in ? p : g.P - 1;                                                                   \
      const int xx = (int)(pcl % g.W);                                                                     \
      const int yy = (int)((pcl / g.W) % g.H);                                                             \
      okb |= in ? (8u << (u * 4)) : 0u;                                                                    \
      re[u] = *reinterpret_cast<const uint4*>(g.e + pcl * g.lde + cv * 8);                                 \
      if (HASY) ry[u] = *reinterpret_cast<const uint4*>(g.y + pcl * g.ldyr + cv * 8);                      \
      _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                     \
        const int sh = (t - 1) * g.D;                                                                      \
        const int c1 = (g.axis ? yy : xx) + sh;                                                            \
        const bool ok = in && c1 >= 0 && c1 < lim;                                                         \
        okb |= ok ? (1u << (u * 4 + t)) : 0u;                                                              \
        rx[u][t] = *reinterpret_cast<const uint4*>(g.x + (ok ? pcl + sh * pstep : pcl) * g.ldx + cv * 8);  \
      }                                                                                                    \
    }                                                                                                      \
  }

  if (s_begin < s_end) FW_ISSUE(s_begin);
  int b = 0;
  for (long s = s_begin; s < s_end; ++s) {
    unsigned char* img = smem + b * BUF;
    // ---- registers -> normalised bf16 rows in LDS
#pragma unroll
    for (int u = 0; u < NP; ++u) {
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
      if (!((okb >> (u * 4 + 3)) & 1u)) og = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(img + stG[u]) = og;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
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
        if (!((okb >> (u * 4 + t)) & 1u)) oa = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(img + stA[u][t]) = oa;
      }
    }
    if (s + 1 < s_end) FW_ISSUE(s + 1);
    __syncthreads();
    // ---- D[n][c] += sum over the stage's pixels, tap by tap
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (C == 16 && ks != wave) continue;             // 16 channels: one fragment per tap, the four waves split the pixels
      const unsigned char* base = img + ks * 32 * 256;
      const bf16x8 gA = tr_pair(base + troffG[0], base + troffG[1]);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bf16x8 aB = tr_pair(base + troffA[t][j][0], base + troffA[t][j][1]);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gA, aB, acc[t][j], 0, 0, 0);
        }
    }
    b ^= 1;
  }
#undef FW_ISSUE

  // ---- this block's row of partial sums: lane holds n = 16 fi + 4 fq + q, c = 16 (fj0 + j) + fr, tap t
  float* row = g.ws + (long)blockIdx.x * (3 * C * C);
  if (C == 16) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);       // [3 * C * C]
    for (int i = tid; i < 3 * C * C; i += NT) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(&red[((fq * 4 + q) * C + fr) * 3 + t], acc[t][0][q]);
    __syncthreads();
    for (int i = tid; i < 3 * C * C; i += NT) row[i] = red[i];
  } else {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) row[((long)(16 * fi + 4 * fq + q) * C + 16 * (fj0 + j) + fr) * 3 + t] = acc[t][j][q];
  }
}

template <int C> constexpr int wg_smem() { return 2 * (((4 * (C / 8) + 15) / 16) * (C == 64 ? 64 : 128) * 256); }

template <int C, bool HASY>
void launch_fw(const FwArgs& g, int grid, hipStream_t stream) {
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc1d_wgrad_kernel<C, HASY>), hipFuncAttributeMaxDynamicSharedMemorySize, wg_smem<C>());
  hipLaunchKernelGGL((fc1d_wgrad_kernel<C, HASY>), dim3(grid), dim3(NT), wg_smem<C>(), stream, g);
}

int wg_rows(long P, int C) {
  const int PT = C == 64 ? 64 : 128;
  const long nstage = (P + PT - 1) / PT;
  long grid = nstage < 512 ? nstage : 512;
  // every block should sweep at least 4 stages: its row of partial sums costs 12 C^2 bytes twice
  if (grid > (nstage + 3) / 4) grid = (nstage + 3) / 4;
  return (int)(grid < 1 ? 1 : grid);
}

bool fc_enabled() {
  static int v = -1;
  if (v < 0) { const char* s = getenv("TSS_FC1D"); v = (s && s[0] == '0') ? 0 : 1; }
  return v != 0;
}

}  // namespace

// forward / backward-data of the three-tap layers on the lean kernel; false: shape not covered, the caller takes the generic kernel
bool tss_fc1d_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                  const float* w_tnc, int torch_layout, const float* bias, void* y, long ldy, double* stats,
                  int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream) {
  if (!fc_enabled() || Cin != N || (N != 16 && N != 32 && N != 64) || (ldx % 8) || (ldy % 4) || !tss::aligned16(x) || !w_tnc ||
      (reinterpret_cast<uintptr_t>(y) & 7u) || (long)B * H * W <= 0)
    return false;
  FcArgs g = {};
  g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis; g.tap_sign = 1;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  if (torch_layout) { g.w_os = 3 * Cin; g.w_ks = 3; g.w_ts = 1; }      // [N][Cin][taps]
  else { g.w_os = Cin; g.w_ks = 1; g.w_ts = N * Cin; }                  // [taps][N][Cin]
  return dispatch_fc<0>(g, N, stream);
}

bool tss_fc1d_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                       const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn, int torch_layout,
                       const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                       void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream) {
  if (!fc_enabled() || Cin != N || (N != 16 && N != 32 && N != 64) || (lde % 8) || (ldei % 4) || !tss::aligned16(e) || !w_tcn ||
      (reinterpret_cast<uintptr_t>(e_in) & 7u) || (long)B * H * W <= 0)
    return false;
  if (yraw && ((ldyr % 8) || !tss::aligned16(yraw) || !ga || !gb || !gce || !gmu)) return false;
  if (xraw && ((ldx % 4) || (reinterpret_cast<uintptr_t>(xraw) & 7u))) return false;
  FcArgs g = {};
  g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis; g.tap_sign = -1;
  g.a0 = (const T*)e; g.lda0 = lde; g.a1 = (const T*)yraw; g.lda1 = ldyr;
  g.c0 = ga;
  if (yraw) { g.c1 = gb; g.c2 = gce; g.c3 = gmu; }
  g.w = w_tcn; g.y = (T*)e_in; g.ldy = ldei; g.stats = bstats;
  if (torch_layout) { g.w_os = 3; g.w_ks = 3 * Cin; g.w_ts = 1; }      // [N][Cin][taps]: output = input channel, contraction = n
  else { g.w_os = N; g.w_ks = 1; g.w_ts = Cin * N; }                    // [taps][Cin][N]
  g.xm = (const T*)xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  return yraw ? dispatch_fc<2>(g, N, stream) : dispatch_fc<1>(g, N, stream);
}

extern "C" int tss_im2col1d3(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                             void* col, int B, int H, int W, int C, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldx % 8) == 0 && ldx >= C && dil >= 1 && (axis == 0 || axis == 1) && x && col, TSS_ERR_SHAPE);
  TSS_REQUIRE((in_mean != nullptr) == (in_scale != nullptr) && (in_bias != nullptr) == (in_scale != nullptr), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(col), TSS_ERR_ALIGN);
  const long total = (long)B * H * W * (C / 8);
  if (total == 0) return TSS_OK;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(im2col1d3_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, in_mean, in_scale,
                     in_bias, in_relu, (T*)col, B, H, W, C, axis, dil);
  return tss::check_last("im2col1d3");
}

/* weight gradient of a three-tap layer in one sweep; rows of partial sums (tss_conv1d3_bwd_weight_rows of them, 3 * N * Cin floats each,
 * torch's [N][Cin][taps] order) go to ws and are added to the gradient by tss_dw_reduce_many.  rows == 0: shape not covered. */
extern "C" int tss_conv1d3_bwd_weight_rows(long P, int Cin, int N, int dtype) {
  extern int g_tss_disable_fast;
  if (dtype != TSS_BF16 || g_tss_disable_fast || !fc_enabled() || Cin != N || (N != 16 && N != 32 && N != 64) || P <= 0) return 0;
  static int v = -1;
  if (v < 0) { const char* s = getenv("TSS_FC1D_WGRAD_SWEEP"); v = (s && s[0] == '0') ? 0 : 1; }
  return v ? wg_rows(P, N) : 0;
}

extern "C" int tss_conv1d3_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                            const float* ga, const float* gb, const float* gce, const float* gmu,
                                            const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                                            int in_relu, float* ws, int B, int H, int W, int Cin, int N, int axis, int dil, int dtype,
                                            void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin == N && (N == 16 || N == 32 || N == 64) && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin && dil >= 1 &&
              (axis == 0 || axis == 1) && e && xraw && ws && (long)B * H * W > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(xraw) && (!yraw || tss::aligned16(yraw)), TSS_ERR_ALIGN);
  FwArgs g = {};
  g.P = (long)B * H * W; g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis;
  g.e = (const T*)e; g.lde = lde; g.y = (const T*)yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = (const T*)xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.ws = ws;
  const int grid = wg_rows(g.P, N);
  tss::ProfScope prof(TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream, (double)g.P * N * (yraw ? 3 : 2) * 2.0, 2.0 * g.P * 3.0 * N * N);
  if (N == 64) { if (yraw) launch_fw<64, true>(g, grid, (hipStream_t)stream); else launch_fw<64, false>(g, grid, (hipStream_t)stream); }
  else if (N == 32) { if (yraw) launch_fw<32, true>(g, grid, (hipStream_t)stream); else launch_fw<32, false>(g, grid, (hipStream_t)stream); }
  else { if (yraw) launch_fw<16, true>(g, grid, (hipStream_t)stream); else launch_fw<16, false>(g, grid, (hipStream_t)stream); }
  return tss::check_last("fc1d_wgrad");
}

/* the same two operators on the layer's own weight tensor (torch's [N][Cin][1][3] / [N][Cin][3][1]), no permuted copy: 1 when the lean
 * kernels cover the shape (then tss_conv1d3_fwd_w / tss_conv1d3_bwd_data_w may be called), 0: permute and use tss_conv1d3_fwd / _bwd_data */
extern "C" int tss_conv1d3_lean_supported(int Cin, int N, int dtype) {
  extern int g_tss_disable_fast;
  return (dtype == TSS_BF16 && !g_tss_disable_fast && fc_enabled() && Cin == N && (N == 16 || N == 32 || N == 64)) ? 1 : 0;
}
extern "C" int tss_conv1d3_fwd_w(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                 const float* w, const float* bias, void* y, long ldy, double* stats,
                                 int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(tss_conv1d3_lean_supported(Cin, N, dtype) && (ldx % 8) == 0 && ldx >= Cin && (ldy % 4) == 0 && ldy >= N && dil >= 1 &&
              (axis == 0 || axis == 1) && w, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, (double)B * H * W * (Cin + N) * 2.0, 2.0 * B * H * W * 3.0 * Cin * N);
  TSS_REQUIRE(tss_fc1d_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w, 1, bias, y, ldy, stats, B, H, W, Cin, N, axis, dil, (hipStream_t)stream),
              TSS_ERR_ALIGN);
  return tss::check_last("fc1d_fwd");
}
extern "C" int tss_conv1d3_bwd_data_w(const void* e, long lde, const void* yraw, long ldyr,
                                      const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                                      const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                      void* e_in, long ldei, double* bstats,
                                      int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(tss_conv1d3_lean_supported(Cin, N, dtype) && (lde % 8) == 0 && lde >= N && (ldei % 4) == 0 && ldei >= Cin && dil >= 1 &&
              (axis == 0 || axis == 1) && w && (!bstats || xraw), TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, (double)B * H * W * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * 2.0,
                      2.0 * B * H * W * 3.0 * Cin * N);
  TSS_REQUIRE(tss_fc1d_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, 1, xraw, ldx, in_mean, in_scale, in_bias, in_relu, e_in, ldei, bstats,
                                B, H, W, Cin, N, axis, dil, (hipStream_t)stream), TSS_ERR_ALIGN);
  return tss::check_last("fc1d_bwd_data");
}

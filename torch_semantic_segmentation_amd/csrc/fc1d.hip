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
  const float* w;                                     // [3][C outputs][C contraction] f32
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
      const float* src = g.w + ((long)tap * C + n) * C + c;
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w; o[4] = (T)b.x; o[5] = (T)b.y; o[6] = (T)b.z; o[7] = (T)b.w;
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

bool fc_enabled() {
  static int v = -1;
  if (v < 0) { const char* s = getenv("TSS_FC1D"); v = (s && s[0] == '0') ? 0 : 1; }
  return v != 0;
}

}  // namespace

// forward / backward-data of the three-tap layers on the lean kernel; false: shape not covered, the caller takes the generic kernel
bool tss_fc1d_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                  const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                  int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream) {
  if (!fc_enabled() || Cin != N || (N != 16 && N != 32 && N != 64) || (ldx % 8) || (ldy % 4) || !tss::aligned16(x) || !tss::aligned16(w_tnc) ||
      (reinterpret_cast<uintptr_t>(y) & 7u) || (long)B * H * W <= 0)
    return false;
  FcArgs g = {};
  g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis; g.tap_sign = 1;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  return dispatch_fc<0>(g, N, stream);
}

bool tss_fc1d_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                       const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                       const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                       void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream) {
  if (!fc_enabled() || Cin != N || (N != 16 && N != 32 && N != 64) || (lde % 8) || (ldei % 4) || !tss::aligned16(e) || !tss::aligned16(w_tcn) ||
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

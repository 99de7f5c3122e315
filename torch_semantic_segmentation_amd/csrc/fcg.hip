// Lean bf16 kernels of ESNet's wide factorized layers: nn.Conv2d(C, C, (1,T) / (T,1), dilation d) with T = 5 taps at 64 channels
// (FCUBlock(64, 5), TSS/models/esnet.py:83-123) and T = 3 taps at 128 channels (FPCUBlock(128, [2, 5, 9]), esnet.py:126-166) --
// forward, backward-data and weight gradient.  Same scheme as fc1d.hip (the B operand of v_mfma_f32_16x16x32_bf16 is one 16-byte vector
// of an NHWC row, loaded by the lane that feeds it), with the contraction walked TAP BY TAP: 3 C / 5 C contraction channels per pixel do
// not fit the registers at once (12 / 10 k-steps x 2 operand sets), so a tile's taps are groups of C / 32 k-steps -- load, convert, and the
// next tap's (or the next tile's first tap's) loads fly under this tap's MFMAs; the accumulators persist across the taps.  The prologue
// constants of a lane's channel vectors live in LDS (4 x 24 registers at 128 channels otherwise); 128-channel layers run 8 waves per block so
// that the 96 KB of weights in LDS are shared by twice as many waves.
#include "common.h"

namespace {

typedef bf16_t T_;
struct GArgs {
  int B, H, W, D, axis, tap_sign;                     // axis 0: taps along W, 1: along H; source = p + tap_sign * (tap - T/2) * D
  const T_* a0; long lda0; const T_* a1; long lda1;   // fwd: x   bwd: e, yraw
  const float* c0; const float* c1; const float* c2; const float* c3; int a_relu;
  const float* w; long w_os, w_ks, w_ts;              // f32 weights: element (tap, output o, contraction k) at w[o * w_os + k * w_ks + tap * w_ts]
  const float* bias;
  T_* y; long ldy; double* stats;
  const T_* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// MODE 0: forward (a = relu?((x - c1) * c0 + c2));  1: backward, g = c0 * e;  2: backward, g = c0 * (e - c2) + c1 * (y - c3)
// NSPL = 2: a PAIR of waves shares a tile, each owning half of the output channels (the 128-channel backward with two operands per tap:
// statistics, accumulators and mask operand of all 8 fragments do not fit 256 registers next to the operands)
template <int C, int T, int MODE, int MT, int NW, int NSPL = 1>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void fcg_kernel(const GArgs g) {
  constexpr int NF = C / 16, NFW = NF / NSPL, NKG = C / 32, TWV = 16 * MT, NTH = NW * 64, RPB = NW / NSPL;
  constexpr bool BWD = MODE != 0;
  extern __shared__ __align__(16) unsigned char smem[];
  // the small tables first: their addresses stay within the 64 KB immediate range of a ds_read (behind 96 KB of weights every access needed an
  // address register of its own -- 40 of them, spilled)
  float* Kc = reinterpret_cast<float*>(smem);                                   // [3][C]: k0, k1, kadd of the prologue
  float* Ec = Kc + 3 * C;                                                       // [4][C]: bias, mask mean / scale / shift
  float* red = Ec + 4 * C;                                                      // [NW][2][C]
  uint4* Wl = reinterpret_cast<uint4*>(smem + (3 + 4 + 2 * NW) * C * 4);        // [T][NF][NKG][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;

  for (int e = tid; e < T * NF * NKG * 64; e += NTH) {
    const int f = e >> 6, l = e & 63;
    const int ks = f % NKG, i = (f / NKG) % NF, tap = f / (NKG * NF);
    const int n = i * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8;
    const float* src = g.w + (long)n * g.w_os + (long)k * g.w_ks + (long)tap * g.w_ts;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T_)src[(long)j * g.w_ks];
    Wl[e] = *reinterpret_cast<const uint4*>(&o);
  }
  for (int c = tid; c < C; c += NTH) {
    const float c0v = g.c0 ? g.c0[c] : 1.f, c1v = g.c1 ? g.c1[c] : 0.f, c2v = g.c2 ? g.c2[c] : 0.f, c3v = (MODE == 2 && g.c3) ? g.c3[c] : 0.f;
    Kc[c] = c0v;
    Kc[C + c] = MODE == 2 ? c1v : 0.f;
    Kc[2 * C + c] = MODE == 0 ? c2v - c1v * c0v : (MODE == 1 ? 0.f : -(c0v * c2v) - c1v * c3v);
    float eb = 0.f, em = 0.f, es = 1.f, eh = 0.f;
    if (g.bias) eb = g.bias[c];
    if (BWD && g.xm) { if (g.mm) em = g.mm[c]; if (g.ms) es = g.ms[c]; if (g.mb) eh = g.mb[c]; }
    Ec[c] = eb; Ec[C + c] = em; Ec[2 * C + c] = es; Ec[3 * C + c] = eh;
  }
  bool plain;
  if (MODE == 0) plain = !g.c0 && !g.c1 && !g.c2 && !g.a_relu;
  else if (MODE == 1) plain = !g.c0;
  else plain = false;
  const float relu_lo = (MODE == 0 && g.a_relu) ? 0.f : -TSS_INF;

  const int i0 = (wave % NSPL) * NFW, wrow = wave / NSPL;      // this wave's output fragments, its row inside the block's row group
  float st1[NFW][4], st2[NFW][4];
#pragma unroll
  for (int i = 0; i < NFW; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  const int tpr = (g.W + TWV - 1) / TWV;
  const long rows = (long)g.B * g.H;
  const long nblk = ((rows + RPB - 1) / RPB) * tpr;      // block-tiles: (group of NW image rows, column range); wave w takes row NW * group + w
  const int lim = g.axis ? g.H : g.W;
  const long pstep = g.axis ? g.W : 1;

  uint4 ra[MT][NKG], rb[MODE == 2 ? MT : 1][MODE == 2 ? NKG : 1];
  uint2 rxm[BWD ? MT : 1][BWD ? NFW : 1];
  uint32_t okb = 0;
  long pc_l = 0; int x0_l = 0, yy_l = 0; bool live_l = false;       // tile whose loads are in flight
  long pc_c = 0; int x0_c = 0; bool live_c = false;                  // tile being accumulated

#define FG_GEOM(BT)                                        \
  {                                                          \
    const long grp = (BT) / tpr;                             \
    const int tx = (int)((BT) - grp * tpr);                  \
    const long by = grp * RPB + wrow;                        \
    live_l = by < rows;                                      \
    const long byc = live_l ? by : rows - 1;                 \
    yy_l = (int)(byc % g.H);                                 \
    x0_l = tx * TWV;                                         \
    pc_l = byc * g.W + x0_l;                                 \
  }
#define FG_ISSUE(TAP)                                                                                         \
  {                                                                                                             \
    const int sh = g.tap_sign * ((TAP) - T / 2) * g.D;                                                          \
    okb = 0;                                                                                                    \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                           \
      const int px = m * 16 + fr;                                                                               \
      const int t = (g.axis ? yy_l : x0_l + px) + sh;                                                           \
      const bool ok = x0_l + px < g.W && t >= 0 && t < lim;                                                     \
      const long q = ok ? pc_l + px + sh * pstep : pc_l;                                                        \
      okb |= ok ? (1u << m) : 0u;                                                                               \
      _Pragma("unroll") for (int ks = 0; ks < NKG; ++ks) {                                                     \
        ra[m][ks] = *reinterpret_cast<const uint4*>(g.a0 + q * g.lda0 + ks * 32 + fq * 8);                      \
        if (MODE == 2) rb[m][ks] = *reinterpret_cast<const uint4*>(g.a1 + q * g.lda1 + ks * 32 + fq * 8);       \
      }                                                                                                         \
    }                                                                                                           \
  }

  long bt = blockIdx.x;
  int tap = 0;
  if (bt < nblk) { FG_GEOM(bt); FG_ISSUE(0); }
  __syncthreads();

  f32x4 acc[MT][NFW];
  while (bt < nblk) {
    if (tap == 0) {
      pc_c = pc_l; x0_c = x0_l; live_c = live_l;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NFW; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 op[MT][NKG];
    asm volatile("" ::: "memory");                    // the prologue constants are re-read from LDS per tap, not kept in 12 NKG registers
#pragma unroll
    for (int ks = 0; ks < NKG; ++ks) {
      float k0[8], k1[8], kadd[8];
      if (C == 128) asm volatile("" ::: "memory");      // one k-step's constants at a time (the scheduler would fetch all four sets first: 96 registers)
      if (!plain) {
        const int c = ks * 32 + fq * 8;
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
          V4<float>::load(Kc + c + h, k0 + h); V4<float>::load(Kc + 2 * C + c + h, kadd + h);
          if (MODE == 2) V4<float>::load(Kc + C + c + h, k1 + h);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        uint4 r = ra[m][ks];
        if (!plain) {
          const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra[m][ks]);
          const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb[MODE == 2 ? m : 0][MODE == 2 ? ks : 0]);
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float lo = blo(ua[h]) * k0[2 * h] + kadd[2 * h];
            float hi = bhi(ua[h]) * k0[2 * h + 1] + kadd[2 * h + 1];
            if (MODE == 2) { lo += blo(ub[h]) * k1[2 * h]; hi += bhi(ub[h]) * k1[2 * h + 1]; }
            if (MODE == 0) { lo = fmaxf(lo, relu_lo); hi = fmaxf(hi, relu_lo); }
            o[2 * h] = (T_)lo; o[2 * h + 1] = (T_)hi;
          }
          r = *reinterpret_cast<const uint4*>(&o);
        }
        if (!((okb >> m) & 1u)) r = make_uint4(0u, 0u, 0u, 0u);
        op[m][ks] = *reinterpret_cast<const bf16x8*>(&r);
      }
    }
    // the mask operand of THIS tile's epilogue, then the next tap's (or the next tile's first tap's) loads
    if (BWD && tap == 0 && g.xm) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int px = m * 16 + fr;
        const long p = pc_c + ((x0_c + px < g.W) ? px : 0);
#pragma unroll
        for (int i = 0; i < NFW; ++i) rxm[m][i] = *reinterpret_cast<const uint2*>(g.xm + p * g.ldxm + (i0 + i) * 16 + fq * 4);
      }
    }
    int ntap = tap + 1; long nbt = bt;
    if (ntap == T) { ntap = 0; nbt = bt + gridDim.x; }
    if (nbt < nblk) {
      if (ntap == 0) FG_GEOM(nbt);
      FG_ISSUE(ntap);
    }
    asm volatile("" ::: "memory");
    const uint4* wt = Wl + tap * (NF * NKG * 64);
#pragma unroll
    for (int ks = 0; ks < NKG; ++ks)
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        const uint4 wr = wt[((i0 + i) * NKG + ks) * 64 + lane];
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wr);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, op[m][ks], acc[m][i], 0, 0, 0);
      }
    if (tap == T - 1 && live_c) {
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        const int nl = (i0 + i) * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(Ec + nl);
        const float cbias[4] = {e0.x, e0.y, e0.z, e0.w};
        float cmm[4] = {0.f, 0.f, 0.f, 0.f}, cms[4] = {1.f, 1.f, 1.f, 1.f}, cmb[4] = {0.f, 0.f, 0.f, 0.f};
        if (BWD) {
          const float4 e1 = *reinterpret_cast<const float4*>(Ec + C + nl);
          const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * C + nl);
          const float4 e3 = *reinterpret_cast<const float4*>(Ec + 3 * C + nl);
          cmm[0] = e1.x; cmm[1] = e1.y; cmm[2] = e1.z; cmm[3] = e1.w;
          cms[0] = e2.x; cms[1] = e2.y; cms[2] = e2.z; cms[3] = e2.w;
          cmb[0] = e3.x; cmb[1] = e3.y; cmb[2] = e3.z; cmb[3] = e3.w;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int px = m * 16 + fr;
          if (x0_c + px < g.W) {
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
              for (int q = 0; q < 4; ++q) o[q] = (T_)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * xc[q]; }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T_)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
            }
            *reinterpret_cast<bf16x4*>(g.y + (pc_c + px) * g.ldy + nl) = o;
          }
        }
      }
    }
    tap = ntap; bt = nbt;
  }
#undef FG_GEOM
#undef FG_ISSUE

  if (g.stats) {
    if (NSPL > 1) {                                    // a wave fills only its own channels: the others must read as 0
      __syncthreads();
      for (int i = tid; i < NW * 2 * C; i += NTH) red[i] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
        if (fr == 0) { red[(wave * 2 + 0) * C + (i0 + i) * 16 + fq * 4 + q] = u; red[(wave * 2 + 1) * C + (i0 + i) * 16 + fq * 4 + q] = w2; }
      }
    __syncthreads();
    if (tid < C) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += (double)red[(w * 2 + 0) * C + tid]; b += (double)red[(w * 2 + 1) * C + tid]; }
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

template <int C, int T, int NW> constexpr int fg_smem() { return T * (C / 16) * (C / 32) * 64 * 16 + (3 + 4 + 2 * NW) * C * 4; }

template <int C, int T, int MODE, int MT, int NW, int NSPL = 1>
void launch_fg(const GArgs& g, hipStream_t stream) {
  constexpr int smem = fg_smem<C, T, NW>();
  static tss::DevOnce attr;
  static int per_cu = 0;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fcg_kernel<C, T, MODE, MT, NW, NSPL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fcg_kernel<C, T, MODE, MT, NW, NSPL>, NW * 64, smem) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb > 2 ? 2 : nb;
  }
  const long tpr = (g.W + 16 * MT - 1) / (16 * MT);
  const long nblk = (((long)g.B * g.H + NW / NSPL - 1) / (NW / NSPL)) * tpr;
  long grid = 256L * per_cu;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;
  if (grid > nblk) grid = nblk;
  hipLaunchKernelGGL((fcg_kernel<C, T, MODE, MT, NW, NSPL>), dim3((int)grid), dim3(NW * 64), smem, stream, g);
}

template <int MODE>
bool dispatch_fg(const GArgs& g, int C, int T, hipStream_t stream) {
  if (C == 64 && T == 5) { launch_fg<64, 5, MODE, MODE == 2 ? 2 : 4, 4>(g, stream); return true; }
  if (C == 128 && T == 3) { launch_fg<128, 3, MODE, MODE == 0 ? 2 : 1, 8, MODE == 2 ? 2 : 1>(g, stream); return true; }
  return false;
}

// ---- weight gradient in one sweep (the scheme of fc1d_wgrad_kernel: pixel-major operands through the transposed-read LDS image),
// T taps, NW waves: wave w owns output-channel fragment w and all C / 16 x T (input fragment, tap) accumulators
typedef __attribute__((ext_vector_type(4))) short v4s;
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
  union { v4s h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lo);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)hi);
  return u.v;
}

struct GwArgs {
  long P; int B, H, W, D, axis;
  const T_* e; long lde; const T_* y; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  const T_* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu;
  float* ws;                                           // [gridDim.x][C * C * T], torch's [N][C][taps] order
};

template <int C, int T, bool HASY>
__global__ __launch_bounds__(C * 4, 1) void fcg_wgrad_kernel(const GwArgs g) {
  constexpr int NF = C / 16, NV = C / 8, NTH = C * 4, PT = NTH / NV, NIMG = ((1 + T) * NV + 15) / 16;      // one pixel x 8 channels per thread and stage
  constexpr int BUF = NIMG * PT * 256, NKS = PT / 32;
  static_assert(NTH / 64 == NF && PT == 32, "one wave per output fragment, 32-pixel stages");
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

  const int fi = wave;
  int troffG[2], troffA[T][NF][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = fq * 8 + 4 * h + (fr >> 2);
    troffG[h] = (fi >> 3) * PT * 256 + img_off(row, (fi & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int F = (1 + t) * NF + j;
        troffA[t][j][h] = (F >> 3) * PT * 256 + img_off(row, (F & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
      }
  }
  const int stG = (cv >> 4) * PT * 256 + img_off(r, cv & 15);
  int stA[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { const int gc = (1 + t) * NV + cv; stA[t] = (gc >> 4) * PT * 256 + img_off(r, gc & 15); }

  f32x4 acc[T][NF];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + gridDim.x - 1) / gridDim.x;
  const long s_begin = (long)blockIdx.x * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;
  const int lim = g.axis ? g.H : g.W;
  const long pstep = g.axis ? g.W : 1;

  uint4 re, ry, rx[T];
  uint32_t okb = 0;                                    // bits 0..T-1: tap inside the image; bit 15: the pixel exists
#define GW_ISSUE(S)                                                                                    \
  {                                                                                                      \
    const long p = (S) * PT + r;                                                                         \
    const bool in = p < g.P;                                                                             \
    const long pcl = in ? p : g.P - 1;                                                                   \
    const int xx = (int)(pcl % g.W);                                                                     \
    const int yy = (int)((pcl / g.W) % g.H);                                                             \
    okb = in ? 0x8000u : 0u;                                                                             \
    re = *reinterpret_cast<const uint4*>(g.e + pcl * g.lde + cv * 8);                                    \
    if (HASY) ry = *reinterpret_cast<const uint4*>(g.y + pcl * g.ldyr + cv * 8);                         \
    _Pragma("unroll") for (int t = 0; t < T; ++t) {                                                     \
      const int sh = (t - T / 2) * g.D;                                                                  \
      const int c1 = (g.axis ? yy : xx) + sh;                                                            \
      const bool ok = in && c1 >= 0 && c1 < lim;                                                         \
      okb |= ok ? (1u << t) : 0u;                                                                        \
      rx[t] = *reinterpret_cast<const uint4*>(g.x + (ok ? pcl + sh * pstep : pcl) * g.ldx + cv * 8);     \
    }                                                                                                    \
  }

  if (s_begin < s_end) GW_ISSUE(s_begin);
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
          o[2 * h] = (T_)lo; o[2 * h + 1] = (T_)hi;
        }
        og = *reinterpret_cast<const uint4*>(&o);
      }
      if (!(okb & 0x8000u)) og = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(img + stG) = og;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        uint4 oa = rx[t];
        if (!aplain) {
          const uint32_t* ux = reinterpret_cast<const uint32_t*>(&rx[t]);
          bf16x8 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            o[2 * h] = (T_)fmaxf(blo(ux[h]) * as[2 * h] + ab[2 * h], relu_lo);
            o[2 * h + 1] = (T_)fmaxf(bhi(ux[h]) * as[2 * h + 1] + ab[2 * h + 1], relu_lo);
          }
          oa = *reinterpret_cast<const uint4*>(&o);
        }
        if (!((okb >> t) & 1u)) oa = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(img + stA[t]) = oa;
      }
    }
    if (s + 1 < s_end) GW_ISSUE(s + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const unsigned char* base = img + ks * 32 * 256;
      const bf16x8 gA = tr_pair(base + troffG[0], base + troffG[1]);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const bf16x8 aB = tr_pair(base + troffA[t][j][0], base + troffA[t][j][1]);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gA, aB, acc[t][j], 0, 0, 0);
        }
    }
    b ^= 1;
  }
#undef GW_ISSUE
  float* row = g.ws + (long)blockIdx.x * (T * C * C);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) row[((long)(16 * fi + 4 * fq + q) * C + 16 * j + fr) * T + t] = acc[t][j][q];
}

template <int C, int T> constexpr int gw_smem() { return 2 * ((((1 + T) * (C / 8) + 15) / 16) * 32 * 256); }

int gw_rows(long P) {
  const long nstage = (P + 31) / 32;
  long grid = 256;
  if (grid > (nstage + 7) / 8) grid = (nstage + 7) / 8;
  return (int)(grid < 1 ? 1 : grid);
}

template <int C, int T, bool HASY>
void launch_gw(const GwArgs& g, int grid, hipStream_t stream) {
  constexpr int smem = gw_smem<C, T>();
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fcg_wgrad_kernel<C, T, HASY>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL((fcg_wgrad_kernel<C, T, HASY>), dim3(grid), dim3(C * 4), smem, stream, g);
}

bool fg_enabled() {
  static int v = -1;
  if (v < 0) { const char* s = getenv("TSS_FCG"); v = (s && s[0] == '0') ? 0 : 1; }
  return v != 0;
}
bool fg_covered(int Cin, int N, int T) { return fg_enabled() && Cin == N && ((N == 64 && T == 5) || (N == 128 && T == 3)); }

}  // namespace

// forward / backward-data on [T][out][contraction] f32 weights (tss_permute_wtaps); false: shape not covered
bool tss_fcg_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                 const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                 int B, int H, int W, int Cin, int N, int T, int axis, int dil, hipStream_t stream) {
  if (!fg_covered(Cin, N, T) || (ldx % 8) || (ldy % 4) || !tss::aligned16(x) || !w_tnc || (reinterpret_cast<uintptr_t>(y) & 7u) || (long)B * H * W <= 0)
    return false;
  GArgs g = {};
  g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis; g.tap_sign = 1;
  g.a0 = (const T_*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.w_os = Cin; g.w_ks = 1; g.w_ts = (long)N * Cin;
  g.bias = bias; g.y = (T_*)y; g.ldy = ldy; g.stats = stats;
  return dispatch_fg<0>(g, N, T, stream);
}

bool tss_fcg_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                      const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                      const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                      void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int T, int axis, int dil, hipStream_t stream) {
  if (!fg_covered(Cin, N, T) || (lde % 8) || (ldei % 4) || !tss::aligned16(e) || !w_tcn || (reinterpret_cast<uintptr_t>(e_in) & 7u) ||
      (long)B * H * W <= 0)
    return false;
  if (yraw && ((ldyr % 8) || !tss::aligned16(yraw) || !ga || !gb || !gce || !gmu)) return false;
  if (xraw && ((ldx % 4) || (reinterpret_cast<uintptr_t>(xraw) & 7u))) return false;
  GArgs g = {};
  g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis; g.tap_sign = -1;
  g.a0 = (const T_*)e; g.lda0 = lde; g.a1 = (const T_*)yraw; g.lda1 = ldyr;
  g.c0 = ga;
  if (yraw) { g.c1 = gb; g.c2 = gce; g.c3 = gmu; }
  g.w = w_tcn; g.w_os = N; g.w_ks = 1; g.w_ts = (long)Cin * N;
  g.y = (T_*)e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = (const T_*)xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  return yraw ? dispatch_fg<2>(g, N, T, stream) : dispatch_fg<1>(g, N, T, stream);
}

/* weight gradient of a square T-tap layer (64 channels x 5 taps, 128 channels x 3 taps; bf16; csrc/fcg.hip) in one sweep: per-block rows
 * of partial sums (T * N * Cin floats, torch's [N][Cin][taps] order) in ws[tss_convtap_bwd_weight_rows(...)][T*N*Cin], added by
 * tss_dw_reduce_many.  rows == 0: shape not covered. */
extern "C" int tss_convtap_bwd_weight_rows(long P, int Cin, int N, int T, int dtype) {
  extern int g_tss_disable_fast;
  if (dtype != TSS_BF16 || g_tss_disable_fast || !fg_covered(Cin, N, T) || P <= 0) return 0;
  return gw_rows(P);
}

extern "C" int tss_convtap_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                            const float* ga, const float* gb, const float* gce, const float* gmu,
                                            const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                                            int in_relu, float* ws, int B, int H, int W, int Cin, int N, int T, int axis, int dil, int dtype,
                                            void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(fg_covered(Cin, N, T) && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin && dil >= 1 && (axis == 0 || axis == 1) &&
              e && xraw && ws && (long)B * H * W > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(xraw) && (!yraw || tss::aligned16(yraw)), TSS_ERR_ALIGN);
  GwArgs g = {};
  g.P = (long)B * H * W; g.B = B; g.H = H; g.W = W; g.D = dil; g.axis = axis;
  g.e = (const T_*)e; g.lde = lde; g.y = (const T_*)yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = (const T_*)xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.ws = ws;
  const int grid = gw_rows(g.P);
  tss::ProfScope prof(TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream, (double)g.P * N * (yraw ? 3 : 2) * 2.0, 2.0 * g.P * T * N * N);
  if (N == 64) { if (yraw) launch_gw<64, 5, true>(g, grid, (hipStream_t)stream); else launch_gw<64, 5, false>(g, grid, (hipStream_t)stream); }
  else { if (yraw) launch_gw<128, 3, true>(g, grid, (hipStream_t)stream); else launch_gw<128, 3, false>(g, grid, (hipStream_t)stream); }
  return tss::check_last("fcg_wgrad");
}

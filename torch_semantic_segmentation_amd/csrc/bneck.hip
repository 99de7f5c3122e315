// Fused backward of the expand half of an inverted residual (BottleneckBlock, TSS/models/fastscnn.py:138-161,
// TSS/models/contextnet.py:129-147):   x --1x1 (Cin -> M = 6 Cin)--> y1 --BN1+ReLU--> a1 --depthwise 3x3 (stride s)--> y2.
//
// The two M-channel tensors of that pair (y1, and the gradient e1 that travels back from the depthwise layer to the 1x1
// layer) are six times larger than anything else in the block; the unfused backward touches them seven times (depthwise
// weight gradient reads y1; depthwise input gradient reads y1 and writes e1; the 1x1 weight gradient and the 1x1 input
// gradient each read e1 and y1): 1.0 ms of a 6.4 ms FastSCNN step at 8 x 3 x 1024 x 2048.  MFMA utilisation of the step is
// 3-5 % (profiles/r01_pmc_mfma.txt), so recomputing is free: here neither tensor is read or written at all.
//
//   y1 tile  = W1 x         recomputed per tile on the matrix cores from the (6x smaller) block input, rounded to bf16 -- bit
//                           for bit the tensor the forward kernel stored (same MFMA, same operand order)
//   e1 tile  = relu'(a1) * dw^T(g2)   by a 3x3 stencil over a window of g2 = BN2'(e2, y2) staged in LDS
//
// and three sweeps consume them in the MFMA accumulator layout (a lane owns 4 consecutive channels of one pixel):
//   pass 0 (chunk-major)  BatchNorm-1 backward sums  sum(e1), sum(e1 (y1 - mean1))  -> slab rows, and the depthwise weight
//                         gradient dWdw[c][t] = sum_q g2[q - off_t][c] a1[q][c]      -> workspace rows
//   pass 1 (chunk-major)  dW1[m][k] = sum_p g1[p][m] x[p][k],  g1 = ga1 (e1 - ce1) + gb1 (y1 - mean1): e1^T and x^T tiles in
//                         LDS, contraction over the 64 pixels of a tile on the matrix cores, accumulators live across tiles
//   pass 2 (tile-major)   dx[p][k] = sum_m g1[p][m] W1[m][k]: accumulators live across the channel chunks of a tile
// (pass 1 needs the coefficients pass 0 produces, so they cannot be one sweep; chunk-major = a block owns 64 of the M
// channels and a share of the tiles, so that no M x Cin accumulator set has to live in one block.)
//
// Tile = 8 x 8 input pixels, chunk = 64 of the M channels, 256 threads = 4 waves; wave w owns channels [16 w, 16 w + 16) of
// the chunk for all 64 pixels (four 16 x 16 fragments).
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256, MC = 64, TP = 64;
constexpr int GS = MC + 8;    // row pitch (elements) of the [pixel][chunk channel] LDS tiles: 144 B, conflict-free 16-byte reads
constexpr int PS = TP + 8;    // row pitch of the [channel][pixel] LDS tiles
constexpr int NCST = 18;      // per-chunk constant rows in LDS (9 taps, BN1 forward x3, BN1 backward x3, BN2 backward x3)

struct BnArgs {
  const T* x; long ldx;
  const T* w1b;                 // bf16 [M][Cin]
  const T* w1tb;                // bf16 [Cin][M]
  const float* mean1; const float* scale1; const float* beta1;
  const float* wdw;             // f32 [M][9]
  const T* e2; long lde2; const T* y2; long ldy2;
  const float* ga2; const float* gb2; const float* gce2; const float* gmu2;
  const float* ga1; const float* gb1; const float* gce1;
  double* bstats1; float* ws_dwdw; float* ws_dw1;
  T* dx; long lddx;
  int B, H, W, Cin, M, Ho, Wo;
  int kwp, xs, nchunk, nsplit, tiles_x, tiles_per_img; long ntiles;
};

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_round(float v) { return (float)(T)v; }

struct TileGeom { long b; int iy0, ix0; };

// pixel order inside the 8 x 8 tile.  Stride 1: row-major.  Stride 2: the four 16-pixel MFMA fragments are the four parity
// classes (py & 1, px & 1) -- every lane of a fragment then sees the SAME set of depthwise taps (1, 2, 2 or 4 of the 9),
// so the stencil runs 9 tap evaluations per 4 fragments instead of 36 with per-lane selects.
template <int S>
__device__ __forceinline__ void pix_of(int p, int& py, int& px) {
  if (S == 1) { py = p >> 3; px = p & 7; }
  else { const int pf = p >> 4, fr = p & 15; py = (fr >> 2) * 2 + (pf >> 1); px = (fr & 3) * 2 + (pf & 1); }
}
__device__ __forceinline__ TileGeom tile_geom(const BnArgs& g, long t) {
  TileGeom tg;
  tg.b = t / g.tiles_per_img;
  const int r = (int)(t - tg.b * g.tiles_per_img);
  const int ty = r / g.tiles_x;
  tg.iy0 = ty * 8; tg.ix0 = (r - ty * g.tiles_x) * 8;
  return tg;
}

// x tile (8 x 8 pixels x Cin) -> Xs[p][k] (row pitch g.xs, zero beyond Cin up to kwp)
template <int S, bool WITH_T>
__device__ __forceinline__ void load_x_tile(const BnArgs& g, const TileGeom& tg, T* Xs, T* XT, int tid) {
  const int nvec = g.Cin >> 3, nvecp = g.kwp >> 3;
  const int total = TP * nvecp;                    // <= 1024
  uint4 r[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    const int p = v / nvecp, cv = v - p * nvecp;
    const bool ok = v < total && cv < nvec;
    int py, px;
    pix_of<S>(ok ? p : 0, py, px);
    const long pix = ((tg.b * g.H + tg.iy0 + py) * (long)g.W + tg.ix0 + px);
    r[u] = *reinterpret_cast<const uint4*>(g.x + pix * g.ldx + (ok ? cv * 8 : 0));
    if (!ok) r[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    if (v < total) {
      const int p = v / nvecp, cv = v - p * nvecp;
      *reinterpret_cast<uint4*>(Xs + p * g.xs + cv * 8) = r[u];
    }
  }
}

// XT[k][p] = Xs[p][k] (pass 1: the pixel index is the contraction of the weight gradient).  Lane = pixel, so the 2-byte
// stores of a wave go to 64 consecutive pixels of one row: conflict-free.  (Transposing while the tile is loaded -- lanes =
// channel vectors of a few pixels -- put 64 lanes on two LDS banks: the 1x1 weight-gradient sweep took 2.2x the statistics
// sweep.)  Call between the barrier that publishes Xs and the barrier in front of the MFMA that reads XT.
__device__ __forceinline__ void transpose_x(const BnArgs& g, const T* Xs, T* XT, int tid) {
  const int p = tid & 63, nvecp = g.kwp >> 3;
  unsigned short* xt = reinterpret_cast<unsigned short*>(XT);
  for (int cv = tid >> 6; cv < nvecp; cv += NT / 64) {
    const uint4 r = *reinterpret_cast<const uint4*>(Xs + p * g.xs + cv * 8);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      xt[(cv * 8 + 2 * h) * PS + p] = (unsigned short)(w[h] & 0xffffu);
      xt[(cv * 8 + 2 * h + 1) * PS + p] = (unsigned short)(w[h] >> 16);
    }
  }
}

// W1 rows [m0, m0 + MC) -> W1s[n][k] (row pitch g.xs), zero beyond M / Cin
__device__ __forceinline__ void stage_w1(const BnArgs& g, int m0, T* W1s, int tid) {
  const int nvec = g.Cin >> 3, nvecp = g.kwp >> 3;
  const int total = MC * nvecp;
  uint4 r[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    const int n = v / nvecp, cv = v - n * nvecp;
    const bool ok = v < total && cv < nvec && m0 + n < g.M;
    r[u] = *reinterpret_cast<const uint4*>(g.w1b + (ok ? (long)(m0 + n) * g.Cin + cv * 8 : 0));
    if (!ok) r[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    if (v < total) { const int n = v / nvecp, cv = v - n * nvecp; *reinterpret_cast<uint4*>(W1s + n * g.xs + cv * 8) = r[u]; }
  }
}

// W1^T rows [0, kwp) x columns [m0, m0 + MC) -> W1T[k][m] (row pitch GS), zero beyond Cin / M
__device__ __forceinline__ void stage_w1t(const BnArgs& g, int m0, T* W1T, int tid) {
  const int total = g.kwp * (MC / 8);              // <= 1024
  uint4 r[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    const int k = v >> 3, cv = v & 7;
    const bool ok = v < total && k < g.Cin && m0 + cv * 8 < g.M;
    r[u] = *reinterpret_cast<const uint4*>(g.w1tb + (ok ? (long)k * g.M + m0 + cv * 8 : 0));
    if (!ok) r[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = tid + u * NT;
    if (v < total) *reinterpret_cast<uint4*>(W1T + (v >> 3) * GS + (v & 7) * 8) = r[u];
  }
}

// per-chunk constants -> Cst[row][MC]: rows 0..8 depthwise taps, 9 s1, 10 sh1 = beta1 - mean1 s1, 11 mean1, 12 ga1, 13 gb1,
// 14 kd1 = -ga1 ce1 - gb1 mean1, 15 ga2, 16 gb2, 17 kd2 = -ga2 ce2 - gb2 mu2.  Channels >= M get zeros everywhere (their W1 rows
// are zero too, so y1 = a1 = e1 = g1 = 0).
__device__ __forceinline__ void stage_consts(const BnArgs& g, int m0, float* Cst, int tid) {
  if (tid < MC) {
    const int c = m0 + tid;
    const bool in = c < g.M;
    const int cc = in ? c : 0;
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = g.wdw[(long)cc * 9 + t];
    const float s1 = g.scale1[cc], m1 = g.mean1[cc], b1 = g.beta1[cc];
    const float a2 = g.ga2[cc], b2 = g.gb2[cc], c2 = g.gce2[cc], u2 = g.gmu2[cc];
    const float a1 = g.ga1 ? g.ga1[cc] : 0.f, bb1 = g.gb1 ? g.gb1[cc] : 0.f, c1 = g.gce1 ? g.gce1[cc] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) Cst[t * MC + tid] = in ? w[t] : 0.f;
    Cst[9 * MC + tid] = in ? s1 : 0.f;
    Cst[10 * MC + tid] = in ? b1 - m1 * s1 : 0.f;
    Cst[11 * MC + tid] = in ? m1 : 0.f;
    Cst[12 * MC + tid] = in ? a1 : 0.f;
    Cst[13 * MC + tid] = in ? bb1 : 0.f;
    Cst[14 * MC + tid] = in ? -(a1 * c1) - bb1 * m1 : 0.f;
    Cst[15 * MC + tid] = in ? a2 : 0.f;
    Cst[16 * MC + tid] = in ? b2 : 0.f;
    Cst[17 * MC + tid] = in ? -(a2 * c2) - b2 * u2 : 0.f;
  }
}

// g2 = ga2 e2 + gb2 y2 + kd2 over the window of OUTPUT pixels whose taps reach the tile, chunk channels, as bf16 into
// Gw[window pixel][channel] (zero outside the image).  Stride 1: 10 x 10 window from (iy0 - 1, ix0 - 1); stride 2: 5 x 5
// from (iy0 / 2, ix0 / 2).  Needs Cst rows 15..17 (barrier before the call).
template <int S>
__device__ __forceinline__ void stage_window(const BnArgs& g, const TileGeom& tg, int m0, const float* Cst, T* Gw, int tid) {
  constexpr int WW = (S == 1) ? 10 : 5, WPX = WW * WW, NV = WPX * (MC / 8), NU = (NV + NT - 1) / NT;
  const int oy0 = (S == 1) ? tg.iy0 - 1 : tg.iy0 / 2, ox0 = (S == 1) ? tg.ix0 - 1 : tg.ix0 / 2;
  uint4 re[NU], ry[NU];
  bool okv[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int v = tid + u * NT;
    const int wp = v >> 3, cv = v & 7;
    const int wy = wp / WW, wx = wp - wy * WW;
    const int oy = oy0 + wy, ox = ox0 + wx;
    const bool ok = v < NV && oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo && m0 + cv * 8 < g.M;
    okv[u] = ok;
    const long q = ok ? ((tg.b * g.Ho + oy) * (long)g.Wo + ox) : 0;
    const int co = ok ? m0 + cv * 8 : 0;
    re[u] = *reinterpret_cast<const uint4*>(g.e2 + q * g.lde2 + co);
    ry[u] = *reinterpret_cast<const uint4*>(g.y2 + q * g.ldy2 + co);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int v = tid + u * NT;
    if (v < NV) {
      const int wp = v >> 3, cv = v & 7;
      const uint32_t ue[4] = {re[u].x, re[u].y, re[u].z, re[u].w}, uy[4] = {ry[u].x, ry[u].y, ry[u].z, ry[u].w};
      const float4 a0 = *reinterpret_cast<const float4*>(Cst + 15 * MC + cv * 8), a1 = *reinterpret_cast<const float4*>(Cst + 15 * MC + cv * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(Cst + 16 * MC + cv * 8), b1 = *reinterpret_cast<const float4*>(Cst + 16 * MC + cv * 8 + 4);
      const float4 k0 = *reinterpret_cast<const float4*>(Cst + 17 * MC + cv * 8), k1 = *reinterpret_cast<const float4*>(Cst + 17 * MC + cv * 8 + 4);
      const float ga[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float kd[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      bf16x8 o;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float lo = ga[2 * h] * bf_lo(ue[h]) + (gb[2 * h] * bf_lo(uy[h]) + kd[2 * h]);
        const float hi = ga[2 * h + 1] * bf_hi(ue[h]) + (gb[2 * h + 1] * bf_hi(uy[h]) + kd[2 * h + 1]);
        o[2 * h] = (T)(okv[u] ? lo : 0.f);
        o[2 * h + 1] = (T)(okv[u] ? hi : 0.f);
      }
      *reinterpret_cast<bf16x8*>(Gw + wp * GS + cv * 8) = o;
    }
  }
}

// one y1 fragment of this wave: channels [16 wave, 16 wave + 16) of the chunk x pixels [16 pf, 16 pf + 16):
// acc[q] = y1[pixel 16 pf + fr][channel 16 wave + 4 fq + q].  The four fragments of a tile are produced one at a time inside a
// ROLLED loop: unrolled, the compiler hoists the 4 x 9 LDS reads of the stencils and the kernel spills 100-400 B per lane.
__device__ __forceinline__ f32x4 mfma_y1(const BnArgs& g, const T* Xs, const T* W1s, int wave, int fr, int fq, int pf) {
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const T* wrow = W1s + (wave * 16 + fr) * g.xs + fq * 8;
  const T* xrow = Xs + (pf * 16 + fr) * g.xs + fq * 8;
  const int nks = g.kwp >> 5;
  for (int ks = 0; ks < nks; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc, 0, 0, 0);
  }
  return acc;
}

// Per-lane constants of the 4 channels a lane owns in the current chunk
struct LaneC { const float* w9; float s1[4], sh1[4], mu1[4], ga1[4], gb1[4], kd1[4]; };   // w9: Cst + cl (taps stay in LDS)
__device__ __forceinline__ void lane_consts(const float* Cst, int cl, LaneC& L) {
  L.w9 = Cst + cl;
  const float4 a = *reinterpret_cast<const float4*>(Cst + 9 * MC + cl), b = *reinterpret_cast<const float4*>(Cst + 10 * MC + cl);
  const float4 c = *reinterpret_cast<const float4*>(Cst + 11 * MC + cl), d = *reinterpret_cast<const float4*>(Cst + 12 * MC + cl);
  const float4 e = *reinterpret_cast<const float4*>(Cst + 13 * MC + cl), f = *reinterpret_cast<const float4*>(Cst + 14 * MC + cl);
  L.s1[0] = a.x; L.s1[1] = a.y; L.s1[2] = a.z; L.s1[3] = a.w;   L.sh1[0] = b.x; L.sh1[1] = b.y; L.sh1[2] = b.z; L.sh1[3] = b.w;
  L.mu1[0] = c.x; L.mu1[1] = c.y; L.mu1[2] = c.z; L.mu1[3] = c.w; L.ga1[0] = d.x; L.ga1[1] = d.y; L.ga1[2] = d.z; L.ga1[3] = d.w;
  L.gb1[0] = e.x; L.gb1[1] = e.y; L.gb1[2] = e.z; L.gb1[3] = e.w; L.kd1[0] = f.x; L.kd1[1] = f.y; L.kd1[2] = f.z; L.kd1[3] = f.w;
}

// The 3x3 stencil at the lane's pixel (py, px) for its 4 channels: e1[q] = sum_t w[t][q] g2[window(t)][q].  WG: the same g2 values
// times the activated input ap[q] are added to the depthwise weight-gradient accumulators accw[t][q] on the way (nothing is
// kept alive in between: 36 fewer registers than returning the nine values).  cl = lane's first channel inside the chunk.
template <int S, bool WG>
__device__ __forceinline__ void stencil(const T* Gw, int pf, int fr, int cl, const LaneC& L, float e1[4], const float ap[4],
                                        float (*accw)[4]) {
  constexpr int WW = (S == 1) ? 10 : 5;
  int py, px;
  pix_of<S>(pf * 16 + fr, py, px);
#pragma unroll
  for (int q = 0; q < 4; ++q) e1[q] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int wy, wx;
      if (S == 1) { wy = py + 2 - ky; wx = px + 2 - kx; }
      else {
        // pf is wave-uniform, so this is a scalar branch: only the taps whose parity matches the fragment's are evaluated
        if ((((pf >> 1) + 1 - ky) | ((pf & 1) + 1 - kx)) & 1) continue;
        wy = (py + 1 - ky) >> 1; wx = (px + 1 - kx) >> 1;   // = output coordinate - window origin
      }
      const uint2 r = *reinterpret_cast<const uint2*>(Gw + (wy * WW + wx) * GS + cl);
      const float4 w4 = *reinterpret_cast<const float4*>(L.w9 + (ky * 3 + kx) * MC);
      const float gv[4] = {bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y)}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        e1[q] += gv[q] * wv[q];
        if (WG) accw[ky * 3 + kx][q] += gv[q] * ap[q];
      }
    }
}

// ------------------------------------------------------------------------------------------------- chunk-major sweeps
// grid = nchunk x nsplit; block (chunk, split) sweeps tiles split, split + nsplit, ...
template <int S, int PASS>
__global__ __launch_bounds__(NT, 2) void bneck_bwd_chunk_kernel(const BnArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* W1s = Xs + TP * g.xs;
  T* Gw = W1s + MC * g.xs;
  float* Cst = reinterpret_cast<float*>(Gw + 100 * GS);
  T* E1T = reinterpret_cast<T*>(Cst + NCST * MC);      // pass 1 only
  T* XT = E1T + MC * PS;                                 // pass 1 only: [kwp][PS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int chunk = (int)blockIdx.x % g.nchunk, split = (int)blockIdx.x / g.nchunk;
  const int m0 = chunk * MC;
  const int cl = wave * 16 + fq * 4;                  // lane's first channel inside the chunk

  stage_w1(g, m0, W1s, tid);
  stage_consts(g, m0, Cst, tid);
  __syncthreads();
  LaneC L;
  lane_consts(Cst, cl, L);

  float se[4] = {0.f, 0.f, 0.f, 0.f}, sey[4] = {0.f, 0.f, 0.f, 0.f};
  float accw[PASS == 1 ? 9 : 1][4];                  // depthwise weight gradient: carried by pass 1 (pass 0 + these 36 registers spills)
#pragma unroll
  for (int t = 0; t < (PASS == 1 ? 9 : 1); ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) accw[t][q] = 0.f;
  constexpr int NKF = 8;                               // kwp / 16 <= 8
  f32x4 acc1[PASS == 1 ? NKF : 1];
#pragma unroll
  for (int kf = 0; kf < (PASS == 1 ? NKF : 1); ++kf) acc1[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nkf = g.kwp >> 4;

  for (long t = split; t < g.ntiles; t += g.nsplit) {
    const TileGeom tg = tile_geom(g, t);
    __syncthreads();                                   // previous tile's reads of Xs / Gw / E1T / XT are done
    load_x_tile<S, PASS == 1>(g, tg, Xs, XT, tid);
    stage_window<S>(g, tg, m0, Cst, Gw, tid);
    __syncthreads();
    if constexpr (PASS == 1) transpose_x(g, Xs, XT, tid);
#pragma unroll 1
    for (int pf = 0; pf < 4; ++pf) {
      const f32x4 acc = mfma_y1(g, Xs, W1s, wave, fr, fq, pf);
      const int p = pf * 16 + fr;
      float y1[4], a1[4], ap[4], e1[4], g1v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y1[q] = bf_round(acc[q]);                      // the stored forward tensor, bit for bit
        a1[q] = y1[q] * L.s1[q] + L.sh1[q];
        ap[q] = fmaxf(a1[q], 0.f);
      }
      stencil<S, PASS == 1>(Gw, pf, fr, cl, L, e1, ap, accw);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float e = a1[q] > 0.f ? bf_round(e1[q]) : 0.f;
        if constexpr (PASS == 0) {
          se[q] += e;
          sey[q] += e * (y1[q] - L.mu1[q]);
        } else {
          g1v[q] = L.ga1[q] * e + (L.gb1[q] * y1[q] + L.kd1[q]);
        }
      }
      if constexpr (PASS == 1) {
        unsigned short* et = reinterpret_cast<unsigned short*>(E1T);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const T b = (T)g1v[q]; et[(cl + q) * PS + p] = *reinterpret_cast<const unsigned short*>(&b); }
      }
    }
    if constexpr (PASS == 1) {
      __syncthreads();
      // T[n][k] += sum_p g1^T[n][p] x^T[k][p]: wave owns n-fragment `wave`, all k-fragments
      const T* arow = E1T + (wave * 16 + fr) * PS + fq * 8;
      const T* brow = XT + fr * PS + fq * 8;
#pragma unroll
      for (int ks = 0; ks < TP / 32; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          if (kf < nkf) {
            const bf16x8 bfv = *reinterpret_cast<const bf16x8*>(brow + kf * 16 * PS + ks * 32);
            acc1[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc1[kf], 0, 0, 0);
          }
        }
      }
    }
  }

  if constexpr (PASS == 0) {
    // sums over the 16 pixel lanes of a row (equal fq): channel (cl + q) ends up in the row's lane 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = row16_sum(se[q]), b = row16_sum(sey[q]);
      const int c = m0 + cl + q;
      if (fr == 0 && c < g.M) {
        g.bstats1[(long)split * 2 * g.M + c] = (double)a;
        g.bstats1[(long)split * 2 * g.M + g.M + c] = (double)b;
        for (int r = split + g.nsplit; r < TSS_STAT_SLABS; r += g.nsplit) {
          g.bstats1[(long)r * 2 * g.M + c] = 0.0;
          g.bstats1[(long)r * 2 * g.M + g.M + c] = 0.0;
        }
      }
    }
  } else {
    // partial dW1 tile of this block: ws_dw1[(chunk * nsplit + split)][n][k], n = 16 wave + 4 fq + q, k = 16 kf + fr
    float* slot = g.ws_dw1 + ((long)chunk * g.nsplit + split) * MC * g.kwp;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (kf < nkf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[(cl + q) * g.kwp + kf * 16 + fr] = acc1[kf][q];
      }
    }
    // depthwise weight gradient rows: sums over the 16 pixel lanes of a row (equal fq), channel (cl + q) in the row's lane 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = m0 + cl + q;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) {
        const float w = row16_sum(accw[tt][q]);
        if (fr == 0 && c < g.M) g.ws_dwdw[(long)split * g.M * 9 + (long)c * 9 + tt] = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- tile-major sweep: dx
template <int S>
__global__ __launch_bounds__(NT, 2) void bneck_bwd_dx_kernel(const BnArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* W1s = Xs + TP * g.xs;
  T* Gw = W1s + MC * g.xs;
  float* Cst = reinterpret_cast<float*>(Gw + 100 * GS);
  T* G = reinterpret_cast<T*>(Cst + NCST * MC);        // [TP][GS]
  T* W1T = G + TP * GS;                                  // [kwp][GS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int cl = wave * 16 + fq * 4;
  constexpr int NKF = 8;
  const int nkf = g.kwp >> 4;

  for (long t = blockIdx.x; t < g.ntiles; t += gridDim.x) {
    const TileGeom tg = tile_geom(g, t);
    __syncthreads();
    load_x_tile<S, false>(g, tg, Xs, nullptr, tid);
    f32x4 dacc[NKF];
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) dacc[kf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int chunk = 0; chunk < g.nchunk; ++chunk) {
      const int m0 = chunk * MC;
      __syncthreads();                                 // previous chunk's MFMA reads of G / W1T (and stencil reads of Gw) are done
      stage_w1(g, m0, W1s, tid);
      stage_w1t(g, m0, W1T, tid);
      stage_consts(g, m0, Cst, tid);
      __syncthreads();
      stage_window<S>(g, tg, m0, Cst, Gw, tid);
      LaneC L;
      lane_consts(Cst, cl, L);
      __syncthreads();
#pragma unroll 1
      for (int pf = 0; pf < 4; ++pf) {
        const f32x4 acc = mfma_y1(g, Xs, W1s, wave, fr, fq, pf);
        const int p = pf * 16 + fr;
        float e1[4];
        stencil<S, false>(Gw, pf, fr, cl, L, e1, nullptr, nullptr);
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float y1 = bf_round(acc[q]);
          const float a1 = y1 * L.s1[q] + L.sh1[q];
          const float e = a1 > 0.f ? bf_round(e1[q]) : 0.f;
          o[q] = (T)(L.ga1[q] * e + (L.gb1[q] * y1 + L.kd1[q]));
        }
        *reinterpret_cast<bf16x4*>(G + p * GS + cl) = o;
      }
      __syncthreads();
      // dx[k][p] += sum_m W1^T[k][m] g1[p][m]: wave owns pixel fragment `wave`, all k-fragments
      const T* grow = G + (wave * 16 + fr) * GS + fq * 8;
      const T* wrow = W1T + fr * GS + fq * 8;
#pragma unroll
      for (int ks = 0; ks < MC / 32; ++ks) {
        const bf16x8 gf = *reinterpret_cast<const bf16x8*>(grow + ks * 32);
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          if (kf < nkf) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + kf * 16 * GS + ks * 32);
            dacc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf, dacc[kf], 0, 0, 0);
          }
        }
      }
    }
    // lane: channels k = 16 kf + 4 fq + q of pixel p = 16 wave + fr
    int py, px;
    pix_of<S>(wave * 16 + fr, py, px);
    const long pix = (tg.b * g.H + tg.iy0 + py) * (long)g.W + tg.ix0 + px;
    T* drow = g.dx + pix * g.lddx + fq * 4;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      if (kf < nkf && kf * 16 + fq * 4 < g.Cin) {
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (T)dacc[kf][q];
        *reinterpret_cast<bf16x4*>(drow + kf * 16) = o;
      }
    }
  }
}

// dst[i] += sum over rows of ws[row][n] (depthwise weight gradient rows; dW1 slots through a strided view)
__global__ __launch_bounds__(256) void bneck_rows_reduce_kernel(const float* ws, float* dst, int n, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r0 = 0; r0 < rows; r0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[(long)(r0 + u < rows ? r0 + u : 0) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (r0 + u < rows) ? v[u] : 0.f;
  }
  dst[i] += s;
}

// dW1[m][k] += sum_split ws[(chunk, split)][m - 64 chunk][k]
__global__ __launch_bounds__(256) void bneck_dw1_reduce_kernel(const float* ws, float* dw1, int M, int Cin, int kwp, int nsplit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * Cin) return;
  const int m = i / Cin, k = i - m * Cin;
  const int chunk = m / MC, n = m - chunk * MC;
  const float* col = ws + ((long)chunk * nsplit * MC + n) * kwp + k;
  float s = 0.f;
  for (int r0 = 0; r0 < nsplit; r0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(long)(r0 + u < nsplit ? r0 + u : 0) * MC * kwp];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (r0 + u < nsplit) ? v[u] : 0.f;
  }
  dw1[i] += s;
}

int fill(BnArgs& g, int B, int H, int W, int Cin, int M, int stride) {
  if (B <= 0 || (H % 8) || (W % 8) || Cin < 8 || Cin > 128 || (Cin % 8) || M < 8 || (M % 8) || (stride != 1 && stride != 2)) return TSS_ERR_SHAPE;
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.M = M;
  g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
  g.kwp = (Cin + 31) & ~31; g.xs = g.kwp + 8;
  g.nchunk = (M + MC - 1) / MC;
  g.tiles_x = W / 8; g.tiles_per_img = (H / 8) * g.tiles_x; g.ntiles = (long)B * g.tiles_per_img;
  long ns = 768 / g.nchunk;            // 3 resident blocks per CU x 256 CUs: one full round of blocks, no tail
  if (ns > g.ntiles) ns = g.ntiles;
  if (ns > TSS_STAT_SLABS) ns = TSS_STAT_SLABS;
  if (ns < 1) ns = 1;
  g.nsplit = (int)ns;
  return TSS_OK;
}

size_t smem_chunk(const BnArgs& g, bool pass1) {
  size_t b = (size_t)(TP + MC) * g.xs * 2 + 100 * GS * 2 + NCST * MC * 4;
  if (pass1) b += (size_t)MC * PS * 2 + (size_t)g.kwp * PS * 2;
  return b;
}
size_t smem_dx(const BnArgs& g) { return (size_t)(TP + MC) * g.xs * 2 + 100 * GS * 2 + NCST * MC * 4 + (size_t)TP * GS * 2 + (size_t)g.kwp * GS * 2; }

template <typename K>
void set_smem(K kernel, size_t bytes, tss::DevOnce& once) {
  if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)bytes;
}

}  // namespace

extern "C" {

int tss_bneck_bwd_supported(int B, int H, int W, int Cin, int M, int stride, int dtype) {
  BnArgs g = {};
  return dtype == TSS_BF16 && fill(g, B, H, W, Cin, M, stride) == TSS_OK ? 1 : 0;
}

/* f32 workspace sizes: which = 0: depthwise weight-gradient rows [nsplit][9 M]; 1: dW1 slots [nchunk][nsplit][64][kwp] */
long tss_bneck_bwd_ws(int B, int H, int W, int Cin, int M, int stride, int which) {
  BnArgs g = {};
  if (fill(g, B, H, W, Cin, M, stride) != TSS_OK) return 0;
  return which == 0 ? (long)g.nsplit * M * 9 : (long)g.nchunk * g.nsplit * MC * g.kwp;
}

int tss_bneck_bwd_stats(const void* x, long ldx, const void* w1_bf16, const float* mean1, const float* scale1, const float* beta1,
                        const float* wdw, const void* e2, long lde2, const void* y2, long ldy2,
                        const float* ga2, const float* gb2, const float* gce2, const float* gmu2,
                        double* bstats1, int B, int H, int W, int Cin, int M, int stride, void* stream) {
  BnArgs g = {};
  const int rc = fill(g, B, H, W, Cin, M, stride);
  if (rc) return rc;
  TSS_REQUIRE(x && w1_bf16 && mean1 && scale1 && beta1 && wdw && e2 && y2 && ga2 && gb2 && gce2 && gmu2 && bstats1, TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= Cin && (lde2 % 8) == 0 && lde2 >= M && (ldy2 % 8) == 0 && ldy2 >= M, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(w1_bf16) && tss::aligned16(e2) && tss::aligned16(y2), TSS_ERR_ALIGN);
  g.x = (const T*)x; g.ldx = ldx; g.w1b = (const T*)w1_bf16; g.mean1 = mean1; g.scale1 = scale1; g.beta1 = beta1; g.wdw = wdw;
  g.e2 = (const T*)e2; g.lde2 = lde2; g.y2 = (const T*)y2; g.ldy2 = ldy2; g.ga2 = ga2; g.gb2 = gb2; g.gce2 = gce2; g.gmu2 = gmu2;
  g.bstats1 = bstats1;
  const size_t sm = smem_chunk(g, false);
  hipStream_t st = (hipStream_t)stream;
  const double bytes = ((double)B * H * W * Cin * g.nchunk + 2.0 * B * g.Ho * g.Wo * M) * 2.0;
  tss::ProfScope prof(TSS_K_DWCONV_BWD_WEIGHT, st, bytes, 2.0 * B * H * W * (double)Cin * M);
  static tss::DevOnce o1, o2;
  if (stride == 1) { set_smem(bneck_bwd_chunk_kernel<1, 0>, sm, o1); hipLaunchKernelGGL((bneck_bwd_chunk_kernel<1, 0>), dim3(g.nchunk * g.nsplit), dim3(NT), sm, st, g); }
  else { set_smem(bneck_bwd_chunk_kernel<2, 0>, sm, o2); hipLaunchKernelGGL((bneck_bwd_chunk_kernel<2, 0>), dim3(g.nchunk * g.nsplit), dim3(NT), sm, st, g); }
  return tss::check_last("bneck_bwd_stats");
}

int tss_bneck_bwd_weight(const void* x, long ldx, const void* w1_bf16, const float* mean1, const float* scale1, const float* beta1,
                         const float* wdw, const void* e2, long lde2, const void* y2, long ldy2,
                         const float* ga2, const float* gb2, const float* gce2, const float* gmu2,
                         const float* ga1, const float* gb1, const float* gce1,
                         float* ws_dw1, float* dw1, float* ws_dwdw, float* dwdw,
                         int B, int H, int W, int Cin, int M, int stride, void* stream) {
  BnArgs g = {};
  const int rc = fill(g, B, H, W, Cin, M, stride);
  if (rc) return rc;
  TSS_REQUIRE(x && w1_bf16 && mean1 && scale1 && beta1 && wdw && e2 && y2 && ga2 && gb2 && gce2 && gmu2 && ga1 && gb1 && gce1 && ws_dw1 && dw1 &&
              ws_dwdw && dwdw, TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= Cin && (lde2 % 8) == 0 && lde2 >= M && (ldy2 % 8) == 0 && ldy2 >= M, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(w1_bf16) && tss::aligned16(e2) && tss::aligned16(y2), TSS_ERR_ALIGN);
  g.x = (const T*)x; g.ldx = ldx; g.w1b = (const T*)w1_bf16; g.mean1 = mean1; g.scale1 = scale1; g.beta1 = beta1; g.wdw = wdw;
  g.e2 = (const T*)e2; g.lde2 = lde2; g.y2 = (const T*)y2; g.ldy2 = ldy2; g.ga2 = ga2; g.gb2 = gb2; g.gce2 = gce2; g.gmu2 = gmu2;
  g.ga1 = ga1; g.gb1 = gb1; g.gce1 = gce1; g.ws_dw1 = ws_dw1; g.ws_dwdw = ws_dwdw;
  const size_t sm = smem_chunk(g, true);
  hipStream_t st = (hipStream_t)stream;
  const double bytes = ((double)B * H * W * Cin * g.nchunk + 2.0 * B * g.Ho * g.Wo * M) * 2.0;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_WEIGHT, st, bytes, 4.0 * B * H * W * (double)Cin * M);
  static tss::DevOnce o1, o2;
  if (stride == 1) { set_smem(bneck_bwd_chunk_kernel<1, 1>, sm, o1); hipLaunchKernelGGL((bneck_bwd_chunk_kernel<1, 1>), dim3(g.nchunk * g.nsplit), dim3(NT), sm, st, g); }
  else { set_smem(bneck_bwd_chunk_kernel<2, 1>, sm, o2); hipLaunchKernelGGL((bneck_bwd_chunk_kernel<2, 1>), dim3(g.nchunk * g.nsplit), dim3(NT), sm, st, g); }
  hipLaunchKernelGGL(bneck_dw1_reduce_kernel, dim3((M * Cin + 255) / 256), dim3(256), 0, st, ws_dw1, dw1, M, Cin, g.kwp, g.nsplit);
  hipLaunchKernelGGL(bneck_rows_reduce_kernel, dim3((M * 9 + 255) / 256), dim3(256), 0, st, ws_dwdw, dwdw, M * 9, g.nsplit);
  return tss::check_last("bneck_bwd_weight");
}

int tss_bneck_bwd_data(const void* x, long ldx, const void* w1_bf16, const void* w1t_bf16,
                       const float* mean1, const float* scale1, const float* beta1,
                       const float* wdw, const void* e2, long lde2, const void* y2, long ldy2,
                       const float* ga2, const float* gb2, const float* gce2, const float* gmu2,
                       const float* ga1, const float* gb1, const float* gce1,
                       void* dx, long lddx, int B, int H, int W, int Cin, int M, int stride, void* stream) {
  BnArgs g = {};
  const int rc = fill(g, B, H, W, Cin, M, stride);
  if (rc) return rc;
  TSS_REQUIRE(x && w1_bf16 && w1t_bf16 && mean1 && scale1 && beta1 && wdw && e2 && y2 && ga2 && gb2 && gce2 && gmu2 && ga1 && gb1 && gce1 && dx, TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= Cin && (lde2 % 8) == 0 && lde2 >= M && (ldy2 % 8) == 0 && ldy2 >= M && (lddx % 4) == 0 && lddx >= Cin, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(w1_bf16) && tss::aligned16(w1t_bf16) && tss::aligned16(e2) && tss::aligned16(y2) && tss::aligned16(dx), TSS_ERR_ALIGN);
  g.x = (const T*)x; g.ldx = ldx; g.w1b = (const T*)w1_bf16; g.w1tb = (const T*)w1t_bf16;
  g.mean1 = mean1; g.scale1 = scale1; g.beta1 = beta1; g.wdw = wdw;
  g.e2 = (const T*)e2; g.lde2 = lde2; g.y2 = (const T*)y2; g.ldy2 = ldy2; g.ga2 = ga2; g.gb2 = gb2; g.gce2 = gce2; g.gmu2 = gmu2;
  g.ga1 = ga1; g.gb1 = gb1; g.gce1 = gce1; g.dx = (T*)dx; g.lddx = lddx;
  const size_t sm = smem_dx(g);
  hipStream_t st = (hipStream_t)stream;
  long grid = g.ntiles < 768 ? g.ntiles : 768;
  const double bytes = (2.0 * B * H * W * Cin + 2.0 * B * g.Ho * g.Wo * M * (stride == 1 ? 1.56 : 1.56)) * 2.0;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, st, bytes, 4.0 * B * H * W * (double)Cin * M);
  static tss::DevOnce o1, o2;
  if (stride == 1) { set_smem(bneck_bwd_dx_kernel<1>, sm, o1); hipLaunchKernelGGL((bneck_bwd_dx_kernel<1>), dim3((int)grid), dim3(NT), sm, st, g); }
  else { set_smem(bneck_bwd_dx_kernel<2>, sm, o2); hipLaunchKernelGGL((bneck_bwd_dx_kernel<2>), dim3((int)grid), dim3(NT), sm, st, g); }
  return tss::check_last("bneck_bwd_data");
}

}  // extern "C"

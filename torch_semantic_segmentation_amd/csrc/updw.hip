// Bilinear upsample (align_corners=True) + dilated depthwise 3x3 convolution as ONE kernel, forward and backward: the
// low-resolution branch of the feature-fusion modules
//   FastSCNN   TSS/models/fastscnn.py:74-76    nn.UpsamplingBilinear2d(4) -> DWConv2dBlock(dilation=4, padding=4)
//   ContextNet TSS/models/contextnet.py:110-122 F.interpolate(size of the spatial branch) -> DWConvBlock(dilation=4, padding=4)
// The source is a 1/32-resolution map (4 MB at 8 x 128 x 32 x 64, L2 / Infinity-Cache resident), its x4 copy a 67 MB
// tensor that the layer-by-layer path writes, reads back in the depthwise forward, and reads twice more in the depthwise
// backward (weight gradient + input gradient: the dilation-4 layer was the last one on the strip kernels of dwconv.hip).
// Here the upsampled tensor never exists:
//   forward   reads the source, writes y (+ the statistics slab row)                         1 tensor pass instead of 3
//   backward  reads e, y, the source; writes the gradient of the UPSAMPLED map (bf16) which   3 passes instead of 6
//             the separable transposed interpolation (resample.hip) folds back to the source resolution
//
// Walk: a block owns a 64-channel slice of one column strip of one image and walks down the rows of ONE residue class
// modulo the dilation (r, r + D, r + 2D, ...): vertically that is a dilation-1 convolution -- every new row feeds exactly
// the three output rows under construction (registers), horizontally the taps sit D pixels apart in the row that was
// just parked in LDS (one barrier per row, two row buffers).  The interpolated row is produced on the fly from the four
// source pixels under each output pixel, with torch's index arithmetic (common.h ac_tap) and rounded to bf16 exactly
// as the materialised tensor would have been, so both paths see the same operand bits.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXD = 8;
constexpr int FPX = 32;      // forward : 8 channel lanes (8 channels each) x 32 pixels
constexpr int BPX = 16;      // backward: 16 channel lanes (4 channels each) x 16 pixels

struct UpDwArgs {
  const bf16_t* x; long ldx; int Hs, Ws;      // source [B][Hs][Ws][C]
  const float* w;                              // [C][3][3]
  bf16_t* y; long ldy; double* stats;          // forward output [B][Ho][Wo][C] + slab rows
  const bf16_t* e; long lde; const bf16_t* yr; long ldyr;             // backward: e = dL/d(BN output), raw conv output
  const float* ga; const float* gb; const float* gce; const float* gmu;
  bf16_t* eup; long ldeu; float* ws;           // gradient of the upsampled map, weight-gradient rows [nunits][C*9]
  int B, Ho, Wo, C, D;
  int nsl, nstrips, nseg, seg_steps, nunits;   // units = (image, strip, residue class, row segment); seg_steps class rows per segment
  int tile_vecs;                               // capacity of the source tile in 16-byte vectors (dynamic LDS)
};

__device__ __forceinline__ void unpack8(const uint4& r, float v[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack4(const uint2& r, float v[4]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}

// The source pixels a unit can touch -- rows [sr0, sr0 + nrows) x columns [sc0, sc0 + ncols) of its image, the 64 channels of its
// slice -- are copied to LDS ONCE, with every load of a thread in flight together; the row walk then has no global load left in it
// (a per-row fetch of the four taps was one exposed L2 round trip per row: 55 us for the 8 x 128 x 128 x 256 layer, 1.6 us per row).
// Layout: 16-byte vector (r, c, v) at index (r * ncols + c) * 8 + v, v = 8-channel vector of the slice.
struct SrcTile { int sr0, nrows, sc0, ncols; };

__device__ __forceinline__ void load_tile(const UpDwArgs& g, const SrcTile& t, int b, int sl, uint4* tile_s) {
  const int total = t.nrows * t.ncols * 8;
  const int nv = (g.C - sl * 64) >> 3;                 // valid 8-channel vectors of this slice (>= 1)
  const bf16_t* base = g.x + ((long)b * g.Hs * g.Ws) * g.ldx + sl * 64;
  for (int i0 = 0; i0 < total; i0 += NT * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = i0 + u * NT + (int)threadIdx.x;
      i = i < total ? i : total - 1;
      const int vv = i & 7, rc = i >> 3;
      const int r = rc / t.ncols, c = rc - r * t.ncols;
      v[u] = *reinterpret_cast<const uint4*>(base + ((long)(t.sr0 + r) * g.Ws + t.sc0 + c) * g.ldx + (vv < nv ? vv : 0) * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NT + (int)threadIdx.x;
      if (i < total) tile_s[i] = v[u];
    }
  }
}

// class rows [j_lo, j_hi) of residue class rho are OWNED by the unit; it walks [js, je) = one more on either side (their
// interpolated / gradient rows feed the owned output rows)
struct Walk { int rho, strip, b, j_lo, j_hi, js, je, nj; };
__device__ __forceinline__ Walk unit_walk(const UpDwArgs& g, int u) {
  Walk w;
  const int seg = u % g.nseg;
  w.rho = (u / g.nseg) % g.D;
  w.strip = (u / (g.nseg * g.D)) % g.nstrips;
  w.b = u / (g.nseg * g.D * g.nstrips);
  w.nj = w.rho < g.Ho ? (g.Ho - w.rho + g.D - 1) / g.D : 0;
  w.j_lo = seg * g.seg_steps;
  w.j_hi = w.j_lo + g.seg_steps < w.nj ? w.j_lo + g.seg_steps : w.nj;
  if (w.j_lo > w.j_hi) w.j_lo = w.j_hi;
  w.js = w.j_lo > 0 ? w.j_lo - 1 : 0;
  w.je = w.j_hi < w.nj ? w.j_hi + 1 : w.nj;
  if (w.j_lo == w.j_hi) { w.js = w.je = w.j_lo; }
  return w;
}

// ------------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(NT, 2) void updw_fwd_kernel(const UpDwArgs g) {
  extern __shared__ __align__(16) unsigned char dyn_smem[];
  uint4* tile_s = reinterpret_cast<uint4*>(dyn_smem);                               // [tile_vecs]
  bf16_t* rows_base = reinterpret_cast<bf16_t*>(tile_s + g.tile_vecs);              // [2][(FPX + 2 MAXD) * 64]
  constexpr int ROWE = (FPX + 2 * MAXD) * 64;
  const int tid = threadIdx.x, p = tid >> 3, cg = tid & 7;
  const int D = g.D;
  const int sl = (int)blockIdx.x % g.nsl, u = (int)blockIdx.x / g.nsl;
  const Walk wk = unit_walk(g, u);
  const int rho = wk.rho, b = wk.b;
  const int ch = sl * 64 + cg * 8;
  const bool ch_on = ch < g.C;
  const int c0 = ch_on ? ch : 0;
  const int x0 = wk.strip * FPX;
  const int col = x0 + p;
  const bool col_on = col < g.Wo && ch_on;
  const bool is_halo = tid < 16 * D;                  // 2 D halo pixels x 8 channel lanes: the first (<= 2) waves
  const int hp = tid >> 3;
  const int hcol = hp < D ? x0 - D + hp : x0 + FPX + (hp - D);
  const bool hcol_on = is_halo && hcol >= 0 && hcol < g.Wo && ch_on;
  const int hslot = hp < D ? hp : FPX + hp;           // pixel slot of the LDS row: slot = column - (x0 - D)
  const float sy = ac_scale(g.Hs, g.Ho), sx = ac_scale(g.Ws, g.Wo);
  // columns the tile covers: [cl, cr]; lanes outside the map read the tile's edge column and are zeroed
  const int cl = x0 - D > 0 ? x0 - D : 0, cr = x0 + FPX + D - 1 < g.Wo - 1 ? x0 + FPX + D - 1 : g.Wo - 1;
  const Tap tx = ac_tap(sx, col < cr ? col : cr, g.Ws);
  const Tap thx = ac_tap(sx, hcol < cl ? cl : (hcol > cr ? cr : hcol), g.Ws);

  SrcTile t;
  {
    t.sc0 = ac_tap(sx, cl, g.Ws).i0;
    t.ncols = ac_tap(sx, cr, g.Ws).i1 - t.sc0 + 1;
    const int r_first = rho + wk.js * D, r_last = rho + (wk.je > wk.js ? wk.je - 1 : wk.js) * D;
    t.sr0 = ac_tap(sy, r_first < g.Ho ? r_first : g.Ho - 1, g.Hs).i0;
    t.nrows = ac_tap(sy, r_last < g.Ho ? r_last : g.Ho - 1, g.Hs).i1 - t.sr0 + 1;
  }
  if (wk.je > wk.js) load_tile(g, t, b, sl, tile_s);

  float wr[9][8];
  {
    float wf[72];
#pragma unroll
    for (int q = 0; q < 18; ++q) V4<float>::load(g.w + (long)c0 * 9 + 4 * q, wf + 4 * q);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) wr[tt][j] = wf[j * 9 + tt];
  }
  float accA[8], accB[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { accA[j] = accB[j] = s1[j] = s2[j] = 0.f; }
  __syncthreads();

  // interpolated pixel from the tile, rounded to bf16 (the bits bilinear_nhwc_fwd_kernel would have stored); zero outside the map
  auto lerp_store = [&](const Tap& ty, const Tap& tc, bool on, bf16_t* dst) {
    const int r0 = (ty.i0 - t.sr0) * t.ncols, r1 = (ty.i1 - t.sr0) * t.ncols;
    const int a0 = tc.i0 - t.sc0, a1 = tc.i1 - t.sc0;
    float a[8], bq[8], c[8], d[8], o[8];
    unpack8(tile_s[(r0 + a0) * 8 + cg], a); unpack8(tile_s[(r0 + a1) * 8 + cg], bq);
    unpack8(tile_s[(r1 + a0) * 8 + cg], c); unpack8(tile_s[(r1 + a1) * 8 + cg], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = ty.l0 * (tc.l0 * a[j] + tc.l1 * bq[j]) + ty.l1 * (tc.l0 * c[j] + tc.l1 * d[j]);
      if (!on) o[j] = 0.f;
    }
    V8<bf16_t>::store(dst, o);
  };
  auto emit = [&](int r, const float (&acc)[8]) {
    if (col_on) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = (bf16_t)acc[j];
        const float rq = (float)o[j];
        s1[j] += rq; s2[j] += rq * rq;
      }
      *reinterpret_cast<bf16x8*>(g.y + ((long)b * g.Ho + r) * g.Wo * g.ldy + (col * (int)g.ldy + c0)) = o;
    }
  };

  for (int j = wk.js; j < wk.je; ++j) {
    const int R = rho + j * D;
    const Tap ty = ac_tap(sy, R, g.Hs);
    bf16_t* row = rows_base + (j & 1) * ROWE;
    lerp_store(ty, tx, col_on, row + (p + D) * 64 + cg * 8);
    if (is_halo) lerp_store(ty, thx, hcol_on, row + hslot * 64 + cg * 8);
    __syncthreads();
    float T[3][8];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const uint4 tv = *reinterpret_cast<const uint4*>(row + (p + D * m) * 64 + cg * 8);
      unpack8(tv, T[m]);
    }
    // the new row R is tap row 2 of output row R - D (complete now), tap row 1 of row R, tap row 0 of row R + D
    float accC[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      accA[q] += wr[6][q] * T[0][q]; accA[q] += wr[7][q] * T[1][q]; accA[q] += wr[8][q] * T[2][q];
      accB[q] += wr[3][q] * T[0][q]; accB[q] += wr[4][q] * T[1][q]; accB[q] += wr[5][q] * T[2][q];
      accC[q] = wr[0][q] * T[0][q]; accC[q] += wr[1][q] * T[1][q]; accC[q] += wr[2][q] * T[2][q];
    }
    if (j - 1 >= wk.j_lo) emit(R - D, accA);            // (j - 1 < j_hi always: je <= j_hi + 1)
#pragma unroll
    for (int q = 0; q < 8; ++q) { accA[q] = accB[q]; accB[q] = accC[q]; }
  }
  if (wk.je > wk.js && wk.je == wk.nj && wk.j_hi == wk.nj) emit(rho + (wk.nj - 1) * D, accA);   // no row below the last one

  // ---- statistics slab row of this unit (columns of its slice); rows no unit owns are zeroed here
  if (g.stats) {
    __syncthreads();
    float* red_s = reinterpret_cast<float*>(dyn_smem);          // [FPX][2][64] floats = 16 KB: aliases the tile / row buffers
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red_s[(p * 2 + 0) * 64 + cg * 8 + j] = col_on ? s1[j] : 0.f;
      red_s[(p * 2 + 1) * 64 + cg * 8 + j] = col_on ? s2[j] : 0.f;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const int chn = sl * 64 + c;
      if (chn < g.C) {
        double a = 0.0;
        for (int q = 0; q < FPX; ++q) a += (double)red_s[(q * 2 + which) * 64 + c];
        g.stats[(long)u * 2 * g.C + which * g.C + chn] = a;
        for (int r = u + g.nunits; r < TSS_STAT_SLABS; r += g.nunits) g.stats[(long)r * 2 * g.C + which * g.C + chn] = 0.0;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward
// One sweep over (e, y): with g = BN'(e, y) and xup the interpolated map,
//   d(xup)[q][c] = sum_{ky,kx} w[ky][kx] * g[q - D(ky-1)][c - D(kx-1)]       three rows under construction, as forward
//   dW[ky][kx]  += xup[q][c] * g[q - D(ky-1)][c - D(kx-1)]                    the SAME nine g taps, times the lane's own
//                                                                              interpolated pixels of rows R-D, R, R+D
// so only g is shared through LDS and xup is needed at the lane's own pixel only (from the source tile).  Every upsampled pixel q
// is owned by exactly one lane of one block, so every (q, tap) pair is counted once over the grid.  A lane owns FOUR channels
// (weights + weight-gradient sums: 72 registers instead of 144, two waves per SIMD), as in dw_bwd_roll_s1_kernel.
// e and y stream from HBM: their rows are requested a whole TRIP of UR steps ahead (plain loads, all unconditional, one wait per
// trip -- the depth the compiler's wait-count pass keeps); with one step of distance every step
// exposed a memory round trip (83 us for the 8 x 128 x 128 x 256 layer, 2.2 us per row).
constexpr int UR = 4;
__global__ __launch_bounds__(NT, 2) void updw_bwd_kernel(const UpDwArgs g) {
  extern __shared__ __align__(16) unsigned char dyn_smem[];
  uint4* tile_s = reinterpret_cast<uint4*>(dyn_smem);
  constexpr int ROWV = (BPX + 2 * MAXD) * 16;
  float4* rows_base = reinterpret_cast<float4*>(tile_s + g.tile_vecs);              // [2][ROWV]
  const int tid = threadIdx.x, p = tid >> 4, cg = tid & 15;
  const int D = g.D;
  const int sl = (int)blockIdx.x % g.nsl, u = (int)blockIdx.x / g.nsl;
  const Walk wk = unit_walk(g, u);
  const int rho = wk.rho, b = wk.b;
  const int ch = sl * 64 + cg * 4;
  const bool ch_on = ch < g.C;
  const int c0 = ch_on ? ch : 0;
  const int x0 = wk.strip * BPX;
  const int col = x0 + p;
  const bool col_on = col < g.Wo && ch_on;
  const bool is_halo = tid < 32 * D;
  const int hp = tid >> 4;
  const int hcol = hp < D ? x0 - D + hp : x0 + BPX + (hp - D);
  const bool hcol_on = is_halo && hcol >= 0 && hcol < g.Wo && ch_on;
  const int hslot = hp < D ? hp : BPX + hp;
  const int ccl = col < g.Wo ? col : x0, hcl = hcol_on ? hcol : x0;      // (x0 < Wo: always a valid column, and inside the tile)
  const float sy = ac_scale(g.Hs, g.Ho), sx = ac_scale(g.Ws, g.Wo);
  const Tap tx = ac_tap(sx, ccl, g.Ws);
  const bf16_t* ysrc = g.yr ? g.yr : g.e;
  const long ldyy = g.yr ? g.ldyr : g.lde;

  SrcTile t;
  {
    const int cr = x0 + BPX - 1 < g.Wo - 1 ? x0 + BPX - 1 : g.Wo - 1;
    t.sc0 = ac_tap(sx, x0 < g.Wo ? x0 : g.Wo - 1, g.Ws).i0;
    t.ncols = ac_tap(sx, cr, g.Ws).i1 - t.sc0 + 1;
    // interpolated rows of class steps js - 1 .. je (one beyond the walk on either side), clamped to the map
    const int r_first = rho + (wk.js > 0 ? wk.js - 1 : 0) * D, r_last = rho + wk.je * D;
    t.sr0 = ac_tap(sy, r_first < g.Ho ? r_first : g.Ho - 1, g.Hs).i0;
    t.nrows = ac_tap(sy, r_last < g.Ho ? r_last : g.Ho - 1, g.Hs).i1 - t.sr0 + 1;
  }

  struct EY { uint2 e, y; };
  auto fetch_ey = [&](int R, int c, EY& r) {
    const long pix = ((long)b * g.Ho + R) * g.Wo + c;
    r.e = *reinterpret_cast<const uint2*>(g.e + pix * g.lde + c0);
    r.y = *reinterpret_cast<const uint2*>(ysrc + pix * ldyy + c0);
  };
  EY nxt[UR], nxth[UR];
  auto fetch_trip = [&](int jb) {
#pragma unroll
    for (int q = 0; q < UR; ++q) {
      int j = jb + q;
      j = j < wk.je ? j : wk.je - 1;
      j = j > 0 ? j : 0;
      int R = rho + j * D;
      R = R < g.Ho ? R : g.Ho - 1;
      fetch_ey(R, ccl, nxt[q]); fetch_ey(R, hcl, nxth[q]);
    }
  };
  if (wk.je > wk.js) { fetch_trip(wk.js); load_tile(g, t, b, sl, tile_s); }

  float ca[4], cb[4], kd[4];
  {
    const float* safe = g.w;
    float t0[4], t1[4], t2[4], t3[4];
    V4<float>::load(g.ga ? g.ga + c0 : safe, t0);
    V4<float>::load(g.yr ? g.gb + c0 : safe, t1);
    V4<float>::load(g.yr ? g.gce + c0 : safe, t2);
    V4<float>::load(g.yr ? g.gmu + c0 : safe, t3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ca[j] = g.ga ? t0[j] : 1.f;
      cb[j] = g.yr ? t1[j] : 0.f;
      kd[j] = g.yr ? -(ca[j] * t2[j]) - cb[j] * t3[j] : 0.f;
    }
  }
  float wr[9][4];
  {
    float wf[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) V4<float>::load(g.w + (long)c0 * 9 + 4 * q, wf + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) wr[tt][j] = wf[j * 9 + tt];
  }
  __syncthreads();

  // the lane's own interpolated pixel of class row j (the forward kernel's operand bits); zero outside the map AND outside the rows
  // this unit owns: the weight gradient counts every interpolated pixel once over the grid, the neighbouring segment counts its own
  auto lerp = [&](int j, float (&o)[4]) {
    const int R = rho + j * D;
    const bool on = col_on && j >= wk.j_lo && j < wk.j_hi;
    const Tap ty = ac_tap(sy, on ? R : 0, g.Hs);
    const int r0 = (on ? ty.i0 - t.sr0 : 0) * t.ncols, r1 = (on ? ty.i1 - t.sr0 : 0) * t.ncols;
    const int a0 = tx.i0 - t.sc0, a1 = tx.i1 - t.sc0;
    const uint2* tv = reinterpret_cast<const uint2*>(tile_s);
    float a[4], bq[4], c[4], d[4];
    unpack4(tv[(r0 + a0) * 16 + cg], a); unpack4(tv[(r0 + a1) * 16 + cg], bq);
    unpack4(tv[(r1 + a0) * 16 + cg], c); unpack4(tv[(r1 + a1) * 16 + cg], d);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = ty.l0 * (tx.l0 * a[j] + tx.l1 * bq[j]) + ty.l1 * (tx.l0 * c[j] + tx.l1 * d[j]);
      o[j] = on ? (float)(bf16_t)v : 0.f;
    }
  };
  auto g_of = [&](const EY& r, bool on) -> float4 {
    float ev[4], yv[4], o[4];
    unpack4(r.e, ev); unpack4(r.y, yv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = ca[j] * ev[j] + (cb[j] * yv[j] + kd[j]);
      o[j] = on ? gj : 0.f;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  };

  float dwa[9][4], accP[4], accC[4], X[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    accP[j] = accC[j] = 0.f;
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) dwa[tt][j] = 0.f;
  }
  if (wk.je > wk.js) { lerp(wk.js - 1, X[0]); lerp(wk.js, X[1]); }
  auto emit = [&](int r, const float (&acc)[4]) {
    if (col_on) V4<bf16_t>::store(g.eup + ((long)b * g.Ho + r) * g.Wo * g.ldeu + (col * (int)g.ldeu + c0), acc);
  };

  for (int jb = wk.js; jb < wk.je; jb += UR) {
    EY cur[UR], curh[UR];
#pragma unroll
    for (int q = 0; q < UR; ++q) { cur[q] = nxt[q]; curh[q] = nxth[q]; }
    fetch_trip(jb + UR);                   // the next trip's rows fly under this trip's arithmetic (always issued: clamped)
#pragma unroll
    for (int q = 0; q < UR; ++q) {
      const int j = jb + q;
      if (j < wk.je) {
        const int R = rho + j * D;
        float4* row = rows_base + (j & 1) * ROWV;
        const float4 gown = g_of(cur[q], col_on);
        row[(p + D) * 16 + cg] = gown;
        if (is_halo) row[hslot * 16 + cg] = g_of(curh[q], hcol_on);
        lerp(j + 1, X[2]);
        __syncthreads();
        float G[3][4];          // g of row R at columns c - D, c, c + D
        {
          const float4 g0 = row[p * 16 + cg], g2 = row[(p + 2 * D) * 16 + cg];
          G[0][0] = g0.x; G[0][1] = g0.y; G[0][2] = g0.z; G[0][3] = g0.w;
          G[1][0] = gown.x; G[1][1] = gown.y; G[1][2] = gown.z; G[1][3] = gown.w;
          G[2][0] = g2.x; G[2][1] = g2.y; G[2][2] = g2.z; G[2][3] = g2.w;
        }
        float accN[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          // tap column kx reads g column c - D(kx - 1) = G[2 - kx]
          accP[qq] += wr[0][qq] * G[2][qq]; accP[qq] += wr[1][qq] * G[1][qq]; accP[qq] += wr[2][qq] * G[0][qq];
          accC[qq] += wr[3][qq] * G[2][qq]; accC[qq] += wr[4][qq] * G[1][qq]; accC[qq] += wr[5][qq] * G[0][qq];
          accN[qq] = wr[6][qq] * G[2][qq]; accN[qq] += wr[7][qq] * G[1][qq]; accN[qq] += wr[8][qq] * G[0][qq];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            dwa[0 + kx][qq] += X[0][qq] * G[2 - kx][qq];
            dwa[3 + kx][qq] += X[1][qq] * G[2 - kx][qq];
            dwa[6 + kx][qq] += X[2][qq] * G[2 - kx][qq];
          }
        }
        if (j - 1 >= wk.j_lo) emit(R - D, accP);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          accP[qq] = accC[qq]; accC[qq] = accN[qq];
          X[0][qq] = X[1][qq]; X[1][qq] = X[2][qq];
        }
      }
    }
  }
  if (wk.je > wk.js && wk.je == wk.nj && wk.j_hi == wk.nj) emit(rho + (wk.nj - 1) * D, accP);

  // ---- weight-gradient partial sums of the block -> its workspace row: three taps at a time through [thread][12] floats
  float* red = reinterpret_cast<float*>(dyn_smem);           // NT * 12 floats = 12 KB (host: dynamic LDS >= that)
  float* wrow = g.ws + (long)u * g.C * 9;
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += 3) {
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
      *reinterpret_cast<float4*>(red + tid * 12 + tt * 4) =
          col_on ? make_float4(dwa[t0 + tt][0], dwa[t0 + tt][1], dwa[t0 + tt][2], dwa[t0 + tt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < 64 * 3; i += NT) {
      const int c = i / 3, tt = i - c * 3;
      const int cgc = c >> 2, jj = c & 3;
      if (sl * 64 + c < g.C) {
        float sum = 0.f;
        for (int q = 0; q < BPX; ++q) sum += red[(q * 16 + cgc) * 12 + tt * 4 + jj];
        wrow[(long)(sl * 64 + c) * 9 + t0 + tt] = sum;
      }
    }
  }
}

// forward: 40 KB (two row segments per class at the benchmark's size, three blocks per CU: 55 us) is preferred over 56 KB (one segment, two
// blocks per CU: 57 us), which remains the fallback when the segments would exceed the 512 statistics slab rows (larger batches)
constexpr int FWD_TILE_BYTES = 40 * 1024, FWD_TILE_MAX = 56 * 1024, BWD_TILE_BYTES = 36 * 1024;
constexpr int BWD_MAX_UNITS = 8192;        // workspace rows of the weight gradient (the forward's units are slab rows: <= 512)
constexpr int FWD_FIXED_BYTES = 2 * (FPX + 2 * MAXD) * 64 * 2, BWD_FIXED_BYTES = 2 * (BPX + 2 * MAXD) * 16 * 16;

// segments per residue class such that a unit's source tile fits `budget` bytes, and the unit count fits the slab rows
bool geometry(UpDwArgs& g, int px, int halo, int budget, int max_units) {
  g.nsl = (g.C + 63) / 64;
  g.nstrips = (g.Wo + px - 1) / px;
  const float sy = g.Ho > 1 ? (float)(g.Hs - 1) / (float)(g.Ho - 1) : 0.f, sx = g.Wo > 1 ? (float)(g.Ws - 1) / (float)(g.Wo - 1) : 0.f;
  int ncols = (int)((px + 2 * halo - 1) * sx) + 4;
  if (ncols > g.Ws) ncols = g.Ws;
  const int njmax = (g.Ho + g.D - 1) / g.D;
  for (int nseg = 1; nseg <= njmax; ++nseg) {
    const int steps = (njmax + nseg - 1) / nseg;
    int nrows = (int)((steps + 3) * g.D * sy) + 4;               // owned steps + one walked + one interpolated row either side
    if (nrows > g.Hs) nrows = g.Hs;
    const long units = (long)g.B * g.nstrips * g.D * nseg;
    if (units > max_units) return false;
    if ((long)nrows * ncols * 128 <= budget) {
      g.nseg = nseg; g.seg_steps = steps; g.nunits = (int)units;
      g.tile_vecs = nrows * ncols * 8;
      return true;
    }
  }
  return false;
}

bool shape_ok(int B, int Hs, int Ws, int Ho, int Wo, int C, int D, int dtype) {
  if (dtype != TSS_BF16 || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return false;
  if (C < 8 || (C % 8) != 0 || D < 1 || D > MAXD) return false;
  UpDwArgs g = {};
  g.Hs = Hs; g.Ws = Ws; g.B = B; g.Ho = Ho; g.Wo = Wo; g.C = C; g.D = D;
  UpDwArgs h = g;
  return (geometry(g, FPX, D, FWD_TILE_BYTES, TSS_STAT_SLABS) || geometry(g, FPX, D, FWD_TILE_MAX, TSS_STAT_SLABS))
         && geometry(h, BPX, 0, BWD_TILE_BYTES, BWD_MAX_UNITS);
}

}  // namespace

extern "C" {

int tss_updw_supported(int B, int Hs, int Ws, int Ho, int Wo, int C, int dil, int dtype) {
  const char* sw = getenv("TSS_UPDW");         // A/B switch: 0 = upsample and depthwise layer as two operators
  if (sw && atoi(sw) == 0) return 0;
  return shape_ok(B, Hs, Ws, Ho, Wo, C, dil, dtype) ? 1 : 0;
}

/* rows of the weight-gradient workspace tss_updw_bwd writes ([rows][C*9] f32); 0: shape not covered */
int tss_updw_ws_rows(int B, int Hs, int Ws, int Ho, int Wo, int C, int dil, int dtype) {
  if (!shape_ok(B, Hs, Ws, Ho, Wo, C, dil, dtype)) return 0;
  UpDwArgs g = {};
  g.Hs = Hs; g.Ws = Ws; g.B = B; g.Ho = Ho; g.Wo = Wo; g.C = C; g.D = dil;
  return geometry(g, BPX, 0, BWD_TILE_BYTES, BWD_MAX_UNITS) ? g.nunits : 0;
}

int tss_updw_fwd(const void* x, long ldx, int Hs, int Ws, const float* w, void* y, long ldy, double* stats,
                 int B, int Ho, int Wo, int C, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(shape_ok(B, Hs, Ws, Ho, Wo, C, dil, dtype) && (ldx % 8) == 0 && (ldy % 8) == 0 && ldx >= C && ldy >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(x && w && y && tss::aligned16(x) && tss::aligned16(y) && tss::aligned16(w), TSS_ERR_ALIGN);
  UpDwArgs g = {};
  g.x = (const bf16_t*)x; g.ldx = ldx; g.Hs = Hs; g.Ws = Ws; g.w = w; g.y = (bf16_t*)y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.Ho = Ho; g.Wo = Wo; g.C = C; g.D = dil;
  // A/B: TSS_UPDW_FWD_TILE=<bytes> -- a smaller source tile means more row segments (more, shorter blocks) and a third block per CU
  static const int tile_budget = getenv("TSS_UPDW_FWD_TILE") ? atoi(getenv("TSS_UPDW_FWD_TILE")) : FWD_TILE_BYTES;
  TSS_REQUIRE(geometry(g, FPX, dil, tile_budget < FWD_TILE_MAX ? tile_budget : FWD_TILE_MAX, TSS_STAT_SLABS)
              || geometry(g, FPX, dil, FWD_TILE_MAX, TSS_STAT_SLABS), TSS_ERR_SHAPE);
  size_t smem = (size_t)g.tile_vecs * 16 + FWD_FIXED_BYTES;
  if (smem < (size_t)FPX * 2 * 64 * 4) smem = (size_t)FPX * 2 * 64 * 4;          // the statistics reduction aliases it
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(updw_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FWD_TILE_MAX + FWD_FIXED_BYTES + 1024);
  tss::ProfScope prof(TSS_K_DWCONV_FWD, (hipStream_t)stream, ((double)B * Hs * Ws + (double)B * Ho * Wo) * C * 2.0, 0);
  hipLaunchKernelGGL(updw_fwd_kernel, dim3(g.nsl * g.nunits), dim3(NT), smem, (hipStream_t)stream, g);
  return tss::check_last("updw_fwd");
}

int tss_updw_bwd(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                 const float* gmu, const float* w, const void* x, long ldx, int Hs, int Ws, void* e_up, long ldeu, float* ws,
                 int B, int Ho, int Wo, int C, int dil, int dtype, void* stream, int* rows_out) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(shape_ok(B, Hs, Ws, Ho, Wo, C, dil, dtype) && (ldx % 8) == 0 && (lde % 8) == 0 && (ldeu % 8) == 0
              && ldx >= C && lde >= C && ldeu >= C && (!yraw || ((ldyr % 8) == 0 && ldyr >= C)), TSS_ERR_SHAPE);
  TSS_REQUIRE(e && w && x && e_up && ws && tss::aligned16(e) && tss::aligned16(x) && tss::aligned16(e_up) && tss::aligned16(w)
              && (!yraw || tss::aligned16(yraw)), TSS_ERR_ALIGN);
  TSS_REQUIRE(!yraw || (gb && gce && gmu), TSS_ERR_SHAPE);
  UpDwArgs g = {};
  g.x = (const bf16_t*)x; g.ldx = ldx; g.Hs = Hs; g.Ws = Ws; g.w = w;
  g.e = (const bf16_t*)e; g.lde = lde; g.yr = (const bf16_t*)yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.eup = (bf16_t*)e_up; g.ldeu = ldeu; g.ws = ws;
  g.B = B; g.Ho = Ho; g.Wo = Wo; g.C = C; g.D = dil;
  TSS_REQUIRE(geometry(g, BPX, 0, BWD_TILE_BYTES, BWD_MAX_UNITS), TSS_ERR_SHAPE);
  if (rows_out) *rows_out = g.nunits;
  size_t smem = (size_t)g.tile_vecs * 16 + BWD_FIXED_BYTES;
  if (smem < (size_t)NT * 12 * 4) smem = (size_t)NT * 12 * 4;                    // the weight-gradient reduction aliases it
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(updw_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_TILE_BYTES + BWD_FIXED_BYTES + 1024);
  tss::ProfScope prof(TSS_K_DWCONV_BWD_DATA, (hipStream_t)stream,
                      ((double)B * Hs * Ws + (double)B * Ho * Wo * (yraw ? 3.0 : 2.0)) * C * 2.0, 0);
  hipLaunchKernelGGL(updw_bwd_kernel, dim3(g.nsl * g.nunits), dim3(NT), smem, (hipStream_t)stream, g);
  return tss::check_last("updw_bwd");
}

}  // extern "C"

// Cross-entropy over NCHW-planar logits (the caller side of the hot path: nn.CrossEntropyLoss(ignore_index)
// as the training scripts use it), plus the argmax / confusion-matrix pass of the evaluator.
// One lane owns 8 consecutive pixels of a row: every class plane is read/written as 16-byte vectors.
// forward : online log-sum-exp over the C planes -> per-pixel lse (saved), sum of losses + valid count (f64 atomics)
// backward: dlogits[c] = (exp(l_c - lse) - [c == t]) * grad_out / count, recomputed from the saved lse
#include "common.h"

namespace {

constexpr int NT = 256;

template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T* logits, const long long* target, float* lse_out,
                                                    double* acc /*[2]: loss sum, valid count*/, long B, int C,
                                                    long HW, int ignore_index) {
  __shared__ double red[2][NT / 64];
  const long groups = B * (HW / 8);
  double lsum = 0.0, lcnt = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    const T* base = logits + b * C * HW + off;
    float m[8], s[8], lt[8];
    long long t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; s[j] = 0.f; lt[j] = 0.f; t[j] = target[b * HW + off + j]; }
    for (int c = 0; c < C; ++c) {
      float v[8];
      V8<T>::load(base + (long)c * HW, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mn = fmaxf(m[j], v[j]);
        s[j] = s[j] * __expf(m[j] - mn) + __expf(v[j] - mn);
        m[j] = mn;
        if (t[j] == c) lt[j] = v[j];
      }
    }
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = m[j] + __logf(s[j]);
      if (t[j] != ignore_index && t[j] >= 0 && t[j] < C) { lsum += (double)(l[j] - lt[j]); lcnt += 1.0; }   // same validity rule as the fused head
    }
    V8<float>::store(lse_out + b * HW + off, l);
  }
  lsum = wave_sum(lsum);
  lcnt = wave_sum(lcnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = lsum; red[1][wave] = lcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < NT / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    atomicAdd(acc, a);
    atomicAdd(acc + 1, c);
  }
}

__global__ void ce_finalize_kernel(const double* acc, float* loss, float* inv_count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double n = acc[1];
    *loss = (float)(acc[0] / n);          // n == 0 -> nan, like torch
    *inv_count = n > 0.0 ? (float)(1.0 / n) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const T* logits, const long long* target, const float* lse,
                                                    const float* inv_count, const float* grad_out, T* dlogits,
                                                    long B, int C, long HW, int ignore_index) {
  const long groups = B * (HW / 8);
  const float gs = (*inv_count) * (grad_out ? *grad_out : 1.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    float l[8], w[8];
    long long t[8];
    V8<float>::load(lse + b * HW + off, l);
#pragma unroll
    for (int j = 0; j < 8; ++j) { t[j] = target[b * HW + off + j]; w[j] = (t[j] != ignore_index && t[j] >= 0 && t[j] < C) ? gs : 0.f; }
    for (int c = 0; c < C; ++c) {
      float v[8], d[8];
      V8<T>::load(logits + (b * C + c) * HW + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = (__expf(v[j] - l[j]) - (t[j] == c ? 1.f : 0.f)) * w[j];
      V8<T>::store(dlogits + (b * C + c) * HW + off, d);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Fused decoder head + loss: cross-entropy of the x`scale` bilinearly upsampled logits
// computed straight from the LOW-RES NHWC logits, loss AND gradient in ONE pass.  The (B, C, H, W) full-resolution
// logits -- 318.8 M elements at 8x1024x2048, the largest tensor of the step -- and their gradient never exist: the
// unfused path moves ~3.8 GB for them (upsample, loss, loss backward, two-pass upsample backward: 1.04 ms of a
// 10 ms step), this kernel reads the target once (134 MB) and is bound by the 19 exponentials per pixel.
//
// The mean reduction makes the gradient linear in one unknown scalar (1 / #valid pixels, times grad_out), so the
// forward pass accumulates the UNSCALED low-res gradient and backward is a scale + cast of 6 M elements.
//
//   block  = 256 consecutive output columns x a band of output rows of one image; lane = one output column.
//   The column taps (i0x, i1x, l0x, l1x) of a lane never change, and the row taps change only every ~scale rows, so
//   the lane keeps the two horizontally interpolated low-res rows aA[c], aB[c] in registers: a logit is ONE FMA,
//   z_c = l0y*aA[c] + l1y*aB[c].  Softmax, loss and dz_c = (p_c - [t == c]) stay in registers; dz is accumulated into
//   gA[c] += l0y*dz_c, gB[c] += l1y*dz_c and only when the row tap advances is a finished low-res row gathered
//   horizontally through LDS (one thread per low-res cell and class) and added to the f32 gradient (global atomics:
//   ~650 per 8 output rows of a block, each address touched by <= 4 blocks).
// Run-to-run determinism: a low-res cell receives contributions from up to four blocks (two bands x two strips).  Each
// block therefore stores ITS contribution to every (row, cell, class) of its footprint into a tile of its own (plain
// stores, every element of the footprint written exactly once, nothing zero-filled), its loss / count partial into a row
// of its own, and the backward kernel gathers the <= 4 tiles of a cell in a fixed order.  (Round 2 added them with f32
// global atomics into a zero-filled buffer: last-bit differences from run to run that a train step amplifies.)
//   ws = [nblocks][2] f64 loss rows, then [nblocks][tile_rows][tile_cells][CP] f32 tiles; block = (b * nband + band) * nstrip + strip
struct CeGeom { int nstrip, nband, band_rows, tile_rows, tile_cells; };

// MODE 0: mean cross-entropy (loss rows + unscaled gradient tiles).  MODE 1: the per-pixel cross-entropy only, written to `pix`
// ([B][H][W] f32, 0 for ignored pixels) -- the input of the OHEM selection (TSS/losses/ohem_loss.py:11-12 without the 318.8 M-element
// logits).  MODE 2: gradient tiles of a WEIGHTED sum of per-pixel losses, the weight of a pixel derived from its stored loss and
// the selection parameters `sel` = [mode, cut, weight of l > cut, weight of l == cut] (tss_ohem_select): OHEM's backward.
template <typename T, int CP, int MODE>
__global__ __launch_bounds__(NT, (CP <= 20 ? 3 : 2)) void upsample_ce_onepass_kernel(const T* low, long ldl, const long long* target,
                                                                 float* tiles, double* lossrows, int B, int C, int h, int w,
                                                                 int H, int W, int ignore_index, const CeGeom geo,
                                                                 float* pix, const float* sel) {
  const int band_rows = geo.band_rows;
  constexpr int MAXCELL = NT + 2, WTAB = 1024;
  constexpr float LOG2E = 1.44269504088896340736f;
  // Hh[k][lane][c]: the one-hot half of the gradient, sum over the rows seen so far of (row weight) * [target == c], kept
  // per low-res row (k = buffer of row rA / rB).  The softmax half lives in registers (gA, gB); the two meet at flush time.
  // The kernel is VALU-bound (SQ_ACTIVE_INST_VALU = 89 % of the SIMD cycles): a compare + select per class for the one-hot
  // term and for the target logit was 84 of the 332 vector instructions per pixel; this way it is two LDS updates, and the
  // target logits are recovered per low-res row as sum_c a[c] * Hh[c] (same products, summed in a different order).
  __shared__ __align__(16) float Hh[2][NT * CP];
  __shared__ int Li0[NT], Li1[NT], Wlo[MAXCELL], Whi[MAXCELL];
  __shared__ float Ll1[NT];
  __shared__ float Wt[WTAB];                          // gather weights [cell][lane - Wlo[cell]] (fixed for the block)
  __shared__ double red[2][NT / 64];
  const int tid = threadIdx.x;
  const int nstrip = geo.nstrip, nband = geo.nband;
  float* const tile = tiles + (long)blockIdx.x * geo.tile_rows * geo.tile_cells * CP;
  int bid = blockIdx.x;
  const int strip = bid % nstrip; bid /= nstrip;
  const int band = bid % nband;
  const long b = bid / nband;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const int x = strip * NT + tid;
  const bool xin = x < W;
  const Tap tx = ac_tap(sx, xin ? x : W - 1, w);
  const int cx0 = ac_tap(sx, strip * NT, w).i0;                     // first low-res column this strip touches
  const int xl = (strip * NT + NT - 1 < W) ? strip * NT + NT - 1 : W - 1;
  const int ncell = ac_tap(sx, xl, w).i1 - cx0 + 1;                 // <= NT + 2 for W >= w (host-checked)
  const int ya = band * band_rows;
  const int yb = ya + band_rows < H ? ya + band_rows : H;
  Li0[tid] = xin ? tx.i0 - cx0 : -1;      // lanes past the image edge match no cell
  Li1[tid] = xin ? tx.i1 - cx0 : -1;
  Ll1[tid] = tx.l1;
  // widest window of lanes whose taps can include one low-res column: uniform bound from the scale
  const int maxwin = sx > 0.f ? (int)(2.f / sx) + 6 : NT;
  const bool wtab = (long)ncell * maxwin <= WTAB;
  for (int cell = tid; cell < ncell; cell += NT) {   // lanes of this strip whose taps can include low-res column cx0 + cell
    int lo, hi;
    ac_window(sx, cx0 + cell, W, &lo, &hi);
    lo -= strip * NT; hi -= strip * NT;
    lo = lo < 0 ? 0 : lo;
    hi = hi > NT - 1 ? NT - 1 : hi;
    if (wtab && hi - lo + 1 > maxwin) hi = lo + maxwin - 1;        // cannot happen (conservative bound); keeps the table safe
    Wlo[cell] = lo;
    Whi[cell] = hi;
  }
#pragma unroll
  for (int c4 = 0; c4 < CP; c4 += 4) {
    *reinterpret_cast<float4*>(&Hh[0][tid * CP + c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&Hh[1][tid * CP + c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (wtab) {
    for (int i = tid; i < ncell * maxwin; i += NT) {
      const int cell = i / maxwin, k = i - cell * maxwin;
      const int l = Wlo[cell] + k;
      float wgt = 0.f;
      if (l <= Whi[cell]) { const float l1 = Ll1[l]; wgt = (Li0[l] == cell ? 1.f - l1 : 0.f) + (Li1[l] == cell ? l1 : 0.f); }
      Wt[i] = wgt;
    }
  }

  int rA = ac_tap(sy, ya, h).i0;
  int rB = rA + (rA < h - 1 ? 1 : 0);
  const int r_first = rA;
  float aA[CP], aB[CP], gA[CP], gB[CP];
  auto load_row = [&](int r, float* a) {   // a[c] = l0x * L[r][i0x][c] + l1x * L[r][i1x][c]
    const T* p0 = low + ((b * h + r) * (long)w + tx.i0) * ldl;
    const T* p1 = low + ((b * h + r) * (long)w + tx.i1) * ldl;
#pragma unroll
    for (int c4 = 0; c4 < CP; c4 += 4) {
      float u[4], v[4];
      V4<T>::load(p0 + c4, u);
      V4<T>::load(p1 + c4, v);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[c4 + q] = (c4 + q < C) ? (tx.l0 * u[q] + tx.l1 * v[q]) * LOG2E : -TSS_INF;
    }
  };
  float lsum = 0.f, lcnt = 0.f;     // loss sum in base-2 units (the rows carry logits * log2(e): one v_exp per class, no multiply)
  // finished low-res row r (buffer k): the lane's gradient g[] - Hh[k][] is parked in Hh[k] (plain stores), then one thread
  // per (cell, class) gathers the <= ~2*scale+4 lanes whose column taps include that cell and adds the sum to dlow; the
  // lane's target logits of the row leave the loss sum.  (LDS float atomics for the scatter were measured at ~180 cycles
  // per wave instruction: 550 us of a 790 us kernel.)
  auto flush_row = [&](int r, int k, const float* g, const float* a) {
    float* G = Hh[k];
    float zt = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < CP; c4 += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(G + tid * CP + c4);
      const float hq[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) if (c4 + q < C) zt += a[c4 + q] * hq[q];     // padding classes: a = -inf, Hh = 0
      *reinterpret_cast<float4*>(G + tid * CP + c4) = make_float4(g[c4] - hq[0], g[c4 + 1] - hq[1], g[c4 + 2] - hq[2], g[c4 + 3] - hq[3]);
    }
    lsum -= zt;
    __syncthreads();
    float* drow = tile + (long)(r - r_first) * geo.tile_cells * CP;
    for (int i = tid; i < ncell * CP; i += NT) {
      const int cell = i / CP, c = i - cell * CP;
      const int llo = Wlo[cell], lhi = Whi[cell];
      float sum = 0.f;
      if (wtab) {
        const float* wt = Wt + cell * maxwin - llo;
        for (int l = llo; l <= lhi; ++l) sum += wt[l] * G[l * CP + c];
      } else {
        for (int l = llo; l <= lhi; ++l) {
          const float l1 = Ll1[l];
          const float wgt = (Li0[l] == cell ? 1.f - l1 : 0.f) + (Li1[l] == cell ? l1 : 0.f);
          sum += wgt * G[l * CP + c];
        }
      }
      drow[i] = sum;                       // [cell][c]: every element of the block's footprint, exactly once
    }
    __syncthreads();
#pragma unroll
    for (int c4 = 0; c4 < CP; c4 += 4) *reinterpret_cast<float4*>(G + tid * CP + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  };

 int kA = 0;                                  // Hh[kA] belongs to row rA, Hh[kA ^ 1] to row rB
  load_row(rA, aA);
  load_row(rB, aB);
#pragma unroll
  for (int c = 0; c < CP; ++c) { gA[c] = 0.f; gB[c] = 0.f; }
  __syncthreads();

  const long long* trow = target + (b * H + ya) * (long)W + (xin ? x : 0);
  long long tnext = *trow;
  float sel_cut = 0.f, sel_gt = 0.f, sel_eq = 0.f;
  if (MODE == 2) { sel_cut = sel[1]; sel_gt = sel[2]; sel_eq = sel[0] != 0.f ? 0.f : sel[3]; }
  float* prow = (MODE != 0) ? pix + (b * H + ya) * (long)W + (xin ? x : 0) : nullptr;
  for (int y = ya; y < yb; ++y) {
    const long long t = tnext;
    if (y + 1 < yb) tnext = trow[(long)(y + 1 - ya) * W];      // next row's target under this row's arithmetic
    const Tap ty = ac_tap(sy, y, h);                           // uniform over the block
    if (ty.i0 != rA) {                                         // row tap advanced (by exactly one: H >= h)
      if (MODE != 1) flush_row(rA, kA, gA, aA);
      kA ^= 1;
      rA = rB;
      rB = rA + (rA < h - 1 ? 1 : 0);
#pragma unroll
      for (int c = 0; c < CP; ++c) { aA[c] = aB[c]; gA[c] = gB[c]; gB[c] = 0.f; }
      load_row(rB, aB);
    }
    float z[CP];
    float m = -TSS_INF, zt = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      z[c] = (c < C) ? ty.l0 * aA[c] + ty.l1 * aB[c] : -TSS_INF;
      m = fmaxf(m, z[c]);
      if (MODE == 1) zt = (t == (long long)c) ? z[c] : zt;    // the target's logit (log2 units): one select per class, this mode only
    }
    const bool valid = xin && t != ignore_index && t >= 0 && t < C;
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      z[c] = __builtin_amdgcn_exp2f(z[c] - m);          // e_c (0 for the padding classes)
      ssum += z[c];
    }
    if (MODE == 1) {      // per-pixel loss in nats, >= 0 (rounding may give -1e-7), 0 for ignored pixels; nothing else in this mode
      const float d = ((m - zt) + __builtin_amdgcn_logf(ssum)) * 0.69314718055994530942f;
      if (xin) prow[(long)(y - ya) * W] = valid ? fmaxf(d, 0.f) : 0.f;
      continue;
    }
    float wpx = 1.f;      // weight of this pixel's loss in the sum (MODE 2: OHEM's selection)
    if (MODE == 2) {
      const float pl = xin ? prow[(long)(y - ya) * W] : 0.f;
      wpx = pl > sel_cut ? sel_gt : (pl == sel_cut ? sel_eq : 0.f);
    }
    if (valid) {
      lsum += m + __builtin_amdgcn_logf(ssum);      // log2
      lcnt += 1.f;
      float* ha = &Hh[kA][tid * CP + (int)t];                 // this lane's own row of the table: no race
      float* hb = &Hh[kA ^ 1][tid * CP + (int)t];
      *ha += ty.l0 * wpx;
      *hb += ty.l1 * wpx;
    }
    const float inv = valid ? __builtin_amdgcn_rcpf(ssum) * wpx : 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float pz = z[c] * inv;
      gA[c] += ty.l0 * pz;
      gB[c] += ty.l1 * pz;
    }
  }
  if (MODE == 1) return;
  flush_row(rA, kA, gA, aA);
  if (rB != rA) flush_row(rB, kA ^ 1, gB, aB);
  else {   // the band ended on the last low-res row (rB == rA): the l1 weights are zero there, nothing to flush
  }
  if (MODE == 2) return;

  double ds = wave_sum((double)lsum * 0.69314718055994530942), dc = wave_sum((double)lcnt);   // back to nats
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { red[0][wave] = ds; red[1][wave] = dc; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, c = 0.0;
    for (int wv = 0; wv < NT / 64; ++wv) { a += red[0][wv]; c += red[1][wv]; }
    lossrows[2 * (long)blockIdx.x] = a;
    lossrows[2 * (long)blockIdx.x + 1] = c;
  }
}

// loss = sum(rows[.][0]) / sum(rows[.][1]) in a fixed order (one block; thread t takes rows t, t + NT, ...)
__global__ __launch_bounds__(NT) void ce_finalize_rows_kernel(const double* rows, int nrows, float* loss, float* inv_count) {
  __shared__ double red[2][NT];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nrows; i += NT) { a += rows[2 * (long)i]; c += rows[2 * (long)i + 1]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = c;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n = red[1][0];
    *loss = (float)(red[0][0] / n);          // n == 0 -> nan, like torch
    *inv_count = n > 0.0 ? (float)(1.0 / n) : 0.f;
  }
}

// first / last low-res row a band's block wrote, first cell / cell count of a strip: the same expressions as in the pass
__device__ __forceinline__ void ce_band_rows(float sy, int band, int band_rows, int H, int h, int* r0, int* r1) {
  const int ya = band * band_rows;
  const int yb = ya + band_rows < H ? ya + band_rows : H;
  *r0 = ac_tap(sy, ya, h).i0;
  *r1 = ac_tap(sy, yb - 1, h).i1;
}
__device__ __forceinline__ void ce_strip_cells(float sx, int strip, int W, int w, int* c0, int* n) {
  const int xl = (strip * NT + NT - 1 < W) ? strip * NT + NT - 1 : W - 1;
  *c0 = ac_tap(sx, strip * NT, w).i0;
  *n = ac_tap(sx, xl, w).i1 - *c0 + 1;
}

// dlow[b][r][cx][:] = (T)(grad_out / count * sum over the tiles that hold cell (r, cx), in (band, strip) order); thread = one
// cell x 8 channels.  Channels >= CP (pitch padding) are written as zeros.
template <typename T, int CP>
__global__ __launch_bounds__(NT) void upsample_ce_gather_kernel(const float* tiles, const float* inv_count, const float* grad_out,
                                                                T* dlow, long ldl, int B, int h, int w, int H, int W,
                                                                const CeGeom geo) {
  const float gs = (*inv_count) * (grad_out ? *grad_out : 1.f);
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const int cv = (int)(ldl / 8);
  const long total = (long)B * h * w * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv) * 8;
    long q = i / cv;
    const int cx = (int)(q % w); q /= w;
    const int r = (int)(q % h);
    const long b = q / h;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    int ylo, yhi, xlo, xhi;
    ac_window(sy, r, H, &ylo, &yhi);
    ac_window(sx, cx, W, &xlo, &xhi);
    for (int band = ylo / geo.band_rows; band <= yhi / geo.band_rows; ++band) {
      int r0, r1;
      ce_band_rows(sy, band, geo.band_rows, H, h, &r0, &r1);
      if (r < r0 || r > r1) continue;
      for (int strip = xlo / NT; strip <= xhi / NT; ++strip) {
        int c0, n;
        ce_strip_cells(sx, strip, W, w, &c0, &n);
        if (cx < c0 || cx >= c0 + n) continue;
        const float* t = tiles + ((((b * geo.nband + band) * geo.nstrip + strip) * geo.tile_rows + (r - r0)) * (long)geo.tile_cells
                                  + (cx - c0)) * CP + c8;
        if (c8 + 4 <= CP) { float u[4]; V4<float>::load(t, u); v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        if (c8 + 8 <= CP) { float u[4]; V4<float>::load(t + 4, u); v[4] += u[0]; v[5] += u[1]; v[6] += u[2]; v[7] += u[3]; }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= gs;
    V8<T>::store(dlow + ((b * h + r) * (long)w + cx) * ldl + c8, v);
  }
}

// argmax over class planes (lowest index wins ties, like torch.argmax) + confusion matrix [C][C] (rows = truth)
template <typename T>
__global__ __launch_bounds__(NT) void argmax_confusion_kernel(const T* logits, const long long* target,
                                                              unsigned char* pred_out, unsigned long long* cm,
                                                              long B, int C, long HW, int ignore_index) {
  extern __shared__ unsigned int scm[];  // [C*C]
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) scm[i] = 0u;
  __syncthreads();
  const long groups = B * (HW / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    for (int c = 0; c < C; ++c) {
      float v[8];
      V8<T>::load(logits + (b * C + c) * HW + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[j] > best[j] || (c == 0)) { best[j] = v[j]; arg[j] = c; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (pred_out) pred_out[b * HW + off + j] = (unsigned char)arg[j];
      if (cm && target) {
        const long long t = target[b * HW + off + j];
        if (t != ignore_index && t >= 0 && t < C) atomicAdd(&scm[(int)t * C + arg[j]], 1u);
      }
    }
  }
  __syncthreads();
  if (cm)
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
      if (scm[i]) atomicAdd(cm + i, (unsigned long long)scm[i]);
}

// Fused evaluation head (eval half): argmax over the classes of the bilinearly upsampled logits
// + confusion-matrix update, straight from the low-res NHWC logits.  Same lane / register layout as
// upsample_ce_onepass_kernel (lane = output column, the two horizontally interpolated low-res rows in registers, one
// FMA per logit); lowest class index wins ties like torch.argmax; rows = truth.
template <typename T, int CP>
__global__ __launch_bounds__(NT, 3) void upsample_argmax_kernel(const T* low, long ldl, const long long* target,
                                                                unsigned char* pred_out, unsigned long long* cm, int B,
                                                                int C, int h, int w, int H, int W, int ignore_index,
                                                                int band_rows) {
  extern __shared__ unsigned int scm[];  // [C*C]
  const int tid = threadIdx.x;
  for (int i = tid; i < C * C; i += NT) scm[i] = 0u;
  __syncthreads();
  const int nstrip = (W + NT - 1) / NT, nband = (H + band_rows - 1) / band_rows;
  int bid = blockIdx.x;
  const int strip = bid % nstrip; bid /= nstrip;
  const int band = bid % nband;
  const long b = bid / nband;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const int x = strip * NT + tid;
  const bool xin = x < W;
  const Tap tx = ac_tap(sx, xin ? x : W - 1, w);
  const int ya = band * band_rows;
  const int yb = ya + band_rows < H ? ya + band_rows : H;
  float aA[CP], aB[CP];
  auto load_row = [&](int r, float* a) {
    const T* p0 = low + ((b * h + r) * (long)w + tx.i0) * ldl;
    const T* p1 = low + ((b * h + r) * (long)w + tx.i1) * ldl;
#pragma unroll
    for (int c4 = 0; c4 < CP; c4 += 4) {
      float u[4], v[4];
      V4<T>::load(p0 + c4, u);
      V4<T>::load(p1 + c4, v);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[c4 + q] = (c4 + q < C) ? tx.l0 * u[q] + tx.l1 * v[q] : -TSS_INF;
    }
  };
  int rA = ac_tap(sy, ya, h).i0;
  int rB = rA + (rA < h - 1 ? 1 : 0);
  load_row(rA, aA);
  load_row(rB, aB);
  for (int y = ya; y < yb; ++y) {
    const Tap ty = ac_tap(sy, y, h);
    if (ty.i0 != rA) {
      rA = rB;
      rB = rA + (rA < h - 1 ? 1 : 0);
#pragma unroll
      for (int c = 0; c < CP; ++c) aA[c] = aB[c];
      load_row(rB, aB);
    }
    float best = -TSS_INF;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float z = ty.l0 * aA[c] + ty.l1 * aB[c];
      if (c < C && (z > best || c == 0)) { best = z; arg = c; }
    }
    if (xin) {
      const long p = (b * H + y) * (long)W + x;
      if (pred_out) pred_out[p] = (unsigned char)arg;
      if (cm && target) {
        const long long t = target[p];
        if (t != ignore_index && t >= 0 && t < C) atomicAdd(&scm[(int)t * C + arg], 1u);
      }
    }
  }
  __syncthreads();
  if (cm)
    for (int i = tid; i < C * C; i += NT)
      if (scm[i]) atomicAdd(cm + i, (unsigned long long)scm[i]);
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

extern "C" {

int tss_cross_entropy_fwd(const void* logits, const long long* target, float* lse, double* acc /*[2], zeroed*/,
                          float* loss, float* inv_count, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(lse), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  {
    tss::ProfScope prof(TSS_K_CE_FWD, (hipStream_t)stream, (double)B * HW * (C * esz(dtype) + 12.0), 0);
    if (dtype == TSS_BF16)
      hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const bf16_t*)logits, target, lse, acc, B, C, HW, ignore_index);
    else
      hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                         (const float*)logits, target, lse, acc, B, C, HW, ignore_index);
  }
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, loss, inv_count);
  return tss::check_last("cross_entropy_fwd");
}

int tss_cross_entropy_bwd(const void* logits, const long long* target, const float* lse, const float* inv_count,
                          const float* grad_out, void* dlogits, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(dlogits) && tss::aligned16(lse), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_CE_BWD, (hipStream_t)stream, (double)B * HW * (2.0 * C * esz(dtype) + 12.0), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)logits, target, lse, inv_count, grad_out, (bf16_t*)dlogits, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)logits, target, lse, inv_count, grad_out, (float*)dlogits, B, C, HW, ignore_index);
  return tss::check_last("cross_entropy_bwd");
}

int tss_argmax_confusion(const void* logits, const long long* target, unsigned char* pred,
                         unsigned long long* confusion /*[C*C] accumulated*/, long B, int C, long HW,
                         int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 64 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  long grid = grid_for(groups);
  if (grid > 1024) grid = 1024;
  tss::ProfScope prof(TSS_K_ARGMAX, (hipStream_t)stream, (double)B * HW * (C * esz(dtype) + 9.0), 0);
  const size_t sh = (size_t)C * C * sizeof(unsigned int);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(argmax_confusion_kernel<bf16_t>, dim3((int)grid), dim3(NT), sh, (hipStream_t)stream,
                       (const bf16_t*)logits, target, pred, confusion, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(argmax_confusion_kernel<float>, dim3((int)grid), dim3(NT), sh, (hipStream_t)stream,
                       (const float*)logits, target, pred, confusion, B, C, HW, ignore_index);
  return tss::check_last("argmax_confusion");
}

int tss_upsample_argmax_confusion(const void* low, long ldl, const long long* target, unsigned char* pred,
                                  unsigned long long* confusion /*[C*C] accumulated*/, int B, int C, int h, int w,
                                  int H, int W, int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 24 && (ldl % 8) == 0 && ldl >= (C + 3) / 4 * 4 && h > 0 && w > 0 && H >= h && W >= w, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(low), TSS_ERR_ALIGN);
  if ((long)B * H * W == 0) return TSS_OK;
  const int nstrip = (W + NT - 1) / NT;
  int band_rows = 64;
  while (band_rows > 16 && (long)B * nstrip * ((H + band_rows - 1) / band_rows) < 2048) band_rows /= 2;
  const long grid = (long)B * nstrip * ((H + band_rows - 1) / band_rows);
  tss::ProfScope prof(TSS_K_ARGMAX, (hipStream_t)stream, (double)B * h * w * C * esz(dtype) + (double)B * H * W * 9.0, 0);
  const size_t sh = (size_t)C * C * sizeof(unsigned int);
#define TSS_AM_LAUNCH(TT, CPV)                                                                                 \
  hipLaunchKernelGGL((upsample_argmax_kernel<TT, CPV>), dim3((int)grid), dim3(NT), sh, (hipStream_t)stream,      \
                     (const TT*)low, ldl, target, pred, confusion, B, C, h, w, H, W, ignore_index, band_rows)
  if (dtype == TSS_BF16) { if (C <= 20) TSS_AM_LAUNCH(bf16_t, 20); else TSS_AM_LAUNCH(bf16_t, 24); }
  else { if (C <= 20) TSS_AM_LAUNCH(float, 20); else TSS_AM_LAUNCH(float, 24); }
#undef TSS_AM_LAUNCH
  return tss::check_last("upsample_argmax_confusion");
}

// Geometry of the pass (the same on the host, in the pass and in the gather): bands / strips and the tile every block owns.
static CeGeom ce_geom(int B, int h, int w, int H, int W) {
  CeGeom g;
  g.nstrip = (W + NT - 1) / NT;
  // bands: enough blocks to fill the chip, but every band pays two extra row flushes
  g.band_rows = 64;
  while (g.band_rows > 16 && (long)B * g.nstrip * ((H + g.band_rows - 1) / g.band_rows) < 2048) g.band_rows /= 2;
  g.nband = (H + g.band_rows - 1) / g.band_rows;
  // conservative tile extent: a band of R output rows touches at most floor((R - 1) * (h - 1) / (H - 1)) + 3 low-res rows
  // (the pass and the gather index the tile by the rows / cells actually touched, which they both compute exactly)
  const double sy = H > 1 ? (double)(h - 1) / (double)(H - 1) : 0.0, sx = W > 1 ? (double)(w - 1) / (double)(W - 1) : 0.0;
  g.tile_rows = (int)((g.band_rows - 1) * sy) + 3;
  g.tile_cells = (int)((NT - 1) * sx) + 3;
  if (g.tile_rows > h) g.tile_rows = h;
  if (g.tile_cells > w) g.tile_cells = w;
  return g;
}

long tss_upsample_ce_ws(int B, int C, int h, int w, int H, int W) {
  if (B <= 0 || C <= 0 || C > 24 || h <= 0 || w <= 0 || H < h || W < w) return 0;
  const CeGeom g = ce_geom(B, h, w, H, W);
  const long nblocks = (long)B * g.nband * g.nstrip;
  const int CP = C <= 20 ? 20 : 24;
  return nblocks * 4 /* 2 doubles */ + nblocks * g.tile_rows * g.tile_cells * CP;
}

int tss_upsample_ce_fwd(const void* low, long ldl, const long long* target, float* ws /* tss_upsample_ce_ws floats, not initialised */,
                        float* loss, float* inv_count,
                        int B, int C, int h, int w, int H, int W, int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 24 && (ldl % 8) == 0 && ldl >= (C + 3) / 4 * 4 && h > 0 && w > 0 && H >= h && W >= w, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(low) && tss::aligned16(ws), TSS_ERR_ALIGN);
  if ((long)B * H * W == 0) return TSS_OK;
  const CeGeom geo = ce_geom(B, h, w, H, W);
  const long grid = (long)B * geo.nstrip * geo.nband;
  double* lossrows = reinterpret_cast<double*>(ws);
  float* tiles = ws + grid * 4;
  {
    tss::ProfScope prof(TSS_K_UPSAMPLE_CE_FWD, (hipStream_t)stream,
                        (double)B * h * w * C * (esz(dtype) + 8.0) + (double)B * H * W * 8.0, 0);
#define TSS_CE_LAUNCH(TT, CPV)                                                                                   \
    hipLaunchKernelGGL((upsample_ce_onepass_kernel<TT, CPV, 0>), dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, \
                       (const TT*)low, ldl, target, tiles, lossrows, B, C, h, w, H, W, ignore_index, geo, nullptr, nullptr)
    if (dtype == TSS_BF16) { if (C <= 20) TSS_CE_LAUNCH(bf16_t, 20); else TSS_CE_LAUNCH(bf16_t, 24); }
    else { if (C <= 20) TSS_CE_LAUNCH(float, 20); else TSS_CE_LAUNCH(float, 24); }
#undef TSS_CE_LAUNCH
  }
  hipLaunchKernelGGL(ce_finalize_rows_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, lossrows, (int)grid, loss, inv_count);
  return tss::check_last("upsample_ce_fwd");
}

// Per-pixel cross-entropy of the upsampled logits ([B][H][W] f32, 0 for ignored pixels), straight from the low-res logits.
int tss_upsample_pixel_ce(const void* low, long ldl, const long long* target, float* pix, int B, int C, int h, int w, int H, int W,
                          int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 24 && (ldl % 8) == 0 && ldl >= (C + 3) / 4 * 4 && h > 0 && w > 0 && H >= h && W >= w && pix, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(low), TSS_ERR_ALIGN);
  if ((long)B * H * W == 0) return TSS_OK;
  const CeGeom geo = ce_geom(B, h, w, H, W);
  const long grid = (long)B * geo.nstrip * geo.nband;
  tss::ProfScope prof(TSS_K_UPSAMPLE_CE_FWD, (hipStream_t)stream, (double)B * h * w * C * esz(dtype) + (double)B * H * W * 12.0, 0);
#define TSS_CE_LAUNCH(TT, CPV)                                                                                       \
  hipLaunchKernelGGL((upsample_ce_onepass_kernel<TT, CPV, 1>), dim3((int)grid), dim3(NT), 0, (hipStream_t)stream,     \
                     (const TT*)low, ldl, target, nullptr, nullptr, B, C, h, w, H, W, ignore_index, geo, pix, nullptr)
  if (dtype == TSS_BF16) { if (C <= 20) TSS_CE_LAUNCH(bf16_t, 20); else TSS_CE_LAUNCH(bf16_t, 24); }
  else { if (C <= 20) TSS_CE_LAUNCH(float, 20); else TSS_CE_LAUNCH(float, 24); }
#undef TSS_CE_LAUNCH
  return tss::check_last("upsample_pixel_ce");
}

// Gradient tiles of sum_p weight(pix[p]) * CE_p with weight = sel[2] for pix > sel[1], sel[3] for pix == sel[1] (top-n mode only),
// else 0 (sel = the params of tss_ohem_select).  ws as for tss_upsample_ce_fwd; tss_upsample_ce_bwd (inv_count -> 1.0) gathers.
int tss_upsample_ohem_grad(const void* low, long ldl, const long long* target, const float* pix, const float* sel, float* ws,
                           int B, int C, int h, int w, int H, int W, int ignore_index, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 24 && (ldl % 8) == 0 && ldl >= (C + 3) / 4 * 4 && h > 0 && w > 0 && H >= h && W >= w && pix && sel && ws,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(low) && tss::aligned16(ws), TSS_ERR_ALIGN);
  if ((long)B * H * W == 0) return TSS_OK;
  const CeGeom geo = ce_geom(B, h, w, H, W);
  const long grid = (long)B * geo.nstrip * geo.nband;
  float* tiles = ws + grid * 4;
  tss::ProfScope prof(TSS_K_UPSAMPLE_CE_FWD, (hipStream_t)stream, (double)B * h * w * C * (esz(dtype) + 8.0) + (double)B * H * W * 12.0, 0);
#define TSS_CE_LAUNCH(TT, CPV)                                                                                       \
  hipLaunchKernelGGL((upsample_ce_onepass_kernel<TT, CPV, 2>), dim3((int)grid), dim3(NT), 0, (hipStream_t)stream,     \
                     (const TT*)low, ldl, target, tiles, nullptr, B, C, h, w, H, W, ignore_index, geo, const_cast<float*>(pix), sel)
  if (dtype == TSS_BF16) { if (C <= 20) TSS_CE_LAUNCH(bf16_t, 20); else TSS_CE_LAUNCH(bf16_t, 24); }
  else { if (C <= 20) TSS_CE_LAUNCH(float, 20); else TSS_CE_LAUNCH(float, 24); }
#undef TSS_CE_LAUNCH
  return tss::check_last("upsample_ohem_grad");
}

int tss_upsample_ce_bwd(const float* ws, const float* inv_count, const float* grad_out, void* dlow, long ldl,
                        int B, int C, int h, int w, int H, int W, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && C <= 24 && (ldl % 8) == 0 && ldl >= (C + 3) / 4 * 4 && h > 0 && w > 0 && H >= h && W >= w, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(ws) && tss::aligned16(dlow), TSS_ERR_ALIGN);
  const long n = (long)B * h * w * ldl;
  if (n == 0) return TSS_OK;
  const CeGeom geo = ce_geom(B, h, w, H, W);
  const float* tiles = ws + (long)B * geo.nstrip * geo.nband * 4;
  tss::ProfScope prof(TSS_K_UPSAMPLE_CE_BWD, (hipStream_t)stream, (double)n * (4.0 + esz(dtype)), 0);
#define TSS_CE_GATHER(TT, CPV)                                                                                      \
  hipLaunchKernelGGL((upsample_ce_gather_kernel<TT, CPV>), dim3(grid_for(n / 8)), dim3(NT), 0, (hipStream_t)stream, \
                     tiles, inv_count, grad_out, (TT*)dlow, ldl, B, h, w, H, W, geo)
  if (dtype == TSS_BF16) { if (C <= 20) TSS_CE_GATHER(bf16_t, 20); else TSS_CE_GATHER(bf16_t, 24); }
  else { if (C <= 20) TSS_CE_GATHER(float, 20); else TSS_CE_GATHER(float, 24); }
#undef TSS_CE_GATHER
  return tss::check_last("upsample_ce_bwd");
}

}  // extern "C"

// Depthwise 3x3 convolution (stride 1/2, dilation 1/4, padding = dilation), NHWC, HBM-bound.
//
// One lane owns 8 consecutive channels (one 16-byte bf16 vector) of one pixel; consecutive lanes own
// consecutive channel vectors, so every load/store instruction of a wave covers whole contiguous
// NHWC rows.  Blocks are persistent and sweep the image in XCD bands (common.h: xcd_tiles) so the
// three input rows a tile shares with its vertical neighbours are served by the same L2.
//   forward        y = dw(relu?((x-mean)*scale+bias))              + per-channel sum / sum-of-squares of y
//   backward-data  e_in = relu'(.) * dw^T(ga*(e-ce)+gb*(y-mu))      + per-channel sum(e_in), sum(e_in*(x-mean))
//   backward-weight dW[c][tap] += sum_p g(p,c) * a(p+tap, c)
// The producer's BatchNorm(+ReLU) is applied on load and this conv's BatchNorm backward is applied on
// load of (e, y): neither normalised activations nor input gradients of BN ever touch HBM.
#include <type_traits>
#include "common.h"
#include "dwroll.h"

namespace {

constexpr int NT_MAX = 256;

#ifdef TSS_TIMING
__device__ unsigned long long g_dw_timing[8];   // debug builds: [prologue, loop, tail, -, -, -, -, blocks] cycles of wave 0
#define TSS_T(var) unsigned long long var; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define TSS_T(var)
#endif

struct DwArgs {
  const void* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu;
  int x_mask;      // backward-data: x is the producer's raw output -> ReLU mask + statistics in the epilogue
  const float* w;  // [C][9]
  void* y; long ldy; double* stats;
  const void* e; long lde; const void* yraw; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  float* dw; float* ws;
  int B, Hin, Win, C, stride, dil, Hout, Wout;
  int CV, NPL;
  // backward-data only: the first `lead` blocks (a multiple of 8, `nred` of them working) sum the workspace rows of the
  // same layer's weight gradient (dw_reduce_block)
  const float* red_ws; float* red_dw; int red_rows, nred, lead;
};

template <typename T> struct StatAcc { typedef float type; };
template <> struct StatAcc<float> { typedef double type; };

// Block-level per-channel reduction of two 8-channel partials per thread; the block's partial sums go to its own
// slab row (plain stores), rows no block owns are zeroed here so the caller never has to clear the buffer.
template <typename A>
__device__ __forceinline__ void flush_stats(const A s1[8], const A s2[8], double* stats, int C, int CV, int NPL,
                                            int cg, int pl, bool active, unsigned char* smem, int lead = 0) {
  const int grid = (int)gridDim.x - lead, bid = (int)blockIdx.x - lead;   // slab row = index among the sweeping blocks
  A* red = reinterpret_cast<A*>(smem);  // [NPL][2][C]
  __syncthreads();
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(pl * 2 + 0) * C + cg * 8 + j] = s1[j];
      red[(pl * 2 + 1) * C + cg * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    const int which = i / C, c = i - which * C;
    double a = 0.0;
    for (int q = 0; q < NPL; ++q) a += (double)red[(q * 2 + which) * C + c];
    stats[(long)bid * 2 * C + i] = a;
    for (int r = bid + grid; r < TSS_STAT_SLABS; r += grid) stats[(long)r * 2 * C + i] = 0.0;
  }
}

// [C][9] f32 weights -> LDS [tap][half][C/8][4]: the 8 weights of a lane's channel vector are two float4, the low halves of
// all channel vectors contiguous, then the high halves -- consecutive lanes read consecutive 16-byte words (conflict-free
// ds_read_b128; with [tap][C] the two float4 of a lane sat 32 bytes from its neighbour's: 2-way conflicts on every read, and
// the transposing store hit one bank 9 times over: SQ_LDS_BANK_CONFLICT = 13 M cycles per launch).  The loop runs over
// the LDS destination (conflict-free stores, gathered L2 reads); all loads of a lane are issued before the first store.
__device__ __forceinline__ void stage_weights(float* wl, const float* w, int C, int tid, int nthreads) {
  const int n = C * 9, hc = C >> 1;
  constexpr int U = 14;   // C <= 768 with >= 192 threads: at most three trips (one for C <= 384)
  for (int base = 0; base < n; base += nthreads * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = base + tid + u * nthreads;
      const int dd = d < n ? d : 0;
      const int t = dd / C, r = dd - t * C;
      const int half = r >= hc ? 1 : 0, q = r - half * hc;
      const int c = (q >> 2) * 8 + half * 4 + (q & 3);
      v[u] = w[c * 9 + t];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = base + tid + u * nthreads;
      if (d < n) wl[d] = v[u];
    }
  }
}
// the two float4 of channel vector cg (= c0 / 8) for tap t
#define TSS_DW_W0(wl, t, C, c0) (*reinterpret_cast<const float4*>((wl) + (t) * (C) + ((c0) >> 1)))
#define TSS_DW_W1(wl, t, C, c0) (*reinterpret_cast<const float4*>((wl) + (t) * (C) + ((C) >> 1) + ((c0) >> 1)))

// 8 per-channel constants of a lane: two unconditional 16-byte loads through a null-safe pointer, then a select
// (a `ptr ? ptr[c] : dflt` per element compiles to a branch and a wait per load: a serial chain of L2 round trips at
// the top of every block)
__device__ __forceinline__ void coef8(const float* p, const float* safe, int c0, bool active, float dflt, float out[8]) {
  const bool has = p != nullptr;
  const float* q = has ? p + (active ? c0 : 0) : safe;
  float v[8];
  V4<float>::load(q, v);
  V4<float>::load(q + 4, v + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (has && active) ? v[j] : dflt;
}

// Tap geometry of one pixel, branch-free: clamped source coordinates + validity, so that all nine loads of a
// window can be issued back to back (no divergent skip between them) and invalid taps are zeroed by a select.
struct Taps { long off[9]; bool ok[9]; };

template <typename T>
__global__ __launch_bounds__(NT_MAX) void dw_fwd_kernel(const DwArgs g) {
  typedef typename StatAcc<T>::type A;
  __shared__ __align__(16) unsigned char smem[NT_MAX * 16 * sizeof(double)];
  __shared__ __align__(16) float wl[9 * 768];  // [tap][C]
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* x = reinterpret_cast<const T*>(g.x);
  T* y = reinterpret_cast<T*>(g.y);
  stage_weights(wl, g.w, g.C, tid, blockDim.x);

  float mu[8], sc[8], sh[8];
  A s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; mu[j] = 0.f; sc[j] = 1.f; sh[j] = 0.f; }
  if (active && g.xs) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = g.xs[c0 + j]; mu[j] = g.xm ? g.xm[c0 + j] : 0.f; sh[j] = g.xb ? g.xb[c0 + j] : 0.f; }
  }
  __syncthreads();
  const long P = (long)g.B * g.Hout * g.Wout;
  const long ntiles = (P + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long p = (long)tile * g.NPL + pl;
    if (!active || p >= P) continue;
    const int ox = (int)(p % g.Wout);
    const long t2 = p / g.Wout;
    const int oy = (int)(t2 % g.Hout);
    const long b = t2 / g.Hout;
    typename V8<T>::Raw raw[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * g.stride + (ky - 1) * g.dil;
      const bool vy = iy >= 0 && iy < g.Hin;
      const int iyc = vy ? iy : 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * g.stride + (kx - 1) * g.dil;
        const bool vx = ix >= 0 && ix < g.Win;
        const int ixc = vx ? ix : 0;
        ok[ky * 3 + kx] = vy && vx;
        raw[ky * 3 + kx] = V8<T>::load_raw(x + ((b * g.Hin + iyc) * (long)g.Win + ixc) * g.ldx + c0);
      }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v[8];
      V8<T>::unpack(raw[t], v);
      const float4 w0 = TSS_DW_W0(wl, t, g.C, c0);
      const float4 w1 = TSS_DW_W1(wl, t, g.C, c0);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = (v[j] - mu[j]) * sc[j] + sh[j];
        if (g.x_relu) a = a > 0.f ? a : 0.f;
        a = ok[t] ? a : 0.f;
        acc[j] += a * wv[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] = V8<T>::round(acc[j]);
      s1[j] += (A)acc[j];
      s2[j] += (A)acc[j] * (A)acc[j];
    }
    V8<T>::store(y + p * g.ldy + c0, acc);
  }
  if (g.stats) flush_stats<A>(s1, s2, g.stats, g.C, g.CV, g.NPL, cg, pl, active, smem);
}

template <typename T>
__global__ __launch_bounds__(NT_MAX) void dw_bwd_data_kernel(const DwArgs g) {
  typedef typename StatAcc<T>::type A;
  __shared__ __align__(16) unsigned char smem[NT_MAX * 16 * sizeof(double)];
  __shared__ __align__(16) float wl[9 * 768];  // [tap][C]
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const T* x = reinterpret_cast<const T*>(g.x);
  T* out = reinterpret_cast<T*>(g.y);
  stage_weights(wl, g.w, g.C, tid, blockDim.x);

  float ca[8], cb[8], ce[8], cm[8], mu[8], sc[8], sh[8];
  A s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = 0; s2[j] = 0; ca[j] = 1.f; cb[j] = 0.f; ce[j] = 0.f; cm[j] = 0.f; mu[j] = 0.f; sc[j] = 1.f; sh[j] = 0.f;
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (g.ga) ca[j] = g.ga[c0 + j];
      if (g.yraw) { cb[j] = g.gb[c0 + j]; ce[j] = g.gce[c0 + j]; cm[j] = g.gmu[c0 + j]; }
      if (g.xm) mu[j] = g.xm[c0 + j];
      if (g.xs) { sc[j] = g.xs[c0 + j]; sh[j] = g.xb ? g.xb[c0 + j] : 0.f; }
    }
  }
  __syncthreads();
  const long P = (long)g.B * g.Hin * g.Win;  // one item per INPUT pixel
  const long ntiles = (P + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long p = (long)tile * g.NPL + pl;
    if (!active || p >= P) continue;
    const int ix = (int)(p % g.Win);
    const long t2 = p / g.Win;
    const int iy = (int)(t2 % g.Hin);
    const long b = t2 / g.Hin;
    typename V8<T>::Raw re[9], ry[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ny = iy - (ky - 1) * g.dil;  // = oy * stride
      const int oy = ny / g.stride;
      const bool vy = ny >= 0 && oy * g.stride == ny && oy < g.Hout;
      const int oyc = vy ? oy : 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int nx = ix - (kx - 1) * g.dil;
        const int ox = nx / g.stride;
        const bool vx = nx >= 0 && ox * g.stride == nx && ox < g.Wout;
        const int oxc = vx ? ox : 0;
        ok[ky * 3 + kx] = vy && vx;
        const long q = (b * g.Hout + oyc) * (long)g.Wout + oxc;
        re[ky * 3 + kx] = V8<T>::load_raw(e + q * g.lde + c0);
        if (yr) ry[ky * 3 + kx] = V8<T>::load_raw(yr + q * g.ldyr + c0);
      }
    }
    typename V8<T>::Raw rx;
    if (g.x) rx = V8<T>::load_raw(x + p * g.ldx + c0);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float ev[8];
      V8<T>::unpack(re[t], ev);
      const float4 w0 = TSS_DW_W0(wl, t, g.C, c0);
      const float4 w1 = TSS_DW_W1(wl, t, g.C, c0);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      if (yr) {
        float yv[8];
        V8<T>::unpack(ry[t], yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gv = ca[j] * (ev[j] - ce[j]) + cb[j] * (yv[j] - cm[j]);
          acc[j] += (ok[t] ? gv : 0.f) * wv[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (ok[t] ? ca[j] * ev[j] : 0.f) * wv[j];
      }
    }
    if (g.x) {
      float xv[8];
      V8<T>::unpack(rx, xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xc = xv[j] - mu[j];
        if (g.x_relu && !(xc * sc[j] + sh[j] > 0.f)) acc[j] = 0.f;
        acc[j] = V8<T>::round(acc[j]);
        s1[j] += (A)acc[j];
        s2[j] += (A)acc[j] * (A)xc;
      }
    }
    V8<T>::store(out + p * g.ldy + c0, acc);
  }
  if (g.stats) flush_stats<A>(s1, s2, g.stats, g.C, g.CV, g.NPL, cg, pl, active, smem);
}

template <typename T>
__global__ __launch_bounds__(NT_MAX) void dw_bwd_weight_kernel(const DwArgs g) {
  __shared__ float sdw[768 * 9];  // C <= 768 on the hot path (checked by the host wrapper)
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const T* x = reinterpret_cast<const T*>(g.x);

  float acc[9][8], ca[8], cb[8], ce[8], cm[8], mu[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ca[j] = 1.f; cb[j] = 0.f; ce[j] = 0.f; cm[j] = 0.f; mu[j] = 0.f; sc[j] = 1.f; sh[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][j] = 0.f;
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (g.ga) ca[j] = g.ga[c0 + j];
      if (g.yraw) { cb[j] = g.gb[c0 + j]; ce[j] = g.gce[c0 + j]; cm[j] = g.gmu[c0 + j]; }
      if (g.xs) { sc[j] = g.xs[c0 + j]; mu[j] = g.xm ? g.xm[c0 + j] : 0.f; sh[j] = g.xb ? g.xb[c0 + j] : 0.f; }
    }
  }
  const long P = (long)g.B * g.Hout * g.Wout;
  const long ntiles = (P + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long p = (long)tile * g.NPL + pl;
    if (!active || p >= P) continue;
    const int ox = (int)(p % g.Wout);
    const long t2 = p / g.Wout;
    const int oy = (int)(t2 % g.Hout);
    const long b = t2 / g.Hout;
    typename V8<T>::Raw raw[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * g.stride + (ky - 1) * g.dil;
      const bool vy = iy >= 0 && iy < g.Hin;
      const int iyc = vy ? iy : 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * g.stride + (kx - 1) * g.dil;
        const bool vx = ix >= 0 && ix < g.Win;
        const int ixc = vx ? ix : 0;
        ok[ky * 3 + kx] = vy && vx;
        raw[ky * 3 + kx] = V8<T>::load_raw(x + ((b * g.Hin + iyc) * (long)g.Win + ixc) * g.ldx + c0);
      }
    }
    float gv[8];
    V8<T>::load(e + p * g.lde + c0, gv);
    if (yr) {
      float yv[8];
      V8<T>::load(yr + p * g.ldyr + c0, yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = ca[j] * (gv[j] - ce[j]) + cb[j] * (yv[j] - cm[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = ca[j] * gv[j];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v[8];
      V8<T>::unpack(raw[t], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = (v[j] - mu[j]) * sc[j] + sh[j];
        if (g.x_relu) a = a > 0.f ? a : 0.f;
        acc[t][j] += gv[j] * (ok[t] ? a : 0.f);
      }
    }
  }
  float* red = sdw;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[t][j];
    }
    __syncthreads();
    for (int c = tid; c < g.C; c += blockDim.x) {
      const int cgc = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int q = 0; q < g.NPL; ++q) sum += red[(q * g.CV + cgc) * 8 + j];
      g.ws[(long)blockIdx.x * g.C * 9 + c * 9 + t] = sum;
    }
  }
}

// dw[i] += sum over the workspace rows.  64 columns per block with the lanes along the columns (every load
// instruction reads one contiguous 256-byte piece of a row), the block's waves taking rows w, w+nw, ... with up to 32
// loads in flight per lane, then an LDS reduction over the waves.  Runs as its own kernel (16 waves) or as extra blocks
// behind the same layer's backward-data grid (3-4 waves: one launch less, hidden behind the streaming blocks).
__device__ __forceinline__ void dw_reduce_block(const float* ws, float* dw, int n, int rows, int bid, float* part /*[nw][64]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = bid * 64 + lane;
  const int cc = col < n ? col : 0;
  float s = 0.f;
  for (int r0 = wave; r0 < rows; r0 += nw * 32) {
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int r = r0 + nw * u; v[u] = ws[(long)(r < rows ? r : 0) * n + cc]; }
#pragma unroll
    for (int u = 0; u < 32; ++u) s += (r0 + nw * u < rows) ? v[u] : 0.f;
  }
  part[wave * 64 + lane] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < n) {
    float t = 0.f;
    for (int q = 0; q < nw; ++q) t += part[q * 64 + threadIdx.x];
    dw[col] += t;
  }
}

// The same with 16-byte loads (n % 4 == 0): a lane owns 4 consecutive columns, a block 256.  Round 4: the one-sweep backward of the
// large 1x1 layers (csrc/pwsweep.hip) leaves 256 rows of 16 - 24 k floats per layer -- 230 MB per FastSCNN step through this
// kernel; with 4-byte loads it ran at 3.2 TB/s (72 us per step).
__device__ __forceinline__ void dw_reduce_block4(const float* ws, float* dw, int n, int rows, int bid, float4* part /*[nw][64]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = (bid * 64 + lane) * 4;
  const int cc = col < n ? col : 0;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = wave; r0 < rows; r0 += nw * 16) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int r = r0 + nw * u; v[u] = *reinterpret_cast<const float4*>(ws + (long)(r < rows ? r : 0) * n + cc); }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (r0 + nw * u < rows) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  part[wave * 64 + lane] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < n) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < nw; ++q) { const float4 p = part[q * 64 + threadIdx.x]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
    float4 d = *reinterpret_cast<float4*>(dw + col);
    d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
    *reinterpret_cast<float4*>(dw + col) = d;
  }
}

constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(RED_WAVES * 64) void dw_reduce_kernel(const float* ws, float* dw, int n, int rows) {
  __shared__ float part[RED_WAVES * 64];
  dw_reduce_block(ws, dw, n, rows, blockIdx.x, part);
}

// the row reductions of several layers in ONE launch (blockIdx.y = layer): the one-sweep backward of a depthwise layer leaves
// its per-block rows in a workspace, and a training step has 13-17 such layers whose 5 us reductions nobody waits for
// before the optimizer -- they are collected and summed together at the end of the backward pass
constexpr int RED_MANY = 40;
struct ReduceJobs { const float* ws[RED_MANY]; float* dw[RED_MANY]; int n[RED_MANY]; int rows[RED_MANY]; };
__device__ __forceinline__ bool red_vec4(const float* ws, const float* dw, int n) {
  return (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(dw)) & 15u) == 0;
}
__global__ __launch_bounds__(RED_WAVES * 64) void dw_reduce_many_kernel(const ReduceJobs j) {
  __shared__ float4 part[RED_WAVES * 64];
  const int job = blockIdx.y;
  const int n = j.n[job];
  if (red_vec4(j.ws[job], j.dw[job], n)) {
    if ((int)blockIdx.x * 256 >= n) return;
    dw_reduce_block4(j.ws[job], j.dw[job], n, j.rows[job], blockIdx.x, part);
  } else {
    if ((int)blockIdx.x * 64 >= n) return;
    dw_reduce_block(j.ws[job], j.dw[job], n, j.rows[job], blockIdx.x, reinterpret_cast<float*>(part));
  }
}

// ------------------------------------------------------------------------------------------------------------
// Strip kernels: one lane owns 8 channels of FOUR pixels of a row (consecutive for stride 2, D apart for stride 1).  The
// 3 x NCOL input window of a strip is loaded once (NCOL = 6 for stride 1 at any dilation, 9 for stride 2)
// and every loaded vector is normalised once, instead of 9 loads + 9 normalisations per pixel: 2-2.4x fewer load
// instructions and 6-12 independent 16-byte loads in flight per lane.  (S, D) are template parameters so the
// window -> (pixel, tap) scatter is resolved at compile time.
constexpr int SW = 4;

template <typename T, int S, int D>
__global__ __launch_bounds__(NT_MAX, ((D == 1 && sizeof(T) == 2) ? 3 : 2)) void dw_fwd_strip_kernel(const DwArgs g) {
  typedef typename StatAcc<T>::type A;
  // stride 1: a strip is 4 pixels D apart (x0, x0+D, ...): the taps of neighbouring strip pixels coincide, so the window
  // is 6 columns for any dilation (with consecutive pixels a dilation-4 strip needs 12 columns and shares nothing)
  constexpr int XS = (S == 1) ? D : 1;                               // pixel step inside a strip
  constexpr int NCOL = (S == 1) ? (SW + 2) : ((SW - 1) * S + 2 * D + 1);
  __shared__ __align__(16) unsigned char smem[NT_MAX * 16 * sizeof(typename StatAcc<T>::type)];
  __shared__ __align__(16) float wl[9 * 768];  // [tap][C]
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* x = reinterpret_cast<const T*>(g.x);
  T* y = reinterpret_cast<T*>(g.y);
  stage_weights(wl, g.w, g.C, tid, blockDim.x);

  float mu[8], sc[8], sh[8];
  A s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
  coef8(g.xs, g.w, c0, active, 1.f, sc);
  coef8(g.xs ? g.xm : nullptr, g.w, c0, active, 0.f, mu);
  coef8(g.xs ? g.xb : nullptr, g.w, c0, active, 0.f, sh);
  // bf16 activations: (x-mu)*s+b is evaluated as x*s + (b-mu*s) (one FMA; the rounding difference is far below bf16)
  constexpr bool FOLD = sizeof(T) == 2;
  if (FOLD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = __builtin_fmaf(-mu[j], sc[j], sh[j]);
  }
  __syncthreads();
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  const int nstrip = ((g.Wout + SW * XS - 1) / (SW * XS)) * XS;   // groups of SW*XS columns, XS interleaved strips each
  const long U = (long)g.B * g.Hout * nstrip;
  const long ntiles = (U + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long u = (long)tile * g.NPL + pl;
    if (!active || u >= U) continue;
    const int xs = (int)(u % nstrip);
    const long t2 = u / nstrip;
    const int oy = (int)(t2 % g.Hout);
    const long b = t2 / g.Hout;
    const int x0 = (xs / XS) * (SW * XS) + (xs % XS);   // consecutive units -> consecutive pixels (coalesced lanes)
    float acc[SW][8];
#pragma unroll
    for (int i = 0; i < SW; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * S + (ky - 1) * D;
      const bool vy = iy >= 0 && iy < g.Hin;
      const T* row = x + ((b * g.Hin + (vy ? iy : 0)) * (long)g.Win) * g.ldx + c0;
      typename V8<T>::Raw raw[NCOL];
      bool ok[NCOL];
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        const int ix = (S == 1) ? x0 + (c - 1) * XS : x0 * S - D + c;
        ok[c] = vy && ix >= 0 && ix < g.Win;
        raw[c] = V8<T>::load_raw(row + (long)(ix < 0 ? 0 : (ix >= g.Win ? g.Win - 1 : ix)) * g.ldx);
      }
      float wv[3][8];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w0 = TSS_DW_W0(wl, ky * 3 + kx, g.C, c0);
        const float4 w1 = TSS_DW_W1(wl, ky * 3 + kx, g.C, c0);
        wv[kx][0] = w0.x; wv[kx][1] = w0.y; wv[kx][2] = w0.z; wv[kx][3] = w0.w;
        wv[kx][4] = w1.x; wv[kx][5] = w1.y; wv[kx][6] = w1.z; wv[kx][7] = w1.w;
      }
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        float v[8];
        V8<T>::unpack(raw[c], v);
        const float lo = ok[c] ? relu_lo : 0.f, hi = ok[c] ? TSS_INF : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = clamp3(FOLD ? v[j] * sc[j] + sh[j] : (v[j] - mu[j]) * sc[j] + sh[j], lo, hi);
#pragma unroll
        for (int i = 0; i < SW; ++i)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((S == 1) ? (i + kx == c) : (i * S + kx * D == c)) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[i][j] += v[j] * wv[kx][j];
            }
      }
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) {
      if (x0 + i * XS < g.Wout) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[i][j] = V8<T>::round(acc[i][j]);
          s1[j] += (A)acc[i][j];
          s2[j] += (A)acc[i][j] * (A)acc[i][j];
        }
        V8<T>::store(y + ((b * g.Hout + oy) * (long)g.Wout + x0 + i * XS) * g.ldy + c0, acc[i]);
      }
    }
  }
  if (g.stats) flush_stats<A>(s1, s2, g.stats, g.C, g.CV, g.NPL, cg, pl, active, smem);
}

template <typename T, int S, int D>
__global__ __launch_bounds__(NT_MAX, (D == 1 ? 2 : 1)) void dw_bwd_weight_strip_kernel(const DwArgs g) {
  // stride 1: a strip is 4 pixels D apart (x0, x0+D, ...): the taps of neighbouring strip pixels coincide, so the window
  // is 6 columns for any dilation (with consecutive pixels a dilation-4 strip needs 12 columns and shares nothing)
  constexpr int XS = (S == 1) ? D : 1;                               // pixel step inside a strip
  constexpr int NCOL = (S == 1) ? (SW + 2) : ((SW - 1) * S + 2 * D + 1);
  __shared__ __align__(16) float sdw[768 * 9];
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const T* x = reinterpret_cast<const T*>(g.x);

  float accw[9][8], ca[8], cb[8], ce[8], cm[8], mu[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int t = 0; t < 9; ++t) accw[t][j] = 0.f;
  }
  const float* safe = reinterpret_cast<const float*>(g.ws);   // any readable f32 address
  coef8(g.ga, safe, c0, active, 1.f, ca);
  coef8(g.yraw ? g.gb : nullptr, safe, c0, active, 0.f, cb);
  coef8(g.yraw ? g.gce : nullptr, safe, c0, active, 0.f, ce);
  coef8(g.yraw ? g.gmu : nullptr, safe, c0, active, 0.f, cm);
  coef8(g.xs, safe, c0, active, 1.f, sc);
  coef8(g.xs ? g.xm : nullptr, safe, c0, active, 0.f, mu);
  coef8(g.xs ? g.xb : nullptr, safe, c0, active, 0.f, sh);
  constexpr bool FOLD = sizeof(T) == 2;
  float kd[8];   // folded backward constant: g = ga*e + gb*y + kd
#pragma unroll
  for (int j = 0; j < 8; ++j) { kd[j] = -(ca[j] * ce[j]) - cb[j] * cm[j]; if (FOLD) sh[j] -= mu[j] * sc[j]; }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  const int nstrip = ((g.Wout + SW * XS - 1) / (SW * XS)) * XS;   // groups of SW*XS columns, XS interleaved strips each
  const long U = (long)g.B * g.Hout * nstrip;
  const long ntiles = (U + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long u = (long)tile * g.NPL + pl;
    if (!active || u >= U) continue;
    const int xs = (int)(u % nstrip);
    const long t2 = u / nstrip;
    const int oy = (int)(t2 % g.Hout);
    const long b = t2 / g.Hout;
    const int x0 = (xs / XS) * (SW * XS) + (xs % XS);   // consecutive units -> consecutive pixels (coalesced lanes)
    float gv[SW][8];
#pragma unroll
    for (int i = 0; i < SW; ++i) {
      const bool in = x0 + i * XS < g.Wout;
      const long q = (b * g.Hout + oy) * (long)g.Wout + (in ? x0 + i * XS : x0);
      float ev[8];
      V8<T>::load(e + q * g.lde + c0, ev);
      if (yr) {
        float yv[8];
        V8<T>::load(yr + q * g.ldyr + c0, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gj = FOLD ? ca[j] * ev[j] + (cb[j] * yv[j] + kd[j]) : ca[j] * (ev[j] - ce[j]) + cb[j] * (yv[j] - cm[j]);
          gv[i][j] = in ? gj : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[i][j] = in ? ca[j] * ev[j] : 0.f;
      }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * S + (ky - 1) * D;
      const bool vy = iy >= 0 && iy < g.Hin;
      const T* row = x + ((b * g.Hin + (vy ? iy : 0)) * (long)g.Win) * g.ldx + c0;
      typename V8<T>::Raw raw[NCOL];
      bool ok[NCOL];
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        const int ix = (S == 1) ? x0 + (c - 1) * XS : x0 * S - D + c;
        ok[c] = vy && ix >= 0 && ix < g.Win;
        raw[c] = V8<T>::load_raw(row + (long)(ix < 0 ? 0 : (ix >= g.Win ? g.Win - 1 : ix)) * g.ldx);
      }
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        float v[8];
        V8<T>::unpack(raw[c], v);
        const float lo = ok[c] ? relu_lo : 0.f, hi = ok[c] ? TSS_INF : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = clamp3(FOLD ? v[j] * sc[j] + sh[j] : (v[j] - mu[j]) * sc[j] + sh[j], lo, hi);
#pragma unroll
        for (int i = 0; i < SW; ++i)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((S == 1) ? (i + kx == c) : (i * S + kx * D == c)) {
#pragma unroll
              for (int j = 0; j < 8; ++j) accw[ky * 3 + kx][j] += gv[i][j] * v[j];
            }
      }
    }
  }
  // block reduction over the pixel lanes, three taps at a time through a [threads][24] LDS slab (24 KB of sdw): no
  // atomics (with 32 channels 64 lanes would add to the same 8 LDS words 9 times over), deterministic, 6 barriers
  float* red = sdw;  // [blockDim][24] floats <= 24.6 KB, reuses the (still unused) sdw storage
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += 3) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int j = 0; j < 8; j += 4)
          *reinterpret_cast<float4*>(red + tid * 24 + tt * 8 + j) =
              make_float4(accw[t0 + tt][j], accw[t0 + tt][j + 1], accw[t0 + tt][j + 2], accw[t0 + tt][j + 3]);
    }
    __syncthreads();
    for (int i = tid; i < g.C * 3; i += blockDim.x) {
      const int c = i / 3, tt = i - c * 3;
      const int cgc = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int q = 0; q < g.NPL; ++q) sum += red[(q * g.CV + cgc) * 24 + tt * 8 + j];
      g.ws[(long)blockIdx.x * g.C * 9 + c * 9 + t0 + tt] = sum;
    }
  }
}

// backward-data on strips of 4 INPUT pixels.  Stride 1: the flipped-tap window of g = BN'(e, y), 2*D + 4 columns.
// Stride 2 (D = 1): only output rows/columns of matching parity contribute: <= 2 rows x 3 columns.
//
// WG = true additionally produces this layer's WEIGHT gradient in the same sweep: dw[t] = sum_p g[p] a[p + off_t] =
// sum_q g[q - off_t] a[q], and (q, q - off_t) are exactly the (input pixel, window column) pairs the input gradient
// e_in[q] = sum_t w[t] g[q - off_t] already visits -- stride 2 included, where only parity-matching pairs exist for both.
// So every g value in the window is multiplied once more, by the ACTIVATED input under the strip (already loaded for
// the ReLU mask), into 72 per-lane accumulators: e, y and x are read once per layer instead of twice, and the separate
// weight-gradient launch disappears.  Rows of per-block partial sums go to g.ws as in dw_bwd_weight_strip_kernel.
// Measured (FastSCNN step, 14 layers): 1.88 ms fused vs 0.95 + 0.70 ms as two launches.  The 72 accumulators put the
// kernel at 256 VGPRs + 198 AGPRs and 1 wave/SIMD, where memory latency and VALU no longer overlap (2.8 TB/s on the
// stride-1 layers, on par with the pair; stride 2 loses).  2-pixel strips: 106 AGPRs, slower still (7.65 vs 7.28 ms/step);
// 2 waves/SIMD spills 500 B/lane.  Kept as an opt-in (TSS_FUSE_DW_BWD=1) until the per-channel constants move to LDS.
constexpr int WG_SW = 4, WG_WAVES = 1;
template <typename T, int S, int D, bool WG>
__global__ __launch_bounds__(NT_MAX, (WG ? WG_WAVES : (D == 1 ? 2 : 1))) void dw_bwd_data_strip_kernel(const DwArgs g) {
  typedef typename StatAcc<T>::type A;
  constexpr int XS = (S == 1) ? D : 1;                  // stride 1: strip pixels D apart, window of 6 output columns
  constexpr int SWW = WG ? WG_SW : SW;                   // pixels per strip
  constexpr int NCOL = (S == 1) ? (SWW + 2) : (SWW / 2 + 1);
  __shared__ __align__(16) unsigned char smem[NT_MAX * 16 * sizeof(typename StatAcc<T>::type)];
  __shared__ __align__(16) float wl[9 * 768];
  TSS_T(tq0);
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < g.lead) {   // carried weight-gradient row reduction (dw_reduce_block)
    if ((int)blockIdx.x < g.nred) dw_reduce_block(g.red_ws, g.red_dw, g.C * 9, g.red_rows, blockIdx.x, reinterpret_cast<float*>(smem));
    return;
  }
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const T* x = reinterpret_cast<const T*>(g.x);
  T* out = reinterpret_cast<T*>(g.y);
  stage_weights(wl, g.w, g.C, tid, blockDim.x);

  float ca[8], cb[8], ce[8], cm[8], mu[8], sc[8], sh[8];
  A s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
  float accw[WG ? 9 : 1][8];
#pragma unroll
  for (int t = 0; t < (WG ? 9 : 1); ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) accw[t][j] = 0.f;
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  coef8(g.ga, g.w, c0, active, 1.f, ca);
  coef8(g.yraw ? g.gb : nullptr, g.w, c0, active, 0.f, cb);
  coef8(g.yraw ? g.gce : nullptr, g.w, c0, active, 0.f, ce);
  coef8(g.yraw ? g.gmu : nullptr, g.w, c0, active, 0.f, cm);
  coef8(g.xm, g.w, c0, active, 0.f, mu);
  coef8(g.xs, g.w, c0, active, 1.f, sc);
  coef8(g.xs ? g.xb : nullptr, g.w, c0, active, 0.f, sh);
  constexpr bool FOLD = sizeof(T) == 2;
  float kd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kd[j] = -(ca[j] * ce[j]) - cb[j] * cm[j];
  __syncthreads();
  const int nstrip = ((g.Win + SWW * XS - 1) / (SWW * XS)) * XS;
  const long U = (long)g.B * g.Hin * nstrip;
  const long ntiles = (U + g.NPL - 1) / g.NPL;
  const TileRange tr = xcd_tiles((int)ntiles, g.lead);
  TSS_T(tq1);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    const long u = (long)tile * g.NPL + pl;
    if (!active || u >= U) continue;
    const int xs = (int)(u % nstrip);
    const long t2 = u / nstrip;
    const int iy = (int)(t2 % g.Hin);
    const long b = t2 / g.Hin;
    const int x0 = (xs / XS) * (SWW * XS) + (xs % XS);   // consecutive units -> consecutive pixels (coalesced lanes)
    float acc[SWW][8];
#pragma unroll
    for (int i = 0; i < SWW; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    // the producer's raw output under the strip (ReLU mask + statistics in the epilogue): issued first, so the round
    // trip runs under the three window rows instead of being exposed at the end of every strip
    const long pbase = (b * g.Hin + iy) * (long)g.Win + x0;
    typename V8<T>::Raw rx[SWW];
    if (g.x) {
#pragma unroll
      for (int i = 0; i < SWW; ++i) rx[i] = V8<T>::load_raw(x + (pbase + (x0 + i * XS < g.Win ? i * XS : 0)) * g.ldx + c0);
    }
    float av[WG ? SWW : 1][8];   // WG: activated input under the strip (zero beyond the row end)
    auto window_row = [&](const int ky) {
      const int ny = iy - (ky - 1) * D;  // = oy * S
      const int oyr = ny / S;
      // parity / border rows contribute nothing: their loads are clamped to row 0 and masked (no branch around the
      // loads, so the rows can overlap)
      const bool vy = ny >= 0 && oyr * S == ny && oyr < g.Hout;
      // stride 2: only output rows of matching parity reach an input row -- 1 or 2 of the 3 taps rows.  The input row of a
      // strip is (nearly always) the same for every lane of a wave, so a tap row nobody needs is skipped by a scalar branch
      // instead of being loaded (clamped), transformed and masked: half of the work of the stride-2 layers.
      if (!WG && S == 2 && __builtin_amdgcn_ballot_w64(vy) == 0ull) return;
      const int oy = vy ? oyr : 0;
      const long rowq = (b * g.Hout + oy) * (long)g.Wout;
      // first output column of the window: stride 1: x0 - D, then every D-th column; stride 2: x0 / 2 (x0 is a multiple of 4)
      const int oc0 = (S == 1) ? (x0 - D) : (x0 / 2);
      typename V8<T>::Raw re[NCOL], ry[NCOL];
      bool ok[NCOL];
#pragma unroll
      for (int c = 0; c < NCOL; ++c) {
        const int ox = oc0 + c * XS;
        ok[c] = vy && ox >= 0 && ox < g.Wout;
        const long q = rowq + (ox < 0 ? 0 : (ox >= g.Wout ? g.Wout - 1 : ox));
        re[c] = V8<T>::load_raw(e + q * g.lde + c0);
        if (yr) ry[c] = V8<T>::load_raw(yr + q * g.ldyr + c0);
      }
      if (WG && ky == 0) {   // rx was requested before this row: it arrives first, the row's loads stay in flight
#pragma unroll
        for (int i = 0; i < SWW; ++i) {
          float xv[8];
          V8<T>::unpack(rx[i], xv);
          const bool in = x0 + i * XS < g.Win;
#pragma unroll
          for (int j = 0; j < 8; ++j) av[i][j] = in ? fmaxf((xv[j] - mu[j]) * sc[j] + sh[j], relu_lo) : 0.f;
        }
      }
      float wv[3][8];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w0 = TSS_DW_W0(wl, ky * 3 + kx, g.C, c0);
        const float4 w1 = TSS_DW_W1(wl, ky * 3 + kx, g.C, c0);
        wv[kx][0] = w0.x; wv[kx][1] = w0.y; wv[kx][2] = w0.z; wv[kx][3] = w0.w;
        wv[kx][4] = w1.x; wv[kx][5] = w1.y; wv[kx][6] = w1.z; wv[kx][7] = w1.w;
      }
      // interior strips (every tap of every lane inside the image: all but the border rows / columns) take a copy of the
      // loop without the per-element validity selects -- 8 v_cndmask per loaded vector, 18 % of the kernel's vector instructions
      bool all_ok = true;
#pragma unroll
      for (int c = 0; c < NCOL; ++c) all_ok = all_ok && ok[c];
      const bool interior = __builtin_amdgcn_ballot_w64(!all_ok) == 0ull;
      auto columns = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
          float gvv[8];
          V8<T>::unpack(re[c], gvv);
          if (yr) {
            float yv[8];
            V8<T>::unpack(ry[c], yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float gj = FOLD ? ca[j] * gvv[j] + (cb[j] * yv[j] + kd[j]) : ca[j] * (gvv[j] - ce[j]) + cb[j] * (yv[j] - cm[j]);
              gvv[j] = (!MASKED || ok[c]) ? gj : 0.f;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gvv[j] = (!MASKED || ok[c]) ? ca[j] * gvv[j] : 0.f;
          }
          // input pixel i of the strip sees output column (oc0 + c) through tap kx when
          //   stride 1:  x0 + i*D - (kx-1)*D == x0 - D + c*D  <=>  i + (2 - kx) == c       (kx counted from the flip)
          //   stride 2:  x0 + i - (kx-1)   == 2 * (x0/2 + c)  <=>  kx == i + 1 - 2c
#pragma unroll
          for (int i = 0; i < SWW; ++i)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const bool hit = (S == 1) ? (i + (2 - kx) == c) : (kx == i + 1 - 2 * c);
              if (hit) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] += gvv[j] * wv[kx][j];
                if (WG) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) accw[WG ? ky * 3 + kx : 0][j] += gvv[j] * av[WG ? i : 0][j];
                }
              }
            }
        }
      };
      // (stride 2 only: the stride-1 instances are at 236 registers and the second copy of the loop makes them spill)
      if (!WG && S == 2 && interior) columns(std::false_type{}); else columns(std::true_type{});
    };
    if constexpr (WG) {      // unrolled: the accumulator index must be static
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) window_row(ky);
    } else {
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) window_row(ky);
    }
#pragma unroll
    for (int i = 0; i < SWW; ++i) {
      if (x0 + i * XS < g.Win) {
        const long p = pbase + i * XS;
        if (g.x_mask) {
          float xv[8];
          V8<T>::unpack(rx[i], xv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xc = xv[j] - mu[j];
            if (g.x_relu && !(xc * sc[j] + sh[j] > 0.f)) acc[i][j] = 0.f;
            acc[i][j] = V8<T>::round(acc[i][j]);
            s1[j] += (A)acc[i][j];
            s2[j] += (A)acc[i][j] * (A)xc;
          }
        }
        V8<T>::store(out + p * g.ldy + c0, acc[i]);
      }
    }
  }
#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TSS_T(tq2);
  if (g.stats) flush_stats<A>(s1, s2, g.stats, g.C, g.CV, g.NPL, cg, pl, active, smem, g.lead);
  if (WG) {
    // block reduction of the weight-gradient accumulators over the pixel lanes, three taps at a time through a
    // [threads][24] slab that reuses the weight stage (the weights are dead): deterministic, no atomics
    float* red = wl;
    float* wrow = g.ws + (long)((int)blockIdx.x - g.lead) * g.C * 9;
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
      __syncthreads();
      if (active) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int j = 0; j < 8; j += 4)
            *reinterpret_cast<float4*>(red + tid * 24 + tt * 8 + j) =
                make_float4(accw[WG ? t0 + tt : 0][j], accw[WG ? t0 + tt : 0][j + 1], accw[WG ? t0 + tt : 0][j + 2],
                            accw[WG ? t0 + tt : 0][j + 3]);
      }
      __syncthreads();
      for (int i = tid; i < g.C * 3; i += blockDim.x) {
        const int c = i / 3, tt = i - c * 3;
        const int cgc = c >> 3, j = c & 7;
        float sum = 0.f;
        for (int q = 0; q < g.NPL; ++q) sum += red[(q * g.CV + cgc) * 24 + tt * 8 + j];
        wrow[c * 9 + t0 + tt] = sum;
      }
    }
  }
#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSS_T(tq3);
  if (threadIdx.x == 0) { atomicAdd(&g_dw_timing[0], tq1 - tq0); atomicAdd(&g_dw_timing[1], tq2 - tq1); atomicAdd(&g_dw_timing[2], tq3 - tq2); atomicAdd(&g_dw_timing[7], 1ull); }
#endif
}

template <typename T>
bool launch_strip(int which, const DwArgs& g, int grid, int threads, hipStream_t st) {
#define TSS_DW_CASE(SS, DD)                                                                                         \
  if (g.stride == SS && g.dil == DD) {                                                                              \
    if (which == 0) hipLaunchKernelGGL((dw_fwd_strip_kernel<T, SS, DD>), dim3(grid), dim3(threads), 0, st, g);        \
    else if (which == 1) hipLaunchKernelGGL((dw_bwd_data_strip_kernel<T, SS, DD, false>), dim3(grid), dim3(threads), 0, st, g); \
    else hipLaunchKernelGGL((dw_bwd_weight_strip_kernel<T, SS, DD>), dim3(grid), dim3(threads), 0, st, g);            \
    return true;                                                                                                    \
  }
  TSS_DW_CASE(1, 1)
  TSS_DW_CASE(2, 1)
  TSS_DW_CASE(1, 4)
#undef TSS_DW_CASE
  return false;
}

int geometry(DwArgs& g, int* threads) {
  if (g.C <= 0 || (g.C % 8) != 0 || g.C > 768) return TSS_ERR_SHAPE;
  g.CV = g.C / 8;
  g.NPL = NT_MAX / g.CV;
  if (g.NPL < 1) return TSS_ERR_SHAPE;
  *threads = (g.CV * g.NPL + 63) / 64 * 64;
  return TSS_OK;
}

// backward-data + weight gradient in one sweep (bf16 only: the f32 parity path keeps the two separate kernels)
bool launch_strip_fused(const DwArgs& g, int grid, int threads, hipStream_t st) {
#define TSS_DW_CASE(SS, DD)                                                                                         \
  if (g.stride == SS && g.dil == DD) {                                                                              \
    hipLaunchKernelGGL((dw_bwd_data_strip_kernel<bf16_t, SS, DD, true>), dim3(grid), dim3(threads), 0, st, g);       \
    return true;                                                                                                    \
  }
  TSS_DW_CASE(1, 1)
  TSS_DW_CASE(2, 1)
  TSS_DW_CASE(1, 4)
#undef TSS_DW_CASE
  return false;
}

inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

inline bool strip_supported(int stride, int dil) { return (stride == 1 && dil == 1) || (stride == 2 && dil == 1) || (stride == 1 && dil == 4); }

// workspace rows the weight-gradient kernels of this layer write (= their grid); g.NPL / Hout / Wout must be set
inline int weight_rows(const DwArgs& g) {
  const long P = (long)g.B * g.Hout * g.Wout;
  const long U = (long)g.B * g.Hout * ((g.Wout + SW - 1) / SW);
  return strip_supported(g.stride, g.dil) ? tss::persistent_blocks((U + g.NPL - 1) / g.NPL, TSS_STAT_SLABS)
                                          : tss::persistent_blocks((P + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
}

}  // namespace

extern "C" {

int tss_dwconv3x3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                      const float* w, void* y, long ldy, double* stats,
                      int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((ldx % 8) == 0 && (ldy % 8) == 0 && ldx >= C && ldy >= C && stride >= 1 && dil >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  DwArgs g = {};
  g.x = x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.w = w;
  g.y = y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.Hin = Hin; g.Win = Win; g.C = C; g.stride = stride; g.dil = dil;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  int threads;
  const int rc = geometry(g, &threads);
  if (rc) return rc;
  const long P = (long)B * g.Hout * g.Wout;
  if (P == 0) return TSS_OK;
  const long U = (long)B * g.Hout * ((g.Wout + SW - 1) / SW);
  tss::ProfScope prof(TSS_K_DWCONV_FWD, (hipStream_t)stream,
                      ((double)B * Hin * Win + (double)P) * C * esz(dtype), 18.0 * P * C);
  if (tss::dwroll_supported(C, stride, dil, dtype)) {   // bf16, dilation 1: row-pipelined through LDS (dwroll.hip)
    tss::dwroll_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w, y, ldy, stats, B, Hin, Win, C, stride, (hipStream_t)stream);
    return tss::check_last("dwconv_fwd");
  }
  const int sgrid = tss::persistent_blocks((U + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
  const bool strip = dtype == TSS_BF16 ? launch_strip<bf16_t>(0, g, sgrid, threads, (hipStream_t)stream)
                                       : launch_strip<float>(0, g, sgrid, threads, (hipStream_t)stream);
  if (!strip) {
    const int grid = tss::persistent_blocks((P + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
    if (dtype == TSS_BF16) hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(dw_fwd_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  }
  return tss::check_last("dwconv_fwd");
}

int tss_dwconv3x3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           void* e_in, long ldei, double* bstats, const float* wg_ws, float* wg_dw,
                           int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= C && (ldei % 8) == 0 && ldei >= C && stride >= 1 && dil >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= C && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(!xraw || ((ldx % 8) == 0 && ldx >= C), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || xraw, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(e_in), TSS_ERR_ALIGN);
  DwArgs g = {};
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu; g.w = w;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.x_mask = xraw != nullptr;
  g.y = e_in; g.ldy = ldei; g.stats = bstats;
  g.B = B; g.Hin = Hin; g.Win = Win; g.C = C; g.stride = stride; g.dil = dil;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  int threads;
  const int rc = geometry(g, &threads);
  if (rc) return rc;
  const long P = (long)B * Hin * Win;
  if (P == 0) return TSS_OK;
  const long Po = (long)B * g.Hout * g.Wout;
  tss::ProfScope prof(TSS_K_DWCONV_BWD_DATA, (hipStream_t)stream,
                      ((double)Po * (yraw ? 2 : 1) + (double)P * (xraw ? 2 : 1)) * C * esz(dtype), 18.0 * Po * C);
  const long U = (long)B * Hin * ((Win + SW - 1) / SW);
  const int sgrid = tss::persistent_blocks((U + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
  const bool carry = wg_ws && wg_dw && strip_supported(stride, dil);
  if (carry) {   // the row reduction of this layer's weight gradient rides in front of the sweeping blocks
    g.red_ws = wg_ws; g.red_dw = wg_dw; g.red_rows = weight_rows(g);
    g.nred = (C * 9 + 63) / 64; g.lead = (g.nred + 7) & ~7;
  } else if (wg_ws && wg_dw) {
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((C * 9 + 63) / 64), dim3(RED_WAVES * 64), 0, (hipStream_t)stream, wg_ws, wg_dw, C * 9, weight_rows(g));
  }
  const bool strip = dtype == TSS_BF16 ? launch_strip<bf16_t>(1, g, sgrid + g.lead, threads, (hipStream_t)stream)
                                       : launch_strip<float>(1, g, sgrid + g.lead, threads, (hipStream_t)stream);
  if (!strip) {
    const int grid = tss::persistent_blocks((P + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
    if (dtype == TSS_BF16) hipLaunchKernelGGL(dw_bwd_data_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(dw_bwd_data_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  }
  return tss::check_last("dwconv_bwd_data");
}

// 1 when the one-sweep backward is also the FASTER choice for the layer (the row-pipelined stride-1 kernel); the 8-channel
// strip variant behind tss_dwconv3x3_bwd_fused stays opt-in
int tss_dwconv3x3_bwd_fused_preferred(int C, int stride, int dil, int dtype) {
  return tss::dwroll_bwd_fused_supported(C, stride, dil, dtype) ? 1 : 0;
}

int tss_dwconv3x3_bwd_fused_supported(int C, int stride, int dil, int dtype) {
  return dtype == TSS_BF16 && strip_supported(stride, dil) && C > 0 && (C % 8) == 0 && C <= 768;
}

int tss_dw_reduce_many(int njobs, const float* const* ws, float* const* dw, const int* n, const int* rows, void* stream) {
  TSS_REQUIRE(njobs >= 0 && (njobs == 0 || (ws && dw && n && rows)), TSS_ERR_SHAPE);
  for (int j0 = 0; j0 < njobs; j0 += RED_MANY) {
    ReduceJobs jobs = {};
    const int cnt = njobs - j0 < RED_MANY ? njobs - j0 : RED_MANY;
    int nmax = 0;
    for (int q = 0; q < cnt; ++q) {
      TSS_REQUIRE(ws[j0 + q] && dw[j0 + q] && n[j0 + q] > 0 && rows[j0 + q] > 0, TSS_ERR_SHAPE);
      jobs.ws[q] = ws[j0 + q]; jobs.dw[q] = dw[j0 + q]; jobs.n[q] = n[j0 + q]; jobs.rows[q] = rows[j0 + q];
      // blocks this job needs: 256 columns per block with 16-byte loads, 64 otherwise (the kernel takes the same decision)
      const bool v4 = (n[j0 + q] & 3) == 0 && ((reinterpret_cast<uintptr_t>(ws[j0 + q]) | reinterpret_cast<uintptr_t>(dw[j0 + q])) & 15u) == 0;
      const int nb = v4 ? (n[j0 + q] + 255) / 256 : (n[j0 + q] + 63) / 64;
      nmax = nb > nmax ? nb : nmax;
    }
    hipLaunchKernelGGL(dw_reduce_many_kernel, dim3(nmax, cnt), dim3(RED_WAVES * 64), 0, (hipStream_t)stream, jobs);
  }
  return tss::check_last("dw_reduce_many");
}

static int bwd_fused_impl(const void* e, long lde, const void* yraw, long ldyr,
                          const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                          const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          int x_pending, void* e_in, long ldei, double* bstats, float* ws, float* dw,
                          int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream, int* rows_out);

int tss_dwconv3x3_bwd_fused(const void* e, long lde, const void* yraw, long ldyr,
                            const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                            const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                            int x_pending, void* e_in, long ldei, double* bstats, float* ws, float* dw,
                            int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream) {
  return bwd_fused_impl(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, x, ldx, in_mean, in_scale, in_bias, in_relu, x_pending, e_in, ldei,
                        bstats, ws, dw, B, Hin, Win, C, stride, dil, dtype, stream, nullptr);
}

int tss_dwconv3x3_bwd_fused_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                  const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                                  const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                  int x_pending, void* e_in, long ldei, double* bstats, float* ws,
                                  int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream, int* rows_out) {
  TSS_REQUIRE(rows_out != nullptr, TSS_ERR_SHAPE);
  return bwd_fused_impl(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, x, ldx, in_mean, in_scale, in_bias, in_relu, x_pending, e_in, ldei,
                        bstats, ws, reinterpret_cast<float*>(ws) /* unused */, B, Hin, Win, C, stride, dil, dtype, stream, rows_out);
}

static int bwd_fused_impl(const void* e, long lde, const void* yraw, long ldyr,
                            const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                            const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          int x_pending, void* e_in, long ldei, double* bstats, float* ws, float* dw,
                          int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream, int* rows_out) {
  TSS_REQUIRE(tss_dwconv3x3_bwd_fused_supported(C, stride, dil, dtype), TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= C && (ldei % 8) == 0 && ldei >= C && (ldx % 8) == 0 && ldx >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= C && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(x && ws && dw && (!bstats || x_pending), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(e_in) && tss::aligned16(x), TSS_ERR_ALIGN);
  DwArgs g = {};
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu; g.w = w;
  g.x = x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.x_mask = x_pending != 0;
  g.y = e_in; g.ldy = ldei; g.stats = bstats; g.ws = ws; g.dw = dw;
  g.B = B; g.Hin = Hin; g.Win = Win; g.C = C; g.stride = stride; g.dil = dil;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  int threads;
  const int rc = geometry(g, &threads);
  if (rc) return rc;
  const long P = (long)B * Hin * Win;
  if (P == 0) return TSS_OK;
  const long Po = (long)B * g.Hout * g.Wout;
  if (tss::dwroll_bwd_fused_supported(C, stride, dil, dtype)) {   // row-pipelined, 4 channels per lane (dwroll.hip)
    int rows;
    {
      tss::ProfScope prof(TSS_K_DWCONV_BWD_DATA, (hipStream_t)stream,
                          ((double)Po * (yraw ? 2 : 1) + (double)P * 2) * C * esz(dtype), 36.0 * Po * C);
      rows = tss::dwroll_bwd_fused(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, x, ldx, in_mean, in_scale, in_bias, in_relu, x_pending,
                                   e_in, ldei, bstats, ws, B, Hin, Win, C, stride, (hipStream_t)stream);
    }
    if (rows_out) *rows_out = rows;      // the caller sums the rows later (tss_dw_reduce_many)
    else hipLaunchKernelGGL(dw_reduce_kernel, dim3((C * 9 + 63) / 64), dim3(RED_WAVES * 64), 0, (hipStream_t)stream, ws, dw, C * 9, rows);
    return tss::check_last("dwconv_bwd_fused");
  }
  const long U = (long)B * Hin * ((Win + WG_SW - 1) / WG_SW);
  const int sgrid = tss::persistent_blocks((U + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
  {
    tss::ProfScope prof(TSS_K_DWCONV_BWD_DATA, (hipStream_t)stream,
                        ((double)Po * (yraw ? 2 : 1) + (double)P * 2) * C * esz(dtype), 36.0 * Po * C);
    launch_strip_fused(g, sgrid, threads, (hipStream_t)stream);
  }
  if (rows_out) *rows_out = sgrid;
  else hipLaunchKernelGGL(dw_reduce_kernel, dim3((C * 9 + 63) / 64), dim3(RED_WAVES * 64), 0, (hipStream_t)stream, ws, dw, C * 9, sgrid);
  return tss::check_last("dwconv_bwd_fused");
}

int tss_dwconv3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                             const float* ga, const float* gb, const float* gce, const float* gmu,
                             const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                             float* dw, float* ws, int defer_reduce, int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= C && (ldx % 8) == 0 && ldx >= C && stride >= 1 && dil >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= C && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(xraw) && ws, TSS_ERR_ALIGN);
  DwArgs g = {};
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.dw = dw; g.ws = ws;
  g.B = B; g.Hin = Hin; g.Win = Win; g.C = C; g.stride = stride; g.dil = dil;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  int threads;
  const int rc = geometry(g, &threads);
  if (rc) return rc;
  const long P = (long)B * g.Hout * g.Wout;
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_DWCONV_BWD_WEIGHT, (hipStream_t)stream,
                      ((double)P * (yraw ? 2 : 1) + (double)B * Hin * Win) * C * esz(dtype), 18.0 * P * C);
  const long U = (long)B * g.Hout * ((g.Wout + SW - 1) / SW);
  const int sgrid = tss::persistent_blocks((U + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
  const bool strip = dtype == TSS_BF16 ? launch_strip<bf16_t>(2, g, sgrid, threads, (hipStream_t)stream)
                                       : launch_strip<float>(2, g, sgrid, threads, (hipStream_t)stream);
  int rows = sgrid;
  if (!strip) {
    const int grid = tss::persistent_blocks((P + g.NPL - 1) / g.NPL, TSS_STAT_SLABS);
    rows = grid;
    if (dtype == TSS_BF16) hipLaunchKernelGGL(dw_bwd_weight_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(dw_bwd_weight_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  }
  if (!defer_reduce)
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((C * 9 + 63) / 64), dim3(RED_WAVES * 64), 0, (hipStream_t)stream, ws, dw, C * 9, rows);
  return tss::check_last("dwconv_bwd_weight");
}

#ifdef TSS_TIMING
int tss_debug_dw_timing(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dw_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dw_timing), z, sizeof(z)); }
  return 0;
}
#endif

}  // extern "C"

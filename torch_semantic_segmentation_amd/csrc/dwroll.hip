// Depthwise 3x3 convolution, bf16, dilation 1, stride 1 / 2: row-pipelined through LDS.
//
// The strip kernels of dwconv.hip load every input vector 4.5 times (a 3 x 6 window per 4 outputs) and normalise it
// as often; the re-reads hit L1 / L2, but each one still occupies a load slot of the lane for a full round trip, and
// those kernels are bound by exactly that (2.7 TB/s algorithmic against 5-6 TB/s for a plain streaming kernel).
// Here every input element is requested ONCE per block:
//   * a block owns one SLICE of <= 8 channel vectors (64 channels = one 128-byte line per pixel) and walks down a
//     column strip of PXL = 256 / CVS pixels, one input row (stride 2: two) per step ("job");
//   * the lanes of the block load the row (one 16-byte vector each, + the two halo columns by the edge lanes), PF
//     jobs ahead, apply the producer's BatchNorm (+ReLU) ONCE and park the f32 result in a two-slot LDS ring;
//   * after one barrier each lane reads its three horizontal neighbours from LDS and updates the partial sums of the
//     three output rows the input row touches (registers): an output row leaves when its last input row has passed.
// Statistics, rounding and the summation order of the taps are those of dw_fwd_strip_kernel (measured: the two agree with an
// f64 evaluation of the layer equally often -- all but 4e-5 of the bf16 outputs -- and with each other in all but 1e-5).
#include "dwroll.h"

#include <cstdlib>

#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int ENT = NT + 16;     // vectors of one LDS row array: PXL * CVS <= 256 plus two halo pixels (2 * CVS <= 16)

struct RollArgs {
  const bf16_t* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu;
  const float* w; bf16_t* y; long ldy; double* stats;
  int B, Hin, Win, C, CV, Hout, Wout;
  int CVS, PXL, nsl, nstrips, RS, nseg, rows_used;
  int dseg, dstrip, db;   // rows_used split into (segment, strip, image) steps
  long units;
};

// position of a block in its job stream: unit (image b, column strip, row segment), step `it` inside it.  A block visits
// units brow, brow + rows_used, ...: the step is pre-split on the host (dseg, dstrip, db) so that moving on is three adds with
// carries, no division inside the loop.  b >= B <=> past the last unit.
struct Cursor { int it, b, strip, seg, x0, o0; };

__device__ __forceinline__ void advance(Cursor& c, const RollArgs& g, int n_iter) {
  if (++c.it == n_iter) {
    c.it = 0;
    c.seg += g.dseg;
    int carry = c.seg >= g.nseg ? 1 : 0;
    c.seg -= carry * g.nseg;
    c.strip += g.dstrip + carry;
    carry = c.strip >= g.nstrips ? 1 : 0;
    c.strip -= carry * g.nstrips;
    c.b += g.db + carry;
    c.x0 = c.strip * g.PXL;
    c.o0 = c.seg * g.RS;
  }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The row requests are inline assembly with hand-placed waits.  Left to the compiler, the prefetch does not survive: its
// wait-count pass loses track of the requests in flight at the loop's back edge (conditional stores in between) and puts
// s_waitcnt vmcnt(0) -- "everything" -- in front of the first use, which drains the whole PF-deep pipeline every time.
// The compiler does not see these loads at all; a use is ordered behind its wait through the "+v" operands, and the
// count passed to the wait is the number of REQUESTS OF THIS KIND issued after the one needed (memory operations retire
// in order, so younger stores in between only make the wait more conservative, never too short).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void request(u32x4& dst, const bf16_t* uniform_base, int elem_offset) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(elem_offset * 2), "s"(uniform_base));
}
template <int N> __device__ __forceinline__ void arrived(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void arrived(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}

// raw bf16 vector -> producer's BatchNorm (+ReLU), zero outside the image -> the two float4 halves of an LDS entry
__device__ __forceinline__ void park(float4* lo4, float4* hi4, int e, const u32x4& raw, bool valid, const float sc[8],
                                     const float sh[8], float relu_lo) {
  float v[8];
  V8<bf16_t>::unpack(make_uint4(raw[0], raw[1], raw[2], raw[3]), v);
  const float lo = valid ? relu_lo : 0.f, hi = valid ? TSS_INF : 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = clamp3(v[j] * sc[j] + sh[j], lo, hi);
  lo4[e] = make_float4(v[0], v[1], v[2], v[3]);
  hi4[e] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void fetch(const float4* lo4, const float4* hi4, int e, float v[8]) {
  const float4 a = lo4[e], b = hi4[e];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Per-channel sums of a block -> its slab row.  Lanes hold NC channels of one pixel lane p; red: >= (NT * 16 + NT) floats of
// LDS no longer in use.  [PXL][2][cw] partials, NT / (2 cw) threads share a column (interleaved rows), one thread adds them.
template <int NC>
__device__ __forceinline__ void flush_slab(const float (&s1)[NC], const float (&s2)[NC], double* stats, int C, int CVS, int PXL,
                                           int sl, int brow, int rows_used, int p, int cg, bool lane_on, float* red) {
  const int tid = threadIdx.x;
  const int cw = CVS * NC;
  __syncthreads();
  if (p < PXL) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      red[(p * 2 + 0) * cw + cg * NC + j] = lane_on ? s1[j] : 0.f;
      red[(p * 2 + 1) * cw + cg * NC + j] = lane_on ? s2[j] : 0.f;
    }
  }
  __syncthreads();
  const int ncol = 2 * cw, share = NT / ncol;
  float part = 0.f;
  const int col = tid % ncol, sub = tid / ncol;
  if (sub < share) {
#pragma unroll 8
    for (int q = sub; q < PXL; q += share) part += red[(q * 2 + col / cw) * cw + col % cw];
  }
  float* part_s = red + NT * 16;      // behind the [PXL][2][cw] block (<= 4096 floats)
  if (sub < share) part_s[sub * ncol + col] = part;
  __syncthreads();
  if (tid < ncol) {
    const int which = tid / cw, c = tid - which * cw;
    const int ch = sl * 64 + c;
    if (ch < C) {
      double a = 0.0;
      for (int q = 0; q < share; ++q) a += (double)part_s[q * ncol + tid];
      stats[(long)brow * 2 * C + which * C + ch] = a;
      for (int r = brow + rows_used; r < TSS_STAT_SLABS; r += rows_used) stats[(long)r * 2 * C + which * C + ch] = 0.0;
    }
  }
}

template <int S> struct RollCfg;
#ifndef TSS_ROLL_PF1
#define TSS_ROLL_PF1 6
#endif
#ifndef TSS_ROLL_PF2
#define TSS_ROLL_PF2 2
#endif
#ifndef TSS_ROLL_BPF1
#define TSS_ROLL_BPF1 4
#endif
#ifndef TSS_ROLL_BPF2
#define TSS_ROLL_BPF2 2
#endif
template <> struct RollCfg<1> { static constexpr int PF = TSS_ROLL_PF1, NRAW = 2, ROWS = 1, ARRS = 1; };   // raw: main, halo
template <> struct RollCfg<2> { static constexpr int PF = TSS_ROLL_PF2, NRAW = 6, ROWS = 2, ARRS = 2; };   // raw: per row main0, main1, halo

template <int S>
__global__ __launch_bounds__(NT, 2) void dw_fwd_roll_kernel(const RollArgs g) {
  typedef RollCfg<S> K;
  constexpr int PF = K::PF;
  extern __shared__ __align__(16) unsigned char dyn_smem[];
  typedef float4 (*Ring)[K::ROWS][K::ARRS][2][ENT];                     // [slot][row][column parity][half][entry]
  Ring ring = reinterpret_cast<Ring>(dyn_smem);
  const int tid = threadIdx.x;
  const int p = tid / g.CVS, cg = tid - p * g.CVS;
  const int sl = (int)blockIdx.x % g.nsl, brow = (int)blockIdx.x / g.nsl;
  const int cv = sl * 8 + cg;
  const bool lane_on = p < g.PXL && cv < g.CV;
  const int c0 = lane_on ? cv * 8 : 0;
  const bool halo_l = lane_on && p == 0, halo_r = lane_on && p == g.PXL - 1;
  const int ldx = (int)g.ldx;   // one input row of a layer stays below 2^31 elements (checked by the caller)

  const int n_iter = (S == 1) ? g.RS + 2 : g.RS + 1;
  Cursor ci, cc;
  ci.it = 0;
  ci.seg = brow % g.nseg;
  ci.strip = (brow / g.nseg) % g.nstrips;
  ci.b = brow / (g.nseg * g.nstrips);
  ci.x0 = ci.strip * g.PXL;
  ci.o0 = ci.seg * g.RS;
  cc = ci;

  u32x4 raw[PF][K::NRAW];
  // request the input row(s) of the job at cursor c.  Every lane issues every load, unconditionally (a lane without a halo
  // column re-requests its own vector, a cursor past the last unit re-reads the last image): with a load under a branch
  // the compiler can no longer count the requests in flight and waits for ALL of them (s_waitcnt vmcnt(0)) before each use,
  // which turns the PF-deep prefetch into a one-deep one.
  const int halo_dx = halo_l ? -1 : (halo_r ? g.PXL : p);          // stride 1: window column of the second request
  auto issue = [&](const Cursor& c, u32x4 (&r)[K::NRAW]) {
    const int b = c.b < g.B ? c.b : g.B - 1;
    if (S == 1) {
      const int iy = clampi(c.o0 - 1 + c.it, 0, g.Hin - 1);
      const bf16_t* row = g.x + ((long)b * g.Hin + iy) * g.Win * g.ldx;      // uniform: scalar base + 32-bit lane offset
      request(r[0], row, clampi(c.x0 + p, 0, g.Win - 1) * ldx + c0);
      request(r[1], row, clampi(c.x0 + halo_dx, 0, g.Win - 1) * ldx + c0);
    } else {
#pragma unroll
      for (int ry = 0; ry < 2; ++ry) {
        const int iy = clampi(2 * (c.o0 + c.it) - 1 + ry, 0, g.Hin - 1);
        const bf16_t* row = g.x + ((long)b * g.Hin + iy) * g.Win * g.ldx;
        const int ix = 2 * (c.x0 + p);
        request(r[3 * ry + 0], row, clampi(ix, 0, g.Win - 1) * ldx + c0);
        request(r[3 * ry + 1], row, clampi(ix + 1, 0, g.Win - 1) * ldx + c0);
        request(r[3 * ry + 2], row, clampi(halo_l ? 2 * c.x0 - 1 : ix, 0, g.Win - 1) * ldx + c0);
      }
    }
  };
  // the first PF jobs are requested before the per-channel constants and the weights: one round trip for all of them
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    issue(ci, raw[k]);
    advance(ci, g, n_iter);
  }

  float sc[8], sh[8];
  {
    float mu[8];
    const bool has = g.xs != nullptr, has_b = has && g.xb != nullptr, has_m = has && g.xm != nullptr;
    const float* safe = g.w;   // any readable 32 bytes
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float a[4], b[4], m[4];
      V4<float>::load(has ? g.xs + c0 + 4 * h : safe, a);
      V4<float>::load(has_b ? g.xb + c0 + 4 * h : safe, b);
      V4<float>::load(has_m ? g.xm + c0 + 4 * h : safe, m);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[4 * h + j] = has ? a[j] : 1.f;
        sh[4 * h + j] = has_b ? b[j] : 0.f;
        mu[4 * h + j] = has_m ? m[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = __builtin_fmaf(-mu[j], sc[j], sh[j]);   // (x-mu)*s+b as x*s + (b-mu*s), as in the strip kernel
  }
  float wr[9][8];     // the lane's 8 x 9 weights: 288 contiguous bytes of [C][9], 18 16-byte loads
  {
    float wf[72];
#pragma unroll
    for (int q = 0; q < 18; ++q) V4<float>::load(g.w + (long)c0 * 9 + 4 * q, wf + 4 * q);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[t][j] = wf[j * 9 + t];
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  // normalise the arrived row(s) of the job at cursor c into ring slot `slot`
  auto write = [&](const Cursor& c, const u32x4 (&r)[K::NRAW], int slot) {
    if (S == 1) {
      const int iy = c.o0 - 1 + c.it;
      const bool vy = iy >= 0 && iy < g.Hin;
      float4* lo4 = ring[slot][0][0][0];
      float4* hi4 = ring[slot][0][0][1];
      if (p < g.PXL) park(lo4, hi4, tid + g.CVS, r[0], vy && c.x0 + p < g.Win, sc, sh, relu_lo);
      if (halo_l) park(lo4, hi4, tid, r[1], vy && c.x0 - 1 >= 0, sc, sh, relu_lo);
      if (halo_r) park(lo4, hi4, tid + 2 * g.CVS, r[1], vy && c.x0 + g.PXL < g.Win, sc, sh, relu_lo);
    } else {
#pragma unroll
      for (int ry = 0; ry < 2; ++ry) {
        const int iy = 2 * (c.o0 + c.it) - 1 + ry;
        const bool vy = iy >= 0 && iy < g.Hin;
        const int ix = 2 * (c.x0 + p);
        // window column j <-> input column 2*x0 - 1 + j; even j in array 0 (entry j/2), odd j in array 1
        if (p < g.PXL) {
          park(ring[slot][ry][1][0], ring[slot][ry][1][1], tid, r[3 * ry + 0], vy && ix < g.Win, sc, sh, relu_lo);                // j = 2p+1
          park(ring[slot][ry][0][0], ring[slot][ry][0][1], tid + g.CVS, r[3 * ry + 1], vy && ix + 1 < g.Win, sc, sh, relu_lo);   // j = 2p+2
        }
        if (halo_l) park(ring[slot][ry][0][0], ring[slot][ry][0][1], tid, r[3 * ry + 2], vy && 2 * c.x0 - 1 >= 0, sc, sh, relu_lo);   // j = 0
      }
    }
  };

  float accA[8], accB[8];   // stride 1: output rows iy-1 and iy under construction; stride 2: accA only
#pragma unroll
  for (int j = 0; j < 8; ++j) { accA[j] = 0.f; accB[j] = 0.f; }

  auto emit = [&](const Cursor& c, int orow, float (&acc)[8]) {
    if (lane_on && orow < g.Hout && c.x0 + p < g.Wout) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j] = V8<bf16_t>::round(acc[j]);
        s1[j] += acc[j];
        s2[j] += acc[j] * acc[j];
      }
      // (row base: wave-uniform 64-bit arithmetic on the scalar unit; the lane adds a 32-bit offset -- the whole address as one 64-bit
      //  vector expression was ~10 vector instructions per emitted row)
      V8<bf16_t>::store(g.y + ((long)c.b * g.Hout + orow) * g.Wout * g.ldy + ((c.x0 + p) * (int)g.ldy + c0), acc);
    }
  };
  auto compute = [&](const Cursor& c, int slot) {
    if (!lane_on) return;
    if (S == 1) {
      float accC[8];
      if (c.it == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float a[8];
        fetch(ring[slot][0][0][0], ring[slot][0][0][1], tid + kx * g.CVS, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          accA[j] += a[j] * wr[6 + kx][j];
          accB[j] += a[j] * wr[3 + kx][j];
          accC[j] = kx == 0 ? a[j] * wr[0][j] : accC[j] + a[j] * wr[kx][j];
        }
      }
      if (c.it >= 2) emit(c, c.o0 + c.it - 2, accA);
#pragma unroll
      for (int j = 0; j < 8; ++j) { accA[j] = accB[j]; accB[j] = accC[j]; }
    } else {
      // window column j = 2p + kx: even j -> array 0 (entry p, p + 1), odd j -> array 1 (entry p)
      if (c.it == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) accA[j] = 0.f;
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {    // input row 2o-1: last tap row of output row o-1, first of output row o
        float a[8];
        fetch(ring[slot][0][kx & 1][0], ring[slot][0][kx & 1][1], tid + (kx >> 1) * g.CVS, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          accA[j] += a[j] * wr[6 + kx][j];
          accB[j] = kx == 0 ? a[j] * wr[0][j] : accB[j] + a[j] * wr[kx][j];
        }
      }
      if (c.it >= 1) emit(c, c.o0 + c.it - 1, accA);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {    // input row 2o: middle tap row of output row o
        float a[8];
        fetch(ring[slot][1][kx & 1][0], ring[slot][1][kx & 1][1], tid + (kx >> 1) * g.CVS, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) accB[j] += a[j] * wr[3 + kx][j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) accA[j] = accB[j];
    }
  };

  int slot = 0;
  while (cc.b < g.B) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if (cc.b >= g.B) break;
      if (S == 1) arrived<K::NRAW * (PF - 1)>(raw[k][0], raw[k][1]);
      else arrived<K::NRAW * (PF - 1)>(raw[k][0], raw[k][1], raw[k][2], raw[k][3 % K::NRAW], raw[k][4 % K::NRAW], raw[k][5 % K::NRAW]);
      write(cc, raw[k], slot);
      asm volatile("" ::: "memory");      // every use of the slot's vectors (the LDS stores) stays in front of its re-request
      issue(ci, raw[k]);
      advance(ci, g, n_iter);
      __syncthreads();
      compute(cc, slot);
      advance(cc, g, n_iter);
      slot ^= 1;
    }
  }
  // The last PF requests (issued past the end of the job stream so that the counts above stay exact) are never used, but
  // they MUST land before this point: the compiler considers their destination registers free from here on and reuses
  // them in the statistics tail, and a row arriving late would overwrite whatever lives there by then.
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    if (S == 1) arrived<0>(raw[k][0], raw[k][1]);
    else arrived<0>(raw[k][0], raw[k][1], raw[k][2], raw[k][3 % K::NRAW], raw[k][4 % K::NRAW], raw[k][5 % K::NRAW]);
  }

  // ---- statistics: the block's slab row (columns of its slice); rows no block owns are zeroed here
  if (g.stats) flush_slab<8>(s1, s2, g.stats, g.C, g.CVS, g.PXL, sl, brow, g.rows_used, p, cg, lane_on, reinterpret_cast<float*>(dyn_smem));
}

// column strips x row segments: the segment height that minimises the longest block (k units of RS + halo rows)
void plan(RollArgs& g, int S) {
  g.CV = g.C / 8;
  g.nsl = (g.CV + 7) / 8;
  g.CVS = g.CV < 8 ? g.CV : 8;
  g.PXL = NT / g.CVS;
  g.nstrips = (g.Wout + g.PXL - 1) / g.PXL;
  const int cap = TSS_STAT_SLABS / g.nsl > 0 ? TSS_STAT_SLABS / g.nsl : 1;   // 2 blocks per CU over all slices
  const int halo = S == 1 ? 2 : 1;
  long best_cost = -1;
  for (int nseg = 1; nseg <= g.Hout; ++nseg) {
    const int RS = (g.Hout + nseg - 1) / nseg;
    if (RS < 4 && nseg > 1) break;
    const int segs = (g.Hout + RS - 1) / RS;
    const long units = (long)g.B * g.nstrips * segs;
    const long k = (units + cap - 1) / cap;
    const long cost = k * (RS + halo) + 6;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      g.RS = RS; g.nseg = segs; g.units = units;
      g.rows_used = (int)((units + k - 1) / k);
    }
  }
  g.dseg = g.rows_used % g.nseg;
  g.dstrip = (g.rows_used / g.nseg) % g.nstrips;
  g.db = g.rows_used / (g.nseg * g.nstrips);
}


// ------------------------------------------------------------------------------------------------------------------
// Backward of a stride-1 layer in ONE sweep: input gradient AND weight gradient from a single pass over (e, y, x).
//
// The two strip kernels of dwconv.hip read e, y and x once each for the input gradient and once more for the weight
// gradient: 7 tensor passes.  Here a lane owns FOUR channels of one pixel (a 64-channel slice is 16 lanes), which
// halves every per-lane array -- the 9 x 4 weights AND the 9 x 4 weight-gradient accumulators fit next to each other at
// two waves per SIMD, which the 8-channel layout could not do (its fused variant in dwconv.hip runs at one wave per SIMD and
// loses to the pair) -- and the row pipeline of the forward kernel does the rest: per step one row of e, y, x is requested
// (PF steps ahead), g = BN'(e, y) and a = relu(BN(x)) are evaluated once and parked in LDS, and after the barrier a lane
// reads its three horizontal neighbours of both:
//   e_in[q] = relu'(x[q]) * sum_{ky,kx} w[ky][kx] * g[q - ky + 1][p - kx + 1]     three rows under construction, as forward
//   dW[ky][kx] += g[o][p] * a[o + ky - 1][p + kx - 1]                              o = rows OWNED by the unit only
// 4 tensor passes instead of 7.  A unit's halo rows feed its e_in rows but not its dW sums (their g belongs to the
// vertical neighbour), so every (o, p) pair is counted exactly once over the grid.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void request2(u32x2& dst, const bf16_t* uniform_base, int elem_offset) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(elem_offset * 2), "s"(uniform_base));
}
template <int N> __device__ __forceinline__ void arrived2(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}
// A raw vector that has to outlive the re-request of its slot is moved out HERE, by an instruction of its own placed after
// the wait.  A plain C++ copy does not do: the wait's "+v" operands are tied, and to keep the old value alive across the next
// request into the same registers the compiler puts the copy in FRONT of the tied statement -- before the wait, i.e. possibly
// before the row has arrived (seen in the ISA of the first version of these kernels: v_mov ahead of s_waitcnt, and statistics
// that were off by 1e-3 whenever a row was late).  tools/check_pending_regs.py scans the ISA for that pattern.
// The same hazard from the other side: the compiler may SINK a computation on a raw vector below the re-request of its slot
// (to shorten a live range) -- the raw value then outlives the request, and the relocation copy again lands in front of the
// wait.  settle() pins values derived from raw vectors before the point where it stands (an empty volatile statement is not
// reordered against the volatile requests), so every raw vector is dead when its slot is requested again.
__device__ __forceinline__ void settle(float (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void settle(float4& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
__device__ __forceinline__ u32x2 keep2(const u32x2& r) {
  u32x2 o;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(o[0]), "=&v"(o[1]) : "v"(r[0]), "v"(r[1]));
  return o;
}
__device__ __forceinline__ void unpack4(const u32x2& r, float v[4]) {
  v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
  v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct RollBwdArgs {
  const bf16_t* e; long lde; const bf16_t* yr; long ldyr; const float* ga; const float* gb; const float* gce; const float* gmu;
  const float* w;
  const bf16_t* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu, x_mask;
  bf16_t* ein; long ldei; double* stats; float* ws;
  int B, H, W, C;
  int CVS, PXL, nsl, nstrips, RS, nseg, rows_used, dseg, dstrip, db;
};
#ifdef TSS_ROLL_TIMING
// debug build (tools/ab_variants.sh ... "-DTSS_ROLL_TIMING"): cycles of wave 0 of every block of dw_bwd_roll_s1_kernel, summed:
// [wait for the row, transform + park, barrier, window + sums + emit, prologue, tail, -, blocks]
__device__ unsigned long long g_roll_timing[8];
#define TSS_RT(var) unsigned long long var; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define TSS_RT(var)
#endif
constexpr int BENT = NT + 32;      // LDS row: PXL * CVS <= 256 entries + two halo pixels (2 * CVS <= 32)
constexpr int BPF = TSS_ROLL_BPF1;   // steps in flight

struct BCursor { int it, b, strip, seg, x0, o0; };
__device__ __forceinline__ void advance(BCursor& c, const RollBwdArgs& g, int n_iter) {
  if (++c.it == n_iter) {
    c.it = 0;
    c.seg += g.dseg;
    int carry = c.seg >= g.nseg ? 1 : 0;
    c.seg -= carry * g.nseg;
    c.strip += g.dstrip + carry;
    carry = c.strip >= g.nstrips ? 1 : 0;
    c.strip -= carry * g.nstrips;
    c.b += g.db + carry;
    c.x0 = c.strip * g.PXL;
    c.o0 = c.seg * g.RS;
  }
}

// RPJ = input rows per step: 2 halves the barriers, the cursor arithmetic and the waits per byte (the steps are instruction-
// issue bound, not latency bound: DESIGN.md section 4)
template <int RPJ>
__global__ __launch_bounds__(NT, 2) void dw_bwd_roll_s1_kernel(const RollBwdArgs g) {
  constexpr int PFB = RPJ == 1 ? BPF : 2;        // steps in flight
  TSS_RT(rt_begin);
#ifdef TSS_ROLL_TIMING
  unsigned long long rt_ph[4] = {0, 0, 0, 0}, rt_ph4 = 0;
#endif
  __shared__ __align__(16) float4 rows_s[4 * RPJ][BENT];   // [slot][row] g rows, then [slot][row] activated input rows; entry = window column * CVS + lane
  auto Gs = [&](int slot_, int jr) -> float4* { return rows_s[slot_ * RPJ + jr]; };
  auto As = [&](int slot_, int jr) -> float4* { return rows_s[2 * RPJ + slot_ * RPJ + jr]; };
  static_assert(sizeof(float4) * 4 * BENT >= sizeof(float) * (NT * 16 + NT), "flush_slab scratch");
  const int tid = threadIdx.x;
  const int p = tid / g.CVS, cg = tid - p * g.CVS;
  const int sl = (int)blockIdx.x % g.nsl, brow = (int)blockIdx.x / g.nsl;
  const int ch = sl * 64 + cg * 4;
  const bool lane_on = p < g.PXL && ch < g.C;
  const int c0 = lane_on ? ch : 0;
  const bool halo_l = lane_on && p == 0, halo_r = lane_on && p == g.PXL - 1;
  const int halo_dx = halo_l ? -1 : (halo_r ? g.PXL : p);
  const int lde = (int)g.lde, ldy = (int)g.ldyr, ldx = (int)g.ldx;
  const bf16_t* ysrc = g.yr ? g.yr : g.e;           // no BatchNorm behind this layer: the y requests re-read e (cb = 0)

  const int n_iter = (g.RS + 2 + RPJ - 1) / RPJ;
  BCursor ci, cc;
  ci.it = 0;
  ci.seg = brow % g.nseg;
  ci.strip = (brow / g.nseg) % g.nstrips;
  ci.b = brow / (g.nseg * g.nstrips);
  ci.x0 = ci.strip * g.PXL;
  ci.o0 = ci.seg * g.RS;
  cc = ci;

  u32x2 raw[PFB][6 * RPJ];      // per row: e, y, x under the lane's pixel; e, y, x of its halo column (edge lanes; the others re-request)
  auto issue = [&](const BCursor& c, u32x2 (&r)[6 * RPJ]) {
    const int b = c.b < g.B ? c.b : g.B - 1;
#pragma unroll
   for (int jr = 0; jr < RPJ; ++jr) {
    const int iy = clampi(c.o0 - 1 + c.it * RPJ + jr, 0, g.H - 1);
    const long rowpix = ((long)b * g.H + iy) * g.W;
    const int xm_ = clampi(c.x0 + p, 0, g.W - 1), xh_ = clampi(c.x0 + halo_dx, 0, g.W - 1);
    const bf16_t* re = g.e + rowpix * g.lde;
    const bf16_t* ry = ysrc + rowpix * (g.yr ? g.ldyr : g.lde);
    const bf16_t* rx = g.x + rowpix * g.ldx;
    const int ldyy = g.yr ? ldy : lde;
    request2(r[6 * jr + 0], re, xm_ * lde + c0);
    request2(r[6 * jr + 1], ry, xm_ * ldyy + c0);
    request2(r[6 * jr + 2], rx, xm_ * ldx + c0);
    request2(r[6 * jr + 3], re, xh_ * lde + c0);
    request2(r[6 * jr + 4], ry, xh_ * ldyy + c0);
    request2(r[6 * jr + 5], rx, xh_ * ldx + c0);
   }
  };
#pragma unroll
  for (int k = 0; k < PFB; ++k) {
    issue(ci, raw[k]);
    advance(ci, g, n_iter);
  }

  // per-channel constants: g = ca * e + (cb * y + kd), a = relu?(x * sc + sh'), xc = x - mu
  float ca[4], cb[4], kd[4], sc[4], sh[4], mu[4];
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
    const bool has = g.xs != nullptr;
    V4<float>::load(has ? g.xs + c0 : safe, t0);
    V4<float>::load(has && g.xb ? g.xb + c0 : safe, t1);
    V4<float>::load(g.xm ? g.xm + c0 : safe, t2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = has ? t0[j] : 1.f;
      mu[j] = g.xm ? t2[j] : 0.f;
      sh[j] = __builtin_fmaf(-(has ? mu[j] : 0.f), sc[j], (has && g.xb) ? t1[j] : 0.f);
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
      for (int t = 0; t < 9; ++t) wr[t][j] = wf[j * 9 + t];
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  float dwa[9][4], accA[4], accB[4], gprev[4], aprev[3][4], s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    accA[j] = accB[j] = gprev[j] = s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) dwa[t][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) aprev[t][j] = 0.f;
  }
  u32x2 xprev = {0u, 0u};

  // g and a of one (row, column) from its raw vectors; zero outside the image
  auto ga_of = [&](const u32x2& re, const u32x2& ry, const u32x2& rx, bool valid, float4& gv, float4& av) {
    float ev[4], yv[4], xv[4], go[4], ao[4];
    unpack4(re, ev); unpack4(ry, yv); unpack4(rx, xv);
    const float lo = valid ? relu_lo : 0.f, hi = valid ? TSS_INF : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = ca[j] * ev[j] + (cb[j] * yv[j] + kd[j]);
      go[j] = valid ? gj : 0.f;
      ao[j] = clamp3(xv[j] * sc[j] + sh[j], lo, hi);
    }
    gv = make_float4(go[0], go[1], go[2], go[3]);
    av = make_float4(ao[0], ao[1], ao[2], ao[3]);
  };

  TSS_RT(rt_loop);
  int slot = 0;
  while (cc.b < g.B) {
#pragma unroll
    for (int k = 0; k < PFB; ++k) {
      if (cc.b >= g.B) break;
      TSS_RT(rt0);
#pragma unroll
      for (int jr = 0; jr < RPJ; ++jr)
        arrived2<6 * RPJ * (PFB - 1)>(raw[k][6 * jr], raw[k][6 * jr + 1], raw[k][6 * jr + 2], raw[k][6 * jr + 3], raw[k][6 * jr + 4], raw[k][6 * jr + 5]);
      TSS_RT(rt1);
      float4 gown[RPJ], aown[RPJ];
      u32x2 xcur[RPJ];
#pragma unroll
      for (int jr = 0; jr < RPJ; ++jr) {
        const int r = cc.o0 - 1 + cc.it * RPJ + jr;
        const bool vy = r >= 0 && r < g.H;
        ga_of(raw[k][6 * jr], raw[k][6 * jr + 1], raw[k][6 * jr + 2], vy && cc.x0 + p < g.W, gown[jr], aown[jr]);
        if (p < g.PXL) { Gs(slot, jr)[tid + g.CVS] = gown[jr]; As(slot, jr)[tid + g.CVS] = aown[jr]; }
        if (halo_l || halo_r) {
          float4 gh, ah;
          ga_of(raw[k][6 * jr + 3], raw[k][6 * jr + 4], raw[k][6 * jr + 5], vy && (halo_l ? cc.x0 - 1 >= 0 : cc.x0 + g.PXL < g.W), gh, ah);
          const int eh = halo_l ? tid : tid + 2 * g.CVS;
          Gs(slot, jr)[eh] = gh; As(slot, jr)[eh] = ah;
        }
        xcur[jr] = keep2(raw[k][6 * jr + 2]);
        settle(gown[jr]); settle(aown[jr]);
      }
      asm volatile("" ::: "memory");      // the LDS stores above are issued (they hold the halo values) before the slot is re-requested
      TSS_RT(rt1a);
      issue(ci, raw[k]);
      advance(ci, g, n_iter);
      TSS_RT(rt2);
      __syncthreads();
      TSS_RT(rt3);
      if (lane_on) {
#pragma unroll
       for (int jr = 0; jr < RPJ; ++jr) {
        const int idx = cc.it * RPJ + jr;          // row o0 - 1 + idx of the unit
        float G[3][4], A[3][4];     // window columns p-1, p, p+1
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 gq = Gs(slot, jr)[tid + q * g.CVS], aq = As(slot, jr)[tid + q * g.CVS];
          G[q][0] = gq.x; G[q][1] = gq.y; G[q][2] = gq.z; G[q][3] = gq.w;
          A[q][0] = aq.x; A[q][1] = aq.y; A[q][2] = aq.z; A[q][3] = aq.w;
        }
        if (idx == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
        }
        // input gradient: g row r is tap row 0 of e_in row r-1, tap row 1 of row r, tap row 2 of row r+1; tap column kx
        // reads g column p + 1 - kx = window column 2 - kx
        float accC[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          accA[j] += wr[0][j] * G[2][j]; accA[j] += wr[1][j] * G[1][j]; accA[j] += wr[2][j] * G[0][j];
          accB[j] += wr[3][j] * G[2][j]; accB[j] += wr[4][j] * G[1][j]; accB[j] += wr[5][j] * G[0][j];
          accC[j] = wr[6][j] * G[2][j]; accC[j] += wr[7][j] * G[1][j]; accC[j] += wr[8][j] * G[0][j];
        }
        const int q = cc.o0 + idx - 2;     // finished e_in row
        const bool prev_owned = idx >= 2 && idx <= g.RS + 1;
        if (prev_owned && q < g.H && cc.x0 + p < g.W) {
          float out[4];
          if (g.x_mask) {
            float xv[4];
            unpack4(jr == 0 ? xprev : xcur[jr > 0 ? jr - 1 : 0], xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool dead = g.x_relu && !(xv[j] * sc[j] + sh[j] > 0.f);
              out[j] = V8<bf16_t>::round(dead ? 0.f : accA[j]);
              s1[j] += out[j];
              s2[j] += out[j] * (xv[j] - mu[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = accA[j];
          }
          V4<bf16_t>::store(g.ein + ((long)cc.b * g.H + q) * g.W * g.ldei + ((cc.x0 + p) * (int)g.ldei + c0), out);
        }
        // weight gradient over the rows this unit owns (idx = 1 .. RS for row r, 2 .. RS + 1 for row r - 1)
        const bool own_cur = idx >= 1 && idx <= g.RS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (gprev is stored already zeroed outside the owned rows: "previous row owned" now == "current row owned" one step ago)
          const float gc = own_cur ? G[1][j] : 0.f, gp = gprev[j];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            dwa[6 + kx][j] += gp * A[kx][j];          // o = r - 1, tap row 2: a[o + 1]
            dwa[3 + kx][j] += gc * A[kx][j];          // o = r,     tap row 1
            dwa[0 + kx][j] += gc * aprev[kx][j];      // o = r,     tap row 0: a[o - 1]
          }
          gprev[j] = gc;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) aprev[kx][j] = A[kx][j];
          accA[j] = accB[j]; accB[j] = accC[j];
        }
       }
        xprev = xcur[RPJ - 1];
      }
#ifdef TSS_ROLL_TIMING
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * RPJ * PFB));     // (the store of this step; the row requests stay in flight)
      TSS_RT(rt4);
      rt_ph[0] += rt1 - rt0; rt_ph[1] += rt2 - rt1a; rt_ph[2] += rt3 - rt2; rt_ph[3] += rt4 - rt3; rt_ph4 += rt1a - rt1;
#endif
      advance(cc, g, n_iter);
      slot ^= 1;
    }
  }
  TSS_RT(rt_tail);
  // the last BPF requests are never used but must land before their registers are reused (see the forward kernel)
#pragma unroll
  for (int k = 0; k < PFB; ++k)
#pragma unroll
    for (int jr = 0; jr < RPJ; ++jr)
      arrived2<0>(raw[k][6 * jr], raw[k][6 * jr + 1], raw[k][6 * jr + 2], raw[k][6 * jr + 3], raw[k][6 * jr + 4], raw[k][6 * jr + 5]);

  float* red = reinterpret_cast<float*>(&rows_s[0][0]);
  if (g.stats) flush_slab<4>(s1, s2, g.stats, g.C, g.CVS, g.PXL, sl, brow, g.rows_used, p, cg, lane_on, red);
  // weight-gradient partial sums of the block -> its workspace row: three taps at a time through [thread][12] floats
  float* wrow = g.ws + (long)brow * g.C * 9;
  const int cw = g.CVS * 4;
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += 3) {
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
      *reinterpret_cast<float4*>(red + tid * 12 + tt * 4) =
          lane_on ? make_float4(dwa[t0 + tt][0], dwa[t0 + tt][1], dwa[t0 + tt][2], dwa[t0 + tt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < cw * 3; i += NT) {
      const int c = i / 3, tt = i - c * 3;
      const int cgc = c >> 2, j = c & 3;
      if (sl * 64 + c < g.C) {
        float sum = 0.f;
        for (int q = 0; q < g.PXL; ++q) sum += red[(q * g.CVS + cgc) * 12 + tt * 4 + j];
        wrow[(long)(sl * 64 + c) * 9 + t0 + tt] = sum;
      }
    }
  }
#ifdef TSS_ROLL_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSS_RT(rt_end);
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) atomicAdd(&g_roll_timing[q], rt_ph[q]);
    atomicAdd(&g_roll_timing[4], rt_loop - rt_begin);
    atomicAdd(&g_roll_timing[5], rt_end - rt_tail);
    atomicAdd(&g_roll_timing[6], rt_ph4);
    atomicAdd(&g_roll_timing[7], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-2 backward in one sweep.  A lane owns 4 channels of ONE output pixel (row o, column po) and of the 2 x 2 input
// pixels under it (rows 2o-1 and 2o of the step, columns 2po and 2po+1).  Per step: one row of e, y (output resolution) and
// two rows of x are requested; g and the ODD-column activations go to LDS (an input gradient at an odd column also needs
// the g of the right neighbour, the weight gradient's left tap the odd column of the left neighbour), the even columns
// stay in registers.
//   rows 2o   : e_in[2o][2po]    = w11 g[o][po]                       e_in[2o][2po+1]   = w12 g[o][po] + w10 g[o][po+1]
//   rows 2o-1 : e_in[2o-1][2po]  = w21 g[o-1][po] + w01 g[o][po]      e_in[2o-1][2po+1] = w22 g[o-1][po] + w20 g[o-1][po+1]
//                                                                                         + w02 g[o][po] + w00 g[o][po+1]
//   dW[ky][kx] += g[o'][po] * a[2o'+ky-1][2po+kx-1]   with (o', ky) = (o, 0), (o, 1) on rows 2o-1, 2o and (o-1, 2) on row 2o-1
// e, y: 1/4 pass each, x and e_in one pass each: 2.5 passes instead of the 4 of the two strip kernels.
constexpr int B2PF = TSS_ROLL_BPF2;

struct B2Cursor { int it, b, strip, seg, x0, o0; };
__device__ __forceinline__ void advance(B2Cursor& c, const RollBwdArgs& g, int n_iter) {
  if (++c.it == n_iter) {
    c.it = 0;
    c.seg += g.dseg;
    int carry = c.seg >= g.nseg ? 1 : 0;
    c.seg -= carry * g.nseg;
    c.strip += g.dstrip + carry;
    carry = c.strip >= g.nstrips ? 1 : 0;
    c.strip -= carry * g.nstrips;
    c.b += g.db + carry;
    c.x0 = c.strip * g.PXL;
    c.o0 = c.seg * g.RS;
  }
}
template <int N>
__device__ __forceinline__ void arrived10(u32x2 (&r)[10]) {
  asm volatile("s_waitcnt vmcnt(%10)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
               : "n"(N));
}

// here g.H, g.W are the INPUT extent; the output extent is (Ho, Wo)
__global__ __launch_bounds__(NT, 2) void dw_bwd_roll_s2_kernel(const RollBwdArgs g, const int Ho, const int Wo) {
  __shared__ __align__(16) float4 rows_s[6][BENT];   // [slot] g, [2 + 2 * slot + row] odd-column activations of the two input rows
  static_assert(sizeof(float4) * 6 * BENT >= sizeof(float) * (NT * 16 + NT), "flush_slab scratch");
  const int tid = threadIdx.x;
  const int p = tid / g.CVS, cg = tid - p * g.CVS;
  const int sl = (int)blockIdx.x % g.nsl, brow = (int)blockIdx.x / g.nsl;
  const int ch = sl * 64 + cg * 4;
  const bool lane_on = p < g.PXL && ch < g.C;
  const int c0 = lane_on ? ch : 0;
  const bool halo_l = lane_on && p == 0, halo_r = lane_on && p == g.PXL - 1;
  const int lde = (int)g.lde, ldx = (int)g.ldx;
  const bf16_t* ysrc = g.yr ? g.yr : g.e;
  const int ldyy = g.yr ? (int)g.ldyr : lde;
  const long ldyl = g.yr ? g.ldyr : g.lde;

  const int n_iter = g.RS + 1;
  B2Cursor ci, cc;
  ci.it = 0;
  ci.seg = brow % g.nseg;
  ci.strip = (brow / g.nseg) % g.nstrips;
  ci.b = brow / (g.nseg * g.nstrips);
  ci.x0 = ci.strip * g.PXL;
  ci.o0 = ci.seg * g.RS;
  cc = ci;

  // raw[.][0..3]: e, y under the lane's output pixel and of its right neighbour column (edge lane; others re-request their own)
  // raw[.][4..9]: x rows 2o-1 / 2o: columns 2po, 2po+1 and the column left of the strip (edge lane; others re-request 2po)
  u32x2 raw[B2PF][10];
  auto issue = [&](const B2Cursor& c, u32x2 (&r)[10]) {
    const int b = c.b < g.B ? c.b : g.B - 1;
    const int o = c.o0 + c.it;
    const long orow = ((long)b * Ho + clampi(o, 0, Ho - 1)) * Wo;
    const bf16_t* re = g.e + orow * g.lde;
    const bf16_t* ry = ysrc + orow * ldyl;
    const int po = clampi(c.x0 + p, 0, Wo - 1), pr = clampi(c.x0 + (halo_r ? p + 1 : p), 0, Wo - 1);
    request2(r[0], re, po * lde + c0);
    request2(r[1], ry, po * ldyy + c0);
    request2(r[2], re, pr * lde + c0);
    request2(r[3], ry, pr * ldyy + c0);
    const int ix = 2 * (c.x0 + p);
    const int x0c = clampi(ix, 0, g.W - 1), x1c = clampi(ix + 1, 0, g.W - 1), xlc = clampi(halo_l ? ix - 1 : ix, 0, g.W - 1);
#pragma unroll
    for (int ry2 = 0; ry2 < 2; ++ry2) {
      const int iy = clampi(2 * o - 1 + ry2, 0, g.H - 1);
      const bf16_t* rx = g.x + ((long)b * g.H + iy) * g.W * g.ldx;
      request2(r[4 + 3 * ry2], rx, x0c * ldx + c0);
      request2(r[5 + 3 * ry2], rx, x1c * ldx + c0);
      request2(r[6 + 3 * ry2], rx, xlc * ldx + c0);
    }
  };
#pragma unroll
  for (int k = 0; k < B2PF; ++k) {
    issue(ci, raw[k]);
    advance(ci, g, n_iter);
  }

  float ca[4], cb[4], kd[4], sc[4], sh[4], mu[4];
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
    const bool has = g.xs != nullptr;
    V4<float>::load(has ? g.xs + c0 : safe, t0);
    V4<float>::load(has && g.xb ? g.xb + c0 : safe, t1);
    V4<float>::load(g.xm ? g.xm + c0 : safe, t2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = has ? t0[j] : 1.f;
      mu[j] = g.xm ? t2[j] : 0.f;
      sh[j] = __builtin_fmaf(-(has ? mu[j] : 0.f), sc[j], (has && g.xb) ? t1[j] : 0.f);
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
      for (int t = 0; t < 9; ++t) wr[t][j] = wf[j * 9 + t];
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  float dwa[9][4], gprev[4], gprev_r[4], s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gprev[j] = gprev_r[j] = s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) dwa[t][j] = 0.f;
  }

  auto g_of = [&](const u32x2& re, const u32x2& ry, bool valid, float (&go)[4]) {
    float ev[4], yv[4];
    unpack4(re, ev); unpack4(ry, yv);
#pragma unroll
    for (int j = 0; j < 4; ++j) go[j] = valid ? ca[j] * ev[j] + (cb[j] * yv[j] + kd[j]) : 0.f;
  };
  auto a_of = [&](const u32x2& rx, bool valid, float (&ao)[4]) {
    float xv[4];
    unpack4(rx, xv);
    const float lo = valid ? relu_lo : 0.f, hi = valid ? TSS_INF : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ao[j] = clamp3(xv[j] * sc[j] + sh[j], lo, hi);
  };
  auto put4 = [&](float4* row, int e, const float (&v)[4]) { row[e] = make_float4(v[0], v[1], v[2], v[3]); };
  auto get4 = [&](const float4* row, int e, float (&v)[4]) { const float4 t = row[e]; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; };
  // finished input-gradient vector at (row q, column px): ReLU mask of the producer, statistics, store
  auto emit = [&](const B2Cursor& c, int q, int px, const u32x2& xraw, float (&acc)[4]) {
    if (q >= 0 && q < g.H && px < g.W) {
      float out[4];
      if (g.x_mask) {
        float xv[4];
        unpack4(xraw, xv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool dead = g.x_relu && !(xv[j] * sc[j] + sh[j] > 0.f);
          out[j] = V8<bf16_t>::round(dead ? 0.f : acc[j]);
          s1[j] += out[j];
          s2[j] += out[j] * (xv[j] - mu[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = acc[j];
      }
      V4<bf16_t>::store(g.ein + ((long)c.b * g.H + q) * g.W * g.ldei + (px * (int)g.ldei + c0), out);
    }
  };

  int slot = 0;
  while (cc.b < g.B) {
#pragma unroll
    for (int k = 0; k < B2PF; ++k) {
      if (cc.b >= g.B) break;
      arrived10<10 * (B2PF - 1)>(raw[k]);
      const int o = cc.o0 + cc.it;
      const bool vo = o < Ho;                       // o >= 0 always
      const int r0 = 2 * o - 1, r1 = 2 * o;
      const bool v0 = r0 >= 0 && r0 < g.H, v1 = r1 < g.H;
      const int ix = 2 * (cc.x0 + p);
      float gown[4], a00[4], a01[4], a10[4], a11[4];     // a[row 0/1][even/odd column]
      g_of(raw[k][0], raw[k][1], vo && cc.x0 + p < Wo, gown);
      a_of(raw[k][4], v0 && ix < g.W, a00);
      a_of(raw[k][5], v0 && ix + 1 < g.W, a01);
      a_of(raw[k][7], v1 && ix < g.W, a10);
      a_of(raw[k][8], v1 && ix + 1 < g.W, a11);
      float4* Gs = rows_s[slot];
      float4* A0 = rows_s[2 + 2 * slot];
      float4* A1 = rows_s[3 + 2 * slot];
      if (p < g.PXL) { put4(Gs, tid, gown); put4(A0, tid + g.CVS, a01); put4(A1, tid + g.CVS, a11); }
      if (halo_r) {
        float gh[4];
        g_of(raw[k][2], raw[k][3], vo && cc.x0 + g.PXL < Wo, gh);
        put4(Gs, tid + g.CVS, gh);
      }
      if (halo_l) {
        float ah[4];
        a_of(raw[k][6], v0 && ix - 1 >= 0, ah);
        put4(A0, tid, ah);
        a_of(raw[k][9], v1 && ix - 1 >= 0, ah);
        put4(A1, tid, ah);
      }
      const u32x2 x00 = keep2(raw[k][4]), x01 = keep2(raw[k][5]), x10 = keep2(raw[k][7]), x11 = keep2(raw[k][8]);
      settle(gown); settle(a00); settle(a01); settle(a10); settle(a11);
      asm volatile("" ::: "memory");
      issue(ci, raw[k]);
      advance(ci, g, n_iter);
      __syncthreads();
      if (lane_on) {
        float gr[4], al0[4], al1[4];       // g of the right neighbour, odd-column activations of the left neighbour
        get4(Gs, tid + g.CVS, gr);
        get4(A0, tid, al0);
        get4(A1, tid, al1);
        const bool first = cc.it == 0, last = cc.it == g.RS;
        float e0[4], e1[4];
        if (!first) {          // row 2o-1 belongs to this unit from its second step on
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            e0[j] = wr[7][j] * gprev[j]; e0[j] += wr[1][j] * gown[j];
            e1[j] = wr[8][j] * gprev[j]; e1[j] += wr[6][j] * gprev_r[j]; e1[j] += wr[2][j] * gown[j]; e1[j] += wr[0][j] * gr[j];
          }
          emit(cc, r0, ix, x00, e0);
          emit(cc, r0, ix + 1, x01, e1);
        }
        if (!last) {           // row 2o
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            e0[j] = wr[4][j] * gown[j];
            e1[j] = wr[5][j] * gown[j]; e1[j] += wr[3][j] * gr[j];
          }
          emit(cc, r1, ix, x10, e0);
          emit(cc, r1, ix + 1, x11, e1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gc = last ? 0.f : gown[j], gp = first ? 0.f : gprev[j];
          dwa[0][j] += gc * al0[j]; dwa[1][j] += gc * a00[j]; dwa[2][j] += gc * a01[j];     // (o, ky 0): row 2o-1
          dwa[3][j] += gc * al1[j]; dwa[4][j] += gc * a10[j]; dwa[5][j] += gc * a11[j];     // (o, ky 1): row 2o
          dwa[6][j] += gp * al0[j]; dwa[7][j] += gp * a00[j]; dwa[8][j] += gp * a01[j];     // (o-1, ky 2): row 2o-1
          gprev[j] = gown[j]; gprev_r[j] = gr[j];
        }
      }
      advance(cc, g, n_iter);
      slot ^= 1;
    }
  }
#pragma unroll
  for (int k = 0; k < B2PF; ++k) arrived10<0>(raw[k]);

  float* red = reinterpret_cast<float*>(&rows_s[0][0]);
  if (g.stats) flush_slab<4>(s1, s2, g.stats, g.C, g.CVS, g.PXL, sl, brow, g.rows_used, p, cg, lane_on, red);
  float* wrow = g.ws + (long)brow * g.C * 9;
  const int cw = g.CVS * 4;
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += 3) {
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
      *reinterpret_cast<float4*>(red + tid * 12 + tt * 4) =
          lane_on ? make_float4(dwa[t0 + tt][0], dwa[t0 + tt][1], dwa[t0 + tt][2], dwa[t0 + tt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < cw * 3; i += NT) {
      const int c = i / 3, tt = i - c * 3;
      const int cgc = c >> 2, j = c & 3;
      if (sl * 64 + c < g.C) {
        float sum = 0.f;
        for (int q = 0; q < g.PXL; ++q) sum += red[(q * g.CVS + cgc) * 12 + tt * 4 + j];
        wrow[(long)(sl * 64 + c) * 9 + t0 + tt] = sum;
      }
    }
  }
}

void plan_bwd(RollBwdArgs& g, int S, int Ho, int Wo, int rpj = 1) {
  const int cv4 = g.C / 4;
  g.nsl = (g.C + 63) / 64;
  g.CVS = cv4 < 16 ? cv4 : 16;
  g.PXL = NT / g.CVS;
  g.nstrips = (Wo + g.PXL - 1) / g.PXL;              // strips and segments partition the OUTPUT pixels
  const int cap = TSS_STAT_SLABS / g.nsl > 0 ? TSS_STAT_SLABS / g.nsl : 1;     // 2 blocks per CU over all slices
  const int halo = S == 1 ? 2 : 1;
  long best_cost = -1;
  for (int nseg = 1; nseg <= Ho; ++nseg) {
    const int RS = (Ho + nseg - 1) / nseg;
    if (RS < 4 && nseg > 1) break;
    const int segs = (Ho + RS - 1) / RS;
    const long units = (long)g.B * g.nstrips * segs;
    const long k = (units + cap - 1) / cap;
    const long cost = k * ((RS + halo + rpj - 1) / rpj) * rpj + 6;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      g.RS = RS; g.nseg = segs;
      g.rows_used = (int)((units + k - 1) / k);
    }
  }
  g.dseg = g.rows_used % g.nseg;
  g.dstrip = (g.rows_used / g.nseg) % g.nstrips;
  g.db = g.rows_used / (g.nseg * g.nstrips);
}

inline size_t ring_bytes(int S) { return (size_t)2 * S * S * 2 * ENT * sizeof(float4); }

}  // namespace

namespace tss {

bool dwroll_supported(int C, int stride, int dil, int dtype) {
  const char* sw = getenv("TSS_DW_ROLL");      // A/B switch, read per call so one process can compare the two paths
  const bool on = !(sw && atoi(sw) == 0);
  return on && dtype == TSS_BF16 && dil == 1 && (stride == 1 || stride == 2) && C >= 8 && (C % 8) == 0 && C <= 768;
}

void dwroll_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                const float* w, void* y, long ldy, double* stats,
                int B, int Hin, int Win, int C, int stride, hipStream_t stream) {
  RollArgs g = {};
  g.x = (const bf16_t*)x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.w = w; g.y = (bf16_t*)y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.Hin = Hin; g.Win = Win; g.C = C;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  plan(g, stride);
  const int grid = g.nsl * g.rows_used;
  static tss::DevOnce once;
  if (once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_fwd_roll_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_bytes(2));
  if (stride == 1) hipLaunchKernelGGL(dw_fwd_roll_kernel<1>, dim3(grid), dim3(NT), ring_bytes(1), stream, g);
  else hipLaunchKernelGGL(dw_fwd_roll_kernel<2>, dim3(grid), dim3(NT), ring_bytes(2), stream, g);
}


bool dwroll_bwd_fused_supported(int C, int stride, int dil, int dtype) {
  const char* sw = getenv("TSS_DW_ROLL_BWD");
  const bool on = !(sw && atoi(sw) == 0);
  return on && dtype == TSS_BF16 && dil == 1 && (stride == 1 || stride == 2) && C >= 8 && (C % 8) == 0 && C <= 768;
}

int dwroll_bwd_fused(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                     const float* gmu, const float* w, const void* x, long ldx, const float* in_mean, const float* in_scale,
                     const float* in_bias, int in_relu, int x_pending, void* e_in, long ldei, double* bstats, float* ws,
                     int B, int H, int W, int C, int stride, hipStream_t stream) {
  RollBwdArgs g = {};
  g.e = (const bf16_t*)e; g.lde = lde; g.yr = (const bf16_t*)yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.w = w; g.x = (const bf16_t*)x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.x_mask = x_pending; g.ein = (bf16_t*)e_in; g.ldei = ldei; g.stats = bstats; g.ws = ws;
  g.B = B; g.H = H; g.W = W; g.C = C;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const char* sw = getenv("TSS_ROLL_RPJ");               // A/B: input rows per step of the stride-1 sweep (default 2)
  const int rpj = (stride == 1 && !(sw && atoi(sw) == 1)) ? 2 : 1;
  plan_bwd(g, stride, Ho, Wo, rpj);
  if (stride == 1 && rpj == 2) hipLaunchKernelGGL(dw_bwd_roll_s1_kernel<2>, dim3(g.nsl * g.rows_used), dim3(NT), 0, stream, g);
  else if (stride == 1) hipLaunchKernelGGL(dw_bwd_roll_s1_kernel<1>, dim3(g.nsl * g.rows_used), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL(dw_bwd_roll_s2_kernel, dim3(g.nsl * g.rows_used), dim3(NT), 0, stream, g, Ho, Wo);
  return g.rows_used;
}

}  // namespace tss

#ifdef TSS_ROLL_TIMING
extern "C" int tss_debug_roll_timing(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_roll_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_roll_timing), z, sizeof(z)); }
  return 0;
}
#endif

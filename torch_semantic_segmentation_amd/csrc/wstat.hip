// Dense 3x3 convolution 128 -> 128, stride 1, padding = dilation <= 18, with the WEIGHTS STATIONARY IN REGISTERS: the atrous branches
// of the ASPP head of BASELINE config 5 (models/aspp.py, rates 6 / 12 / 18 on the 256 x 512 map of a 2048 x 4096 image) and
// ConvBlock(128, 128, 3, padding=1) of ContextNet (TSS/models/contextnet.py:55) in eval mode.
//
// Why.  A CU ingests ~12 B / clock whatever the source (MI355X_MICROARCH.md, prologue bursts; the L2 counters of the stream kernel
// of atrous.hip agree: 85 % L2 hits, 27 % of the L2 request rate, and still 5.4 us per tap).  The stream kernel pulls 64 KB of
// activation fragments plus 32 KB of weights per tap and 256 pixels through that port -- 47 B per matrix-pipe clock: the port, not
// the matrix pipe, sets its pace (97 us per launch, 15.5 % of the bf16 peak).  The convolution needs 72 matrix clocks per pixel on a
// CU, i.e. it may ingest ~860 B per pixel before the port binds.  This kernel ingests ~500:
//   * weights: all nine taps of a wave's 32 output channels live in its registers for the whole launch (9 x 4 k-steps x 2
//     fragments x 16 B = 288 registers of the 512 a lone wave per SIMD owns): 295 KB per CU, read ONCE, no LDS, no per-tap barrier;
//   * activations: a block owns a 64-pixel column strip and walks down the rows of ONE residue class modulo the dilation
//     (y = r, r + D, r + 2D, ...): consecutive output rows of that walk share two of their three input rows, so a ring of four
//     row buffers in LDS ((64 + 2 D) pixels x 272 B each) sees every input row once per walk (+ 2 / J for the ends of a walk of J
//     rows, + 2 D / 64 for the column halo); the nine taps are nine (row buffer, column shift) views of that ring -- any dilation;
//   * the four waves of a block split the OUTPUT CHANNELS (32 each) and share the ring: every activation fragment read from LDS
//     (one ds_read_b128) feeds 2 MFMAs, the weight operand comes from registers: 0.5 LDS reads per MFMA.
// Launch: 256 blocks; the output rows in walk order are cut into equal contiguous ranges (a block walks one or two segments).
// Forward of a MATERIALISED input only (no pending BatchNorm / ReLU on load); the statistics of a training-mode BatchNorm behind the
// layer leave as one slab row per block.  Everything else stays on atrous.hip / conv3x3.hip / convgemm.hip.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256, TP = 64, KC = 128, NC = 128, RS = KC, MAXD = 18, ROWPX = TP + 2 * MAXD;
constexpr int RING = 4;
constexpr int ROWVEC = ROWPX * (KC / 8);                 // 16-byte vectors of a full row buffer (1600)
constexpr int VPT = (ROWVEC + NT - 1) / NT;              // per thread (7)

__device__ __attribute__((aligned(16))) unsigned short g_wstat_zero[8];

struct WsArgs {
  const T* x; long ldx; const T* w9; T* y; long ldy; double* stats;
  int B, H, W, D;
  int nstrip;
  long nrows;       // output (row, strip) units = B * nstrip * H, in walk order (see below); block i owns units [i, i + 1) * nrows / blocks
};

template <bool STATS>
__global__ __launch_bounds__(NT, 1) void conv3x3_wstat_kernel(const WsArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);                   // [RING][ROWPX][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int D = g.D, H = g.H, W = g.W;
  const int rowpx = TP + 2 * D;                           // pixels of a row buffer in use
  const int nbase = wave * 32;

  // row loader: LDS-DMA (global_load_lds_dwordx4: no staging registers -- the 288 weight registers leave none; a register-staged
  // loader spilled to scratch and serialised its seven loads per row).  One wave-instruction fills 1 KB = four pixels of the ring
  // (lane = pixel l / 16, position l % 16); the XOR image is produced on the SOURCE side: position pos of pixel c holds chunk
  // pos ^ (c & 15).  Pixels outside the image read a 16-byte zero.  Wave w moves pieces w, w + 4, ...
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int dma_pl = lane >> 4, dma_pos = lane & 15;
  const int npiece = (rowpx + 3) >> 2;
  auto row_dma = [&](int b, int yy, int x0, int slot) {
    const bool rowin = yy >= 0 && yy < H;
    const T* rowbase = g.x + ((long)b * H + (rowin ? yy : 0)) * W * g.ldx;        // uniform
    for (int q = wave; q < npiece; q += 4) {
      const int c = q * 4 + dma_pl;
      const int xx = x0 - D + c;
      const bool ok = rowin && c < rowpx && xx >= 0 && xx < W;
      const T* src = ok ? rowbase + (long)xx * g.ldx + ((dma_pos ^ (c & 15)) << 3) : reinterpret_cast<const T*>(g_wstat_zero);
      T* dst = ring + ((long)slot * ROWPX + q * 4) * RS;             // wave-uniform
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
    }
  };

  // ---- work: the output rows in WALK ORDER -- (image, strip) major, then residue class r = y mod D, then j (y = r + j D) -- cut into
  // gridDim.x equal contiguous ranges.  A range is one or two (rarely more) walk segments; every block does the same number of rows.
  const int qd = H / D, rem = H - qd * D;                 // classes r < rem have qd + 1 rows, the others qd
  const long lo = (long)blockIdx.x * g.nrows / gridDim.x, hi = (long)(blockIdx.x + 1) * g.nrows / gridDim.x;
  auto decode = [&](long u, int& b, int& sx, int& r, int& j, int& nj) {
    const long bs = u / H;
    int pp = (int)(u - bs * H);
    sx = (int)(bs % g.nstrip); b = (int)(bs / g.nstrip);
    if (pp < rem * (qd + 1)) { r = pp / (qd + 1); j = pp - r * (qd + 1); nj = qd + 1; }
    else { pp -= rem * (qd + 1); r = rem + pp / qd; j = pp - (r - rem) * qd; nj = qd; }
  };

  // the first segment's three input rows are requested before the weights: both are in flight together
  long u = lo;
  int b = 0, sx = 0, r = 0, j0 = 0, nj = 1;
  if (u < hi) {
    decode(u, b, sx, r, j0, nj);
#pragma unroll 1
    for (int i = j0 - 1; i <= j0 + 1; ++i) row_dma(b, r + i * D, sx * TP, i - (j0 - 1));
  }

  // ---- this wave's weights: [tap][k-step][fragment] -> 16 bytes per lane.  Fragment nf, row rr holds output channel
  // nbase + (rr / 4) * 8 + nf * 4 + rr % 4: a lane's accumulators (rows fq * 4 + q of both fragments) are then 8 CONSECUTIVE channels
  // nbase + fq * 8 .. + 7 of its pixel -- one 16-byte store per pixel fragment
  bf16x8 wr[9][4][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        wr[tap][k2][nf] = *reinterpret_cast<const bf16x8*>(
            g.w9 + ((long)tap * NC + nbase + ((fr >> 2) << 3) + nf * 4 + (fr & 3)) * KC + k2 * 32 + fq * 8);

  float st1[STATS ? 8 : 1], st2[STATS ? 8 : 1];      // sums / sums of squares of this lane's 8 channels (training-mode BatchNorm behind the layer)
  if (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { st1[q] = 0.f; st2[q] = 0.f; }
  }
  bool first = true;
  while (u < hi) {
    if (!first) {
      decode(u, b, sx, r, j0, nj);
      __syncthreads();                                    // the previous segment's last row is done with the ring
#pragma unroll 1
      for (int i = j0 - 1; i <= j0 + 1; ++i) row_dma(b, r + i * D, sx * TP, i - (j0 - 1));
    }
    first = false;
    const int x0 = sx * TP;
    long left = hi - u;
    const int j1 = (nj - j0 < left) ? nj : j0 + (int)left;      // rows j0 .. j1 - 1 of this walk
    u += j1 - j0;
    __syncthreads();                                      // (waits for this wave's DMA pieces, then for the other waves)
    // input row i (class units) lives in slot (i - (j0 - 1)) % RING
#pragma unroll 1
    for (int j = j0; j < j1; ++j) {
      const bool more = j + 1 < j1;
      // the next walk step's new input row j + 2 goes to the slot of row j - 2, which nobody reads any more (the barrier that ended
      // the previous row): in flight under this row's MFMAs
      if (more) row_dma(b, r + (j + 2) * D, x0, (j + 3 - j0) % RING);
      f32x4 acc[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const int s0 = (j - j0) % RING;                     // slot of input row j - 1
      // 36 steps (tap, k-step), written out as a software pipeline: the four activation fragments of step s + 2 are requested from
      // LDS before the eight MFMAs of step s (left to itself the compiler reads one fragment, waits, issues two MFMAs, reads the next)
      // LDS image: a pixel is 256 B (no padding); its 16-byte chunk c sits at position c ^ (pixel & 15).  ds_read_b128 serves a wave in
      // four groups of 16 lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): with lane = (fr, fq) reading chunk k2 * 4 + fq of
      // pixel base + fr this XOR image is conflict-free for even column shifts, and it is the lane-linear image LDS-DMA writes
      const T* rowb[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int slot = s0 + ky;
        slot = slot >= RING ? slot - RING : slot;
        rowb[ky] = ring + (long)slot * ROWPX * RS;
      }
      int pixoff[3], key[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) { const int p = fr + kx * D; pixoff[kx] = p * RS; key[kx] = p & 15; }
      auto frag_addr = [&](int step) -> const T* {
        const int tap = step >> 2, k2 = step & 3, ky = tap / 3, kx = tap - ky * 3;
        return rowb[ky] + pixoff[kx] + ((((k2 << 2) | fq) ^ key[kx]) << 3);
      };
      bf16x8 af[3][4];                                    // fragments of steps s, s + 1, s + 2
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const T* a0 = frag_addr(p);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[p][m] = *reinterpret_cast<const bf16x8*>(a0 + m * 16 * RS);
      }
#pragma unroll
      for (int step = 0; step < 36; ++step) {
        const int cur = step % 3;
        if (step + 2 < 36) {
          const T* an = frag_addr(step + 2);
#pragma unroll
          for (int m = 0; m < 4; ++m) af[(step + 2) % 3][m] = *reinterpret_cast<const bf16x8*>(an + m * 16 * RS);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[step >> 2][step & 3][0], af[cur][m], acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[step >> 2][step & 3][1], af[cur][m], acc[m][1], 0, 0, 0);
        }
        // issue order of this step, pinned: each MFMA is followed by one of the step's LDS reads and by the register moves / address
        // arithmetic the compiler needs for the NEXT step (an MFMA holds the vector issue port for 8 of its 16 cycles: one read and one
        // or two 4-cycle moves fit behind it; issued as a burst in front of the eight MFMAs they cost 48 cycles per step)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // 2 VALU
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);               // nothing moves between steps: the reads keep their two-step distance
      }
      if (more) __syncthreads();                          // DMA of row j + 2 landed (vmcnt) and visible; every wave is done with row j - 1
      // ---- store (after the barrier: in flight under the next row's MFMAs): lane = pixel m * 16 + fr, channels nbase + fq * 8 .. + 7
      const int yy = r + j * D;
      T* yrow0 = g.y + ((long)b * H + yy) * W * g.ldy + nbase + fq * 8;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int xx = x0 + m * 16 + fr;
        if (xx < W) {
          bf16x8 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) { o[q] = (T)acc[m][0][q]; o[4 + q] = (T)acc[m][1][q]; }
          *reinterpret_cast<bf16x8*>(yrow0 + (long)xx * g.ldy) = o;
          if (STATS) {                                    // from the bits that are stored
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float v = (float)o[q]; st1[STATS ? q : 0] += v; st2[STATS ? q : 0] += v * v; }
          }
        }
      }
    }
  }
  // ---- statistics slab row of this block: a wave owns its 32 channels alone -- sum over the 16 pixel lanes, no LDS
  if (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float a = row16_sum(st1[STATS ? q : 0]), c2 = row16_sum(st2[STATS ? q : 0]);
      if (fr == 0) {
        const int ch = nbase + fq * 8 + q;
        g.stats[(long)blockIdx.x * 2 * NC + ch] = (double)a;
        g.stats[(long)blockIdx.x * 2 * NC + NC + ch] = (double)c2;
        for (int rr = blockIdx.x + gridDim.x; rr < TSS_STAT_SLABS; rr += gridDim.x) {
          g.stats[(long)rr * 2 * NC + ch] = 0.0;
          g.stats[(long)rr * 2 * NC + NC + ch] = 0.0;
        }
      }
    }
  }
}

}  // namespace

// Called by tss_conv3x3_stream_fwd's caller chain (convgemm.hip) first.  Returns false when the call is outside this kernel's domain.
bool tss_conv3x3_wstat_fwd(const void* x, long ldx, const float* in_scale, int in_relu, const void* w9, void* y, long ldy,
                           double* stats, int B, int H, int W, int Cin, int N, int stride, int dil, hipStream_t stream) {
  static const bool off = getenv("TSS_CONV3X3_WSTAT") && atoi(getenv("TSS_CONV3X3_WSTAT")) == 0;      // A/B switch
  if (off || in_scale || in_relu || stride != 1 || dil < 1 || dil > MAXD || Cin != KC || N != NC || (ldx % 8) != 0 ||
      (ldy % 8) != 0 || (long)B * H * W == 0 || H < dil || !tss::aligned16(y))
    return false;
  WsArgs g = {};
  g.x = (const T*)x; g.ldx = ldx; g.w9 = (const T*)w9; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.H = H; g.W = W; g.D = dil;
  g.nstrip = (W + TP - 1) / TP;
  g.nrows = (long)B * g.nstrip * H;
  constexpr int smem = RING * ROWPX * RS * (int)sizeof(T);
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wstat_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wstat_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  }
  const int grid = g.nrows < 256 ? (int)g.nrows : 256;
  if (stats) hipLaunchKernelGGL(conv3x3_wstat_kernel<true>, dim3(grid), dim3(NT), smem, stream, g);
  else hipLaunchKernelGGL(conv3x3_wstat_kernel<false>, dim3(grid), dim3(NT), smem, stream, g);
  return true;
}

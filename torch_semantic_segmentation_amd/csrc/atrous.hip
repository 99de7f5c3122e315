// Dense 3x3 convolution (stride 1, padding = dilation, ANY dilation) on the matrix cores with the activations streamed straight
// from global memory into the MFMA operand registers: the atrous branches of the ASPP head of BASELINE config 5
// (models/aspp.py: three 128 -> 128 convolutions with rates 6 / 12 / 18 on the 256 x 512 map of a 2048 x 4096 image,
// 38.7 GFLOP each -- the only MFMA-bound kernels of the repository: 1152 FLOP per output byte).
//
// Why not the LDS-halo kernel (conv3x3.hip).  It keeps the 3 x (64 + 2 D)-pixel halo of a 64-pixel tile in LDS (77 KB at
// D = 18) next to two 34 KB weight taps: one block per CU, 64 pixels per 288 KB of restaged weights, 3.41 ms per image for
// the three branches (the general tap loop: 3.15 ms).  Here NOTHING of the activations goes through LDS:
//   * the operand of v_mfma_f32_16x16x32_bf16 that carries the pixels wants, per lane (fr, fq), 8 consecutive channels
//     (fq * 8 ...) of pixel fr: in NHWC that IS one aligned 16-byte global load.  A tap is a pointer offset
//     ((ky - 1) * D * W + (kx - 1) * D pixels) and a validity select -- no halo, no transform, any dilation;
//   * a wave owns 64 pixels x all 128 output channels (128 accumulator registers): every 16-byte weight fragment it reads
//     from LDS feeds 4 MFMAs, every activation fragment it loads feeds 8 -- 0.25 LDS reads per MFMA (the LDS array
//     saturates at 1 per 16-cycle MFMA), so the matrix pipe, not LDS, sets the pace;
//   * a block is 4 waves = 256 consecutive pixels; the weights of a tap (32 KB, bf16 [tap][output][input]) are staged
//     into one of two LDS buffers while the previous tap's 512 MFMAs per wave run; LDS holds only these two buffers
//     (69.6 KB);
//   * one wave per SIMD (128 accumulators + two taps of activation fragments + a tap of weights in flight = ~330 registers of
//     the 512 a lone wave may use; at two waves per SIMD the same code spilled 100 registers): the activation fragments of the
//     WHOLE next tap are requested before the current tap's 512 MFMAs per wave (register double buffer, a tap ahead).
// Input must be materialised (no pending BatchNorm / ReLU on load: the decoder hands ASPP a joined tensor); statistics of
// the output (training-mode BatchNorm behind the convolution) as in pwfast.hip, one slab row per block.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256, TM = 256, NCH = 128, RS = 128 + 8;

// what a tap outside the image reads: 16-byte loads of zeros (same address arithmetic as a real pixel: fq * 8 + k-step * 32 < 256)
__device__ __attribute__((aligned(16))) unsigned short g_zero_px[256];

struct AtrousArgs {
  const T* x; long ldx; const T* w9; T* y; long ldy; double* stats;
  int B, H, W, K, N, D;
  long P; int ntiles;
};

// NKS_T: k-steps per tap when known at compile time (4: 128 input channels), 0: g.K / 32 at run time.  FULLN: N == 128 (no
// fragment guards).  The main instance <4, true> is straight-line code per tap: every guard below folds away.
template <int NKS_T, bool FULLN, bool STATS, int DIST>
__global__ __launch_bounds__(NT, 1) void conv3x3_stream_kernel(const AtrousArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* Ws = reinterpret_cast<T*>(smem);                         // [2][NCH][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int K = g.K, N = g.N, D = g.D, H = g.H, W = g.W;
  const int nks = NKS_T ? NKS_T : (K >> 5);                   // k-steps of 32 input channels per tap
  const int FN = FULLN ? 8 : ((N + 15) >> 4);                 // 16-channel output fragments
  // weight staging role: 16-byte vector wv of row wr + 16 * u
  const int nvec = K >> 3;                                    // vectors per weight row (4, 8, 12 or 16)
  const int wvv = tid % nvec, wr0 = tid / nvec, wrs = NT / nvec;
  const int nwu = (NCH + wrs - 1) / wrs;                      // passes over the 128 rows (<= 8 for K >= 32)
  auto w_issue = [&](int tap, uint4 (&wreg)[8]) {
    const T* wt = g.w9 + (long)tap * N * K;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = wr0 + u * wrs;
      if (NKS_T == 4 && FULLN) {         // 128 x 128: every pass, every row (no guards: a guarded load compiles to a branch + wait)
        wreg[u] = *reinterpret_cast<const uint4*>(wt + (long)r * K + wvv * 8);
      } else {
        const bool on = u < nwu && r < N;
        const uint4 v = *reinterpret_cast<const uint4*>(wt + (on ? (long)r * K + wvv * 8 : 0));
        wreg[u] = on ? v : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto w_store = [&](int buf, const uint4 (&wreg)[8]) {
    T* dst = Ws + buf * NCH * RS;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = wr0 + u * wrs;
      if ((NKS_T == 4 && FULLN) || (u < nwu && r < NCH)) *reinterpret_cast<uint4*>(dst + r * RS + wvv * 8) = wreg[u];
    }
  };
  float st1[STATS ? 8 : 1][4], st2[STATS ? 8 : 1][4];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }
  }

  const TileRange tr = xcd_tiles(g.ntiles);
  for (int tile = tr.begin; tile < tr.end; tile += tr.step) {
    // this lane's four pixels (fragment m: pixel tile * 256 + wave * 64 + m * 16 + fr)
    int py[4], px[4];
    long pb[4];
    bool pin[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long p = (long)tile * TM + wave * 64 + m * 16 + fr;
      pin[m] = p < g.P;
      const long pc = pin[m] ? p : 0;
      px[m] = (int)(pc % W);
      const long t = pc / W;
      py[m] = (int)(t % H);
      pb[m] = pc;
    }
    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // all 16-byte activation fragments of a tap (4 pixel fragments x nks k-steps: 64 registers): requested one whole tap ahead
    // (~2000 MFMA cycles: an L2 or HBM round trip hides under them; a single wave per SIMD has nobody else to hide it)
    // taps are visited in a per-block rotation (t + block) % 9: every CU restages the same 32 KB of weights per tap, and all of them
    // asking for the SAME tap at the same moment queues on the few L2 channels that hold it (104 -> 95 us per launch; with the weight
    // traffic removed altogether -- wrong results, timing only -- 64 us: what is left of the gap to the matrix pipe's 15.5 us is the
    // one-tap prefetch distance at one wave per SIMD.  A two-tiles-per-tap variant (half the weight traffic, 256 accumulators)
    // spilled 250 registers and took 220 us; not kept.)
    const int rot = (int)(blockIdx.x % 9);
    auto wtap = [&](int t) { const int v = t + rot; return v >= 9 ? v - 9 : v; };
    auto load_tap = [&](int t, uint4 (&dst)[4][4]) {
      const int tap = wtap(t);
      const int ky = tap / 3, kx = tap - ky * 3;
      const int dy = (ky - 1) * D, dx = (kx - 1) * D;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int yy = py[m] + dy, xx = px[m] + dx;
        const bool ok = pin[m] && yy >= 0 && yy < H && xx >= 0 && xx < W;      // (yy in range: same image)
        const T* src = (ok ? g.x + (pb[m] + (long)dy * W + dx) * g.ldx : reinterpret_cast<const T*>(g_zero_px)) + fq * 8;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
          if (k2 < nks) dst[k2][m] = *reinterpret_cast<const uint4*>(src + k2 * 32);
      }
    };
    // one k-step: 8 weight fragments x 4 pixel fragments.  The LDS read of fragment i + 1 is issued before the MFMAs of fragment i
    // (written out as a software pipeline: left to itself the compiler waits for every read right behind it)
    auto mfma_step = [&](const T* wbuf, int k2, uint4 (&cur)[4][4]) {
      bf16x8 wf = *reinterpret_cast<const bf16x8*>(wbuf + k2 * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (FULLN || i < FN) {
          bf16x8 wn = wf;
          if (i + 1 < 8 && (FULLN || i + 1 < FN)) wn = *reinterpret_cast<const bf16x8*>(wbuf + (i + 1) * 16 * RS + k2 * 32);
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, *reinterpret_cast<const bf16x8*>(&cur[k2][m]), acc[m][i], 0, 0, 0);
          wf = wn;
        }
      }
    };
    // Order inside a tap (pinned with scheduling barriers): k-step 0 first -- its operands were requested a whole tap ago, so the
    // wait the compiler puts in front of it is free, and it covers k-steps 1..3 too; only THEN the next tap's requests (weights
    // first, then activations) -- issued before that wait they would be waited for as well (the wait-count pass falls back to
    // vmcnt(0) for a loop-carried prefetch) and every tap would start with an exposed round trip (measured: 510 us per launch)
    if (DIST == 1) {
      uint4 wreg[8];
      auto do_tap = [&](int tap, uint4 (&cur)[4][4], uint4 (&nxt)[4][4]) {
        const T* wbuf = Ws + (tap & 1) * NCH * RS + fr * RS + fq * 8;
        mfma_step(wbuf, 0, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) { w_issue(wtap(tap + 1), wreg); load_tap(tap + 1, nxt); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2)
          if (k2 < nks) mfma_step(wbuf, k2, cur);
        if (tap + 1 < 9) w_store((tap + 1) & 1, wreg);
        __syncthreads();               // next tap's buffer complete / this tap's buffer free for tap + 2
      };
      uint4 xa[4][4], xb[4][4];
      load_tap(0, xa);
      w_issue(wtap(0), wreg);
      w_store(0, wreg);
      __syncthreads();
#pragma unroll 1
      for (int tp = 0; tp < 8; tp += 2) {
        do_tap(tp, xa, xb);
        do_tap(tp + 1, xb, xa);
      }
      do_tap(8, xa, xb);
    } else {
      // prefetch distance TWO taps, nine taps written out (register arrays indexed by constants only): the activation fragments of
      // tap t + 2 and its weights are requested while tap t computes -- a lone wave per SIMD has ~1.7 us of matrix work between a
      // request and its use instead of ~0.85 us (128 accumulators + 3 x 64 activation + 32 weight registers of the 512 a lone
      // wave may use; the weights stay one tap ahead: a second register set for them spilled 52 registers)
      uint4 xs[3][4][4], wr[8];
      load_tap(0, xs[0]);
      w_issue(wtap(0), wr);
      load_tap(1, xs[1]);
      w_store(0, wr);
      __syncthreads();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const T* wbuf = Ws + (tap & 1) * NCH * RS + fr * RS + fq * 8;
        mfma_step(wbuf, 0, xs[tap % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) w_issue(wtap(tap + 1), wr);
        if (tap + 2 < 9) load_tap(tap + 2, xs[(tap + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2)
          if (k2 < nks) mfma_step(wbuf, k2, xs[tap % 3]);
        if (tap + 1 < 9) w_store((tap + 1) & 1, wr);
        __syncthreads();
      }
    }
    // ---- epilogue: lane = pixel fr, channels i * 16 + fq * 4 .. + 3
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (pin[m]) {
        T* yrow = g.y + pb[m] * g.ldy + fq * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (FULLN || i < FN) {
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (T)acc[m][i][q];
            if (STATS) {
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[STATS ? i : 0][q] += rq; st2[STATS ? i : 0][q] += rq * rq; }
            }
            *reinterpret_cast<bf16x4*>(yrow + i * 16) = o;
          }
        }
      }
    }
  }
  // ---- statistics slab row of this block
  if (STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][2][NCH]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = row16_sum(st1[STATS ? i : 0][q]), w2 = row16_sum(st2[STATS ? i : 0][q]);
        if (fr == 0) {
          red[(wave * 2 + 0) * NCH + i * 16 + fq * 4 + q] = u;
          red[(wave * 2 + 1) * NCH + i * 16 + fq * 4 + q] = w2;
        }
      }
    __syncthreads();
    if (tid < N) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += (double)red[(w * 2 + 0) * NCH + tid]; b += (double)red[(w * 2 + 1) * NCH + tid]; }
      const int row = blockIdx.x, rows_used = gridDim.x;
      g.stats[(long)row * 2 * N + tid] = a;
      g.stats[(long)row * 2 * N + N + tid] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * N + tid] = 0.0;
        g.stats[(long)rr * 2 * N + N + tid] = 0.0;
      }
    }
  }
}

}  // namespace

// Called by tss_conv3x3_fwd (convgemm.hip) before the other kernels.  Returns false when the call is outside this kernel's
// domain (nothing launched): pending BatchNorm / ReLU on the input, stride != 1, channel counts.
bool tss_conv3x3_stream_fwd(const void* x, long ldx, const float* in_scale, int in_relu, const void* w9, void* y, long ldy,
                            double* stats, int B, int H, int W, int Cin, int N, int stride, int dil, hipStream_t stream) {
  static const bool off = getenv("TSS_CONV3X3_STREAM") && atoi(getenv("TSS_CONV3X3_STREAM")) == 0;      // A/B switch
  if (off || in_scale || in_relu || stride != 1 || dil < 1 || Cin < 32 || Cin > 128 || (Cin % 32) != 0 || N < 16 || N > NCH ||
      (N % 16) != 0 || (ldx % 8) != 0 || (ldy % 4) != 0 || (long)B * H * W == 0)
    return false;
  AtrousArgs g = {};
  g.x = (const T*)x; g.ldx = ldx; g.w9 = (const T*)w9; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  g.B = B; g.H = H; g.W = W; g.K = Cin; g.N = N; g.D = dil;
  g.P = (long)B * H * W;
  g.ntiles = (int)((g.P + TM - 1) / TM);
  constexpr int smem = 2 * NCH * RS * (int)sizeof(T);
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_stream_kernel<4, true, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_stream_kernel<4, true, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_stream_kernel<0, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_stream_kernel<0, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  }
  const bool full = Cin == 128 && N == 128;
  const int grid = tss::persistent_blocks(g.ntiles, TSS_STAT_SLABS);
  if (full && !stats) hipLaunchKernelGGL((conv3x3_stream_kernel<4, true, false, 1>), dim3(grid), dim3(NT), smem, stream, g);
  else if (full) hipLaunchKernelGGL((conv3x3_stream_kernel<4, true, true, 1>), dim3(grid), dim3(NT), smem, stream, g);
  else if (!stats) hipLaunchKernelGGL((conv3x3_stream_kernel<0, false, false, 1>), dim3(grid), dim3(NT), smem, stream, g);
  else hipLaunchKernelGGL((conv3x3_stream_kernel<0, false, true, 1>), dim3(grid), dim3(NT), smem, stream, g);
  return true;
}

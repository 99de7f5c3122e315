// Second stage of the 1x1 weight gradient (wgrad.hip): dW[n][k] += sum over the workspace slots the wgfast blocks wrote.
// Shared between wgrad.hip (stand-alone kernel) and pwfast.hip: when the same layer's backward-data kernel is launched
// next, these blocks ride in front of ITS grid (blockIdx < nred) -- one launch less per layer (>= 5 us each in a
// replayed graph) and the reduction runs in the shadow of the streaming blocks.
#pragma once
#include <cstdlib>
#include "common.h"

namespace tss_wg {

constexpr int TN = 128, TK = 128, NT = 256;
#ifndef WG_SMALL_PX
#define WG_SMALL_PX 256
#endif

// workspace slot extent along one axis: min(dim, 128) rounded up to the 16-wide MFMA fragment
__host__ __device__ inline int ws_dim(int d) { return ((d < TN ? d : TN) + 15) & ~15; }

// how wgrad.hip splits the pixel range of a layer: stage length, blocks per output tile, number of tiles
struct Split { int pt, nsplit, tiles; };
inline Split split_for(long P, int K, int N) {
  const int wn = N < TN ? N : TN, wk = K < TK ? K : TK;
  Split s;
  s.pt = ((wn > wk ? wn : wk) <= 64) ? 128 : 64;      // 128-pixel stages when both chunk widths are <= 64 channels
  s.tiles = ((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  const long nstage = (P + s.pt - 1) / s.pt;
  // a block amortises its set-up + partial tile over >= 512 pixels (256 and 1024 measured: worse) -- unless that leaves
  // CUs without a block (the 1/32-resolution layers: 160-192 blocks, each a serial chain of 8 stages at one memory
  // latency per stage, tools/wg_timing.sh): then 256 pixels per block (784 -> 681 us over the 27 launches; 128: 705 us)
  const long full = s.tiles * ((P + 511) / 512);
  static const long small_full = getenv("TSS_WG_SMALL_FULL") ? atol(getenv("TSS_WG_SMALL_FULL")) : 256;   // A/B override
  const long min_px = full < small_full ? WG_SMALL_PX : 512;
  const long min_stages = (min_px + s.pt - 1) / s.pt;
  long ns = 1024 / s.tiles;
  if (ns < 1) ns = 1;
  if (ns > (nstage + min_stages - 1) / min_stages) ns = (nstage + min_stages - 1) / min_stages;
  if (ns < 1) ns = 1;
  s.nsplit = (int)ns;
  return s;
}
inline int reduce_blocks(int K, int N, int tiles) { return tiles * (ws_dim(N) * ws_dim(K) / 256); }

struct ReduceArgs { const float* ws; float* dw; int nsplit, nchn, ND, KD; long drs, dcs; int nred; };

inline ReduceArgs reduce_args(const float* ws, float* dw, long P, int K, int N) {   // 1x1 layout: dW[n][k] at n*K + k
  const Split sp = split_for(P, K, N);
  ReduceArgs r;
  r.ws = ws; r.dw = dw; r.nsplit = sp.nsplit; r.nchn = (N + TN - 1) / TN; r.ND = N; r.KD = K; r.drs = K; r.dcs = 1;
  r.nred = reduce_blocks(K, N, sp.tiles);
  return r;
}

// One block (256 threads) sums 256 consecutive elements of the slots (one float4 per lane: 1 KB per wave load), its 4
// waves taking slots w, w+4, ... with up to 16 loads in flight per lane.  `part` = 4 x 64 float4 of LDS.
__device__ __forceinline__ void reduce_block(const ReduceArgs& r, int bid, float4* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TNe = ws_dim(r.ND), TKe = ws_dim(r.KD);
  const int slot_n = TNe * TKe, segs = slot_n >> 8;               // slot_n is a multiple of 256
  const int tile = bid / segs, seg = bid - tile * segs;
  // slot layout = the MFMA accumulator layout, fragment after fragment: element (n, k) of the tile sits at
  //   ((n / 16) * (TKe / 16) + k / 16) * 256 + (n % 4) * 64 + ((n % 16) / 4) * 16 + k % 16
  // so a wgfast store instruction (one accumulator register of 64 lanes) writes 256 contiguous bytes, and 4 consecutive
  // elements here are 4 consecutive k of one row
  const int idx = seg * 256 + lane * 4;
  const int frag = seg, off = lane * 4;                           // one fragment per 256-element segment
  const int fi = frag / (TKe >> 4), fj = frag - fi * (TKe >> 4);
  const int nl = fi * 16 + ((off & 63) >> 4) * 4 + (off >> 6), kl = fj * 16 + (off & 15);
  const int nc = tile % r.nchn, kc = tile / r.nchn;
  const int ncw = (r.ND - nc * TN < TN) ? (r.ND - nc * TN) : TN, kcw = (r.KD - kc * TK < TK) ? (r.KD - kc * TK) : TK;
  const int FN = (ncw + 15) >> 4, FK = (kcw + 15) >> 4;
  const bool valid = nl < FN * 16 && kl < FK * 16;                // written by the blocks of this tile
  const float* col = r.ws + (long)tile * r.nsplit * slot_n + (valid ? idx : 0);
  float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = wave; r0 < r.nsplit; r0 += 4 * 16) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int q = r0 + 4 * u; v[u] = *reinterpret_cast<const float4*>(col + (long)(q < r.nsplit ? q : 0) * slot_n); }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (r0 + 4 * u < r.nsplit) { sacc.x += v[u].x; sacc.y += v[u].y; sacc.z += v[u].z; sacc.w += v[u].w; }
    }
  }
  part[wave * 64 + lane] = sacc;
  __syncthreads();
  if (threadIdx.x < 64 && valid) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 p4 = part[q * 64 + threadIdx.x]; t[0] += p4.x; t[1] += p4.y; t[2] += p4.z; t[3] += p4.w; }
    const int n = nc * TN + nl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kc * TK + kl + q;
      if (n < r.ND && k < r.KD) r.dw[(long)n * r.drs + (long)k * r.dcs] += t[q];
    }
  }
}

}  // namespace tss_wg

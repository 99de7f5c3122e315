// BatchNorm statistics reduction shared by the stand-alone finalize kernels (pointwise.hip) and by the blocks that ride in front
// of another launch (wgrad.hip: the BatchNorm-backward finalize of the NEXT layer of the backward pass in front of a
// weight-gradient grid -- the two are independent, so the finalize costs no launch of its own).
#pragma once
#include "common.h"

namespace tss_fin {

// Column sums over the TSS_STAT_SLABS partial rows for a group of FIN_CH channels per block.  A wave's 64 lanes are
// FIN_RG row groups x 2 columns (sum, second moment) x FIN_CH channels: every load instruction reads 2 x FIN_RG contiguous
// 64-byte segments; the 16 waves take interleaved rows (8 loads per lane, all in flight at once) and meet in LDS.
// History: one wave per channel with lanes along the ROWS (64 different cache lines per instruction): 8-10 us per finalize
// x 88 launches per step; 32 channels per block: 6.4 us -- each block still streamed 262 KB through ONE CU, and a layer
// has only C/32 = 2..24 such blocks; 8 channels per block: 65 KB per block, four times as many CUs pulling: 5.35 us
// (4 channels per block, 32-byte segments: 6.5 us).
constexpr int FIN_CH = 8, FIN_RG = 4, FIN_WAVES = 4, FIN_NT = FIN_WAVES * 64;
static_assert(2 * FIN_CH * FIN_RG == 64, "one wave = row groups x 2 columns x channels");
__device__ __forceinline__ void slab_sum(const double* slabs, int C, int blk, double* s0, double* s1, int* c_out) {
  __shared__ double red[FIN_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & (FIN_CH - 1), hs = (lane / FIN_CH) & 1, rg = lane / (2 * FIN_CH);
  const int c = blk * FIN_CH + cl;
  const bool in = c < C;
  const double* col = slabs + (long)hs * C + (in ? c : 0);
  constexpr int RSTEP = FIN_WAVES * FIN_RG;        // rows covered by one load instruction of the block
  constexpr int R = TSS_STAT_SLABS / RSTEP;        // loads per lane
  static_assert(TSS_STAT_SLABS % RSTEP == 0, "slab rows must divide evenly");
  double acc = 0.0;
  double v[R];   // all loads of the lane in flight: one memory round trip per finalize
#pragma unroll
  for (int u = 0; u < R; ++u) v[u] = col[(long)(wave * FIN_RG + rg + RSTEP * u) * 2 * C];
#pragma unroll
  for (int u = 0; u < R; ++u) acc += v[u];
  red[wave][lane] = in ? acc : 0.0;
  __syncthreads();
  double a = 0.0, b = 0.0;
  if (threadIdx.x < FIN_CH) {
#pragma unroll
    for (int w = 0; w < FIN_WAVES; ++w)
#pragma unroll
      for (int q = 0; q < FIN_RG; ++q) {
        a += red[w][q * 2 * FIN_CH + threadIdx.x];
        b += red[w][q * 2 * FIN_CH + FIN_CH + threadIdx.x];
      }
  }
  *s0 = a; *s1 = b; *c_out = blk * FIN_CH + threadIdx.x;   // valid for threadIdx.x < FIN_CH
}


// bstats = [sum(e), sum(e*(y-mean))] over the N = count elements of each channel (centred: no cancellation).
// training: g = k*(e - c1 - xhat*c2), k = gamma*invstd, c1 = sum(e)/N, c2 = sum(e*xhat)/N, xhat = (y-mean)*invstd
//           =>  g = ga*(e - ce) + gb*(y - mean)   with ga = k, ce = c1, gb = -k*c2*invstd
// frozen  : g = k*e
// One block of FIN_NT threads finalizes channels [blk * FIN_CH, (blk + 1) * FIN_CH).
__device__ __forceinline__ void bn_bwd_finalize_block(const tss_bn_bwd_job& j, int blk) {
  const int C = j.C;
  const int cp = min(blk * FIN_CH + (int)(threadIdx.x & (FIN_CH - 1)), C - 1);
  const float r_in = j.invstd[cp];                       // requested ahead of the slab rows, as in bn_finalize_kernel
  const float g_in = (j.gamma ? j.gamma : j.invstd)[cp];
  const float dg_in = (j.dgamma ? j.dgamma : j.invstd)[cp];
  const float db_in = (j.dbeta ? j.dbeta : j.invstd)[cp];
  double se, sey;
  int c;
  slab_sum(j.bstats, C, blk, &se, &sey, &c);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double r = r_in;
  const double dg = r * sey;
  const double db = se;
  if (j.dgamma) j.dgamma[c] = (j.accumulate ? dg_in : 0.f) + (float)dg;
  if (j.dbeta) j.dbeta[c] = (j.accumulate ? db_in : 0.f) + (float)db;
  const double k = (j.gamma ? (double)g_in : 1.0) * r;
  if (j.training) {
    const double c1 = db / j.count, c2 = dg / j.count;
    j.ga[c] = (float)k;
    j.gb[c] = (float)(-k * c2 * r);
    j.gce[c] = (float)c1;
  } else {
    j.ga[c] = (float)k; j.gb[c] = 0.f; j.gce[c] = 0.f;
  }
}
inline int fin_blocks(int C) { return (C + FIN_CH - 1) / FIN_CH; }

}  // namespace tss_fin

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


// ---- the exchange of cross-replica BatchNorm statistics inside a finalize block (see csrc/xchg.hip for the protocol)
constexpr int XMAXB = 96;                 // blocks per exchange: channels <= 768, 8 per block
constexpr int XCELL = 18;                 // 64-bit words per cell: 16 sums, the count, the flag
constexpr int XSLOTS = 4;
constexpr int XMAXW = 8;                  // ranks of one node
constexpr long XHEAD = 8;                 // header words: [0] error code (1: a peer's flag did not arrive in time)

__host__ __device__ inline long cell_off(int slot, int rank, int blk) { return XHEAD + (((long)slot * XMAXW + rank) * XMAXB + blk) * XCELL; }

struct Xchg {
  unsigned long long* peers[XMAXW];       // every rank's mailbox as mapped into THIS process (peers[rank] = our own)
  int rank, world;
  unsigned long long* ctr;                // [XMAXB] call counters of this rank (ordinary device memory)
};

__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (a, b) of thread t < FIN_CH: this rank's two sums of channel blk * FIN_CH + t; cnt: its element count.  Returns the sums over
// all ranks (valid for t < FIN_CH) and the global count.  The whole exchange is done by WAVE 0 of the block (the 8 threads that hold
// the sums, the <= 8 pollers and the flag writer are all in it), so the only ordering it needs is the wave's own: data stores
// (system scope: write-through, past every cache) -> s_waitcnt vmcnt(0) -> flag store; flag poll -> data loads, all system-scope loads.
// No __threadfence_system (an L2 write-back + invalidate, ~3.5 us each way) and no block barrier.
__device__ __forceinline__ void exchange(const Xchg& x, int blk, double& a, double& b, double& cnt) {
  const int t = threadIdx.x;
  if (t >= 64) return;
  unsigned long long seq = (t == 0) ? x.ctr[blk] + 1ull : 0ull;
  seq = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(seq >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)seq);
  const int slot = (int)(seq % XSLOTS);
  if (t < FIN_CH) {
    for (int p = 0; p < x.world; ++p) {
      unsigned long long* cell = x.peers[p] + cell_off(slot, x.rank, blk);
      st_sys(cell + 2 * t, (unsigned long long)__double_as_longlong(a));
      st_sys(cell + 2 * t + 1, (unsigned long long)__double_as_longlong(b));
      if (t == 0) st_sys(cell + 16, (unsigned long long)__double_as_longlong(cnt));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every data store of the wave has been acknowledged
  if (t == 0) {
    for (int p = 0; p < x.world; ++p) st_sys(x.peers[p] + cell_off(slot, x.rank, blk) + 17, seq);
  }
  // the peers' cells arrive in OUR mailbox: one poller per rank, bounded (~4 s of the 100 MHz wall clock)
  if (t < x.world) {
    const unsigned long long* flag = x.peers[x.rank] + cell_off(slot, t, blk) + 17;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (ld_sys(flag) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) {
        st_sys(x.peers[x.rank], 1ull);          // error word of our own mailbox (tss_bn_xchg_error)
        break;
      }
    }
  }
  // (the wave reconverges here: every poller has seen its flag before any lane loads a cell)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    const int tt = t < FIN_CH ? t : 0;
    for (int p = 0; p < x.world; ++p) {    // rank order: the same sum, bit for bit, on every rank
      const unsigned long long* cell = x.peers[x.rank] + cell_off(slot, p, blk);
      sa += __longlong_as_double((long long)ld_sys(cell + 2 * tt));
      sb += __longlong_as_double((long long)ld_sys(cell + 2 * tt + 1));
      sc += __longlong_as_double((long long)ld_sys(cell + 16));
    }
    if (t < FIN_CH) { a = sa; b = sb; }
    cnt = sc;
  }
  if (t == 0) x.ctr[blk] = seq;
}


// bstats = [sum(e), sum(e*(y-mean))] over the N = count elements of each channel (centred: no cancellation).
// training: g = k*(e - c1 - xhat*c2), k = gamma*invstd, c1 = sum(e)/N, c2 = sum(e*xhat)/N, xhat = (y-mean)*invstd
//           =>  g = ga*(e - ce) + gb*(y - mean)   with ga = k, ce = c1, gb = -k*c2*invstd
// frozen  : g = k*e
// One block of FIN_NT threads finalizes channels [blk * FIN_CH, (blk + 1) * FIN_CH).  j.xchg_world > 0 (cross-replica statistics,
// apex semantics): the input gradient uses the GLOBAL sums, d(gamma) / d(beta) this replica's own.
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
  const double se_l = se, sey_l = sey;
  double cnt = j.count;
  if (j.xchg_world > 0) {
    Xchg x;
#pragma unroll
    for (int p = 0; p < XMAXW; ++p) x.peers[p] = (unsigned long long*)j.xchg_peers[p];
    x.rank = j.xchg_rank; x.world = j.xchg_world; x.ctr = (unsigned long long*)j.xchg_counters;
    exchange(x, blk, se, sey, cnt);
  }
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double r = r_in;
  if (j.dgamma) j.dgamma[c] = (j.accumulate ? dg_in : 0.f) + (float)(r * sey_l);
  if (j.dbeta) j.dbeta[c] = (j.accumulate ? db_in : 0.f) + (float)se_l;
  const double k = (j.gamma ? (double)g_in : 1.0) * r;
  if (j.training) {
    const double c1 = se / cnt, c2 = r * sey / cnt;
    j.ga[c] = (float)k;
    j.gb[c] = (float)(-k * c2 * r);
    j.gce[c] = (float)c1;
  } else {
    j.ga[c] = (float)k; j.gb[c] = 0.f; j.gce[c] = 0.f;
  }
}
inline int fin_blocks(int C) { return (C + FIN_CH - 1) / FIN_CH; }

}  // namespace tss_fin

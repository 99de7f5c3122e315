// Small per-channel / elementwise kernels of the hot path (all HBM- or latency-bound):
//   * BatchNorm statistics -> affine ("finalize"), eval affine, BatchNorm-backward coefficients
//   * join: out = relu?(affA(a) + affB(b))   -- the only place a normalised activation is written
//   * join backward: e = dout * relu'(out) plus the per-channel sums the two BN-backwards need
//   * dropout (counter-based Philox, mask recomputed in backward), bias gradient, 3x3 weight
//     re-layout, fused flat AdamW
#include <cstdlib>
#include "common.h"
#include "bnfin.h"

namespace {

constexpr int NT = 256;

using namespace tss_fin;

__global__ __launch_bounds__(FIN_NT) void bn_finalize_kernel(const double* sums, double count, const float* gamma,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   long long* num_batches, float* mean_out, float* invstd_out,
                                   float* scale, int C) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  // the per-channel operands of the tail are requested before the slab rows (independent of them): their round trip
  // hides under the slab reduction instead of following it.  Null pointers read a valid dummy and are selected away.
  const int cp = min(blockIdx.x * FIN_CH + (int)(threadIdx.x & (FIN_CH - 1)), C - 1);
  const float g_in = (gamma ? gamma : mean_out)[cp];
  const float rm_in = (running_mean ? running_mean : mean_out)[cp];
  const float rv_in = (running_var ? running_var : mean_out)[cp];
  double ssum, ssq;
  int c;
  slab_sum(sums, C, blockIdx.x, &ssum, &ssq, &c);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double mean = ssum / count;
  double var = ssq / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? g_in : 1.f;
  const float m = (float)mean;
  mean_out[c] = m;
  invstd_out[c] = invstd;
  scale[c] = g * invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * rm_in + momentum * m;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * rv_in + momentum * (float)unbiased;
  }
}

// ---- cross-replica (Sync) BatchNorm: the slab rows of a replica are summed to one [2C + 1] f64 vector (sums, sums of
// squares / centred products, element count), the caller all-reduces it over the ranks (RCCL), and the finalize
// kernels below read the global vector.  apex.parallel.SyncBatchNorm semantics: forward normalises with the global
// batch statistics and updates the running statistics with them; backward uses the global sums for the input
// gradient and the LOCAL sums for d(gamma), d(beta) (the gradient all-reduce of the data-parallel step adds those).
__global__ __launch_bounds__(FIN_NT) void slab_reduce_kernel(const double* slabs, double count, double* out, int C) {
  double s0, s1;
  int c;
  slab_sum(slabs, C, blockIdx.x, &s0, &s1, &c);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * C] = count;
  if (threadIdx.x >= FIN_CH || c >= C) return;
  out[c] = s0;
  out[C + c] = s1;
}

__global__ void bn_finalize_sync_kernel(const double* gsums, const float* gamma, float eps, float momentum,
                                        float* running_mean, float* running_var, long long* num_batches,
                                        float* mean_out, float* invstd_out, float* scale, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches) *num_batches += 1;
  if (c >= C) return;
  const double count = gsums[2 * C];
  const double mean = gsums[c] / count;
  double var = gsums[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float m = (float)mean;
  mean_out[c] = m;
  invstd_out[c] = invstd;
  scale[c] = (gamma ? gamma[c] : 1.f) * invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_sync_kernel(const double* bstats, const double* gsums,
                                                                      const float* invstd, const float* gamma, int accumulate,
                                                                      float* dgamma, float* dbeta, float* ga, float* gb,
                                                                      float* gce, int C) {
  double se, sey;
  int c;
  slab_sum(bstats, C, blockIdx.x, &se, &sey, &c);           // this replica's sums: parameter gradients
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double r = invstd[c];
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)(r * sey);
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)se;
  const double count = gsums[2 * C];
  const double k = (gamma ? (double)gamma[c] : 1.0) * r;
  const double c1 = gsums[c] / count, c2 = r * gsums[C + c] / count;   // global means of e and e * xhat
  ga[c] = (float)k;
  gb[c] = (float)(-k * c2 * r);
  gce[c] = (float)c1;
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* running_mean,
                                      const float* running_var, float eps, float* mean_out, float* invstd_out,
                                      float* scale, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(running_var[c] + eps);
  mean_out[c] = running_mean[c];
  invstd_out[c] = invstd;
  scale[c] = (gamma ? gamma[c] : 1.f) * invstd;
}

// every eval-mode BatchNorm of a model in one launch.  table: njobs x 6 int64 -- gamma (or 0), running_mean, running_var,
// out (mean at out[0..C), invstd at out[C..2C), scale at out[2C..3C)), C, eps (f32 bits in the low word).
__global__ __launch_bounds__(128) void bn_eval_affine_batched_kernel(const long long* table, int njobs) {
  const long long* job = table + (long)blockIdx.y * 6;
  const int C = (int)job[4];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* gamma = (const float*)job[0];
  const float* rm = (const float*)job[1];
  const float* rv = (const float*)job[2];
  float* out = (float*)job[3];
  const float eps = __int_as_float((int)job[5]);
  const float g = (gamma ? gamma : rm)[c];
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  out[c] = rm[c];
  out[C + c] = invstd;
  out[2 * C + c] = (gamma ? g : 1.f) * invstd;
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_kernel(const tss_bn_bwd_job j) { bn_bwd_finalize_block(j, blockIdx.x); }

// ------------------------------------------------------------------------------------------ join
struct JoinArgs {
  const void* a; long lda; const float* ma; const float* sa; const float* ba;
  const void* b; long ldb; const float* mb; const float* sb; const float* bb;
  void* out; long ldo; int relu;
  // backward
  const void* dout; long lddo; void* e; long lde; double* stats_a; double* stats_b;
  long P; int C, CV, NPL;
  // dropout folded into a ReLU join (Classifier: ... -> BN -> ReLU -> Dropout): forward keeps/zeroes with the Philox mask of
  // dropout_kernel; backward needs no mask -- out > 0 <=> kept and active -- only the 1/(1-p) factor (dscale)
  float drop_p; const unsigned long long* seed_slot; float dscale;
};

// 8 per-channel constants of a lane: two unconditional 16-byte loads through a null-safe pointer, then a select
__device__ __forceinline__ void coef8(const float* p, const float* safe, int c0, bool active, float dflt, float out[8]) {
  const bool has = p != nullptr;
  const float* q = has ? p + (active ? c0 : 0) : safe;
  float v[8];
  V4<float>::load(q, v);
  V4<float>::load(q + 4, v + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (has && active) ? v[j] : dflt;
}

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__global__ void dropout_tick_kernel(unsigned long long* counter, unsigned long long* slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { *slot = *counter; *counter += 1ull; }
}

template <typename T>
__global__ __launch_bounds__(NT) void join_fwd_kernel(const JoinArgs g) {
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  if (pl >= g.NPL) return;
  const int c0 = cg * 8;
  const T* a = reinterpret_cast<const T*>(g.a);
  const T* b = reinterpret_cast<const T*>(g.b);
  T* out = reinterpret_cast<T*>(g.out);
  // constants and both operands are loaded without branches (a `ptr ? load : c` is a branch + wait per load; an `if (b)`
  // around the second operand makes it a second dependent round trip): the small joins were pure latency chains
  float ma[8], sa[8], ba[8], mb[8], sb[8], bb[8];
  const float* safe = reinterpret_cast<const float*>(g.a);     // any readable, 16-byte aligned address
  coef8(g.sa, safe, c0, true, 1.f, sa);
  coef8(g.sa ? g.ma : nullptr, safe, c0, true, 0.f, ma);
  coef8(g.sa ? g.ba : nullptr, safe, c0, true, 0.f, ba);
  coef8(g.sb, safe, c0, true, 1.f, sb);
  coef8(g.sb ? g.mb : nullptr, safe, c0, true, 0.f, mb);
  coef8(g.sb ? g.bb : nullptr, safe, c0, true, 0.f, bb);
  const long stride = (long)gridDim.x * g.NPL;
  const float relu_lo = g.relu ? 0.f : -__builtin_inff();
  const unsigned long long dseed = g.seed_slot ? *g.seed_slot : 0ull;
  const uint32_t dk0 = (uint32_t)dseed, dk1 = (uint32_t)(dseed >> 32) ^ 0x5EEDu;
  const float dinv = 1.f / (1.f - g.drop_p);
  const uint32_t dthresh = (uint32_t)((double)g.drop_p * 4294967296.0);
  const bool hb = b != nullptr;
  const T* b2 = hb ? b : a;                                      // dummy second stream when there is none
  const long ldb2 = hb ? g.ldb : g.lda;
  // JU pixels per trip: all 2*JU loads of a lane are issued before the first dependent instruction.  One pixel per trip left
  // 32 KB in flight per CU (8 waves x 64 lanes x 4 x 16 B) -- 3.2-3.8 TB/s on the 200 MB joins by Little's law alone.
  constexpr int JU = 4;
  for (long p0 = (long)blockIdx.x * g.NPL + pl; p0 < g.P; p0 += stride * JU) {
    typename V8<T>::Raw ra[JU], rb[JU];
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const long p = p0 + u * stride;
      const long q = p < g.P ? p : p0;
      ra[u] = V8<T>::load_raw(a + q * g.lda + c0);
      rb[u] = V8<T>::load_raw(b2 + q * ldb2 + c0);
    }
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const long p = p0 + u * stride;
      if (p >= g.P) break;
      float v[8], uu[8];
      V8<T>::unpack(ra[u], v);
      V8<T>::unpack(rb[u], uu);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (v[j] - ma[j]) * sa[j] + ba[j];
        if (hb) v[j] += (uu[j] - mb[j]) * sb[j] + bb[j];
        v[j] = fmaxf(v[j], relu_lo);
      }
      if (g.seed_slot) {   // same counters / keys as dropout_kernel on the materialised tensor: identical mask, identical bits
        uint32_t r0[4], r1[4];
        const uint64_t ctr = (uint64_t)(p * g.C + c0) >> 2;
        philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), dk0, dk1, r0);
        philox4x32((uint32_t)(ctr + 1), (uint32_t)((ctr + 1) >> 32), dk0, dk1, r1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (r0[j] >= dthresh) ? V8<T>::round(v[j]) * dinv : 0.f;
          v[4 + j] = (r1[j] >= dthresh) ? V8<T>::round(v[4 + j]) * dinv : 0.f;
        }
      }
      V8<T>::store(out + p * g.ldo + c0, v);
    }
  }
}

template <typename T> struct StatAcc { typedef float type; };
template <> struct StatAcc<float> { typedef double type; };

template <typename T>
__global__ __launch_bounds__(NT) void join_bwd_kernel(const JoinArgs g) {
  typedef typename StatAcc<T>::type A;
  __shared__ __align__(16) unsigned char smem[NT * 8 * 3 * sizeof(double)];
  const int tid = threadIdx.x;
  const int cg = tid % g.CV, pl = tid / g.CV;
  const bool active = pl < g.NPL;
  const int c0 = cg * 8;
  const T* dout = reinterpret_cast<const T*>(g.dout);
  const T* out = reinterpret_cast<const T*>(g.out);
  const T* a = reinterpret_cast<const T*>(g.a);
  const T* b = reinterpret_cast<const T*>(g.b);
  T* e = reinterpret_cast<T*>(g.e);
  A s0[8], sA[8], sB[8];
  float ma[8], mb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0; sA[j] = 0; sB[j] = 0; }
  const float* safe = reinterpret_cast<const float*>(g.dout);
  coef8(g.ma, safe, c0, active, 0.f, ma);
  coef8(g.mb, safe, c0, active, 0.f, mb);
  const long stride = (long)gridDim.x * g.NPL;
  if (active) {
    // up to four streams per pixel (dout, out for the ReLU mask, the two raw branch outputs for the statistics): all
    // issued together -- absent ones re-read dout -- instead of four dependent round trips
    const bool hr = g.relu != 0, ha = g.stats_a != nullptr, hbb = g.stats_b != nullptr;
    const T* po = hr ? out : dout; const long ldo = hr ? g.ldo : g.lddo;
    const T* pa = ha ? a : dout;   const long lda = ha ? g.lda : g.lddo;
    const T* pb = hbb ? b : dout;  const long ldb = hbb ? g.ldb : g.lddo;
    constexpr int JU = sizeof(T) == 2 ? 4 : 2;   // pixels per trip, all loads of a lane issued first (see join_fwd_kernel)
    for (long p0 = (long)blockIdx.x * g.NPL + pl; p0 < g.P; p0 += stride * JU) {
      typename V8<T>::Raw rd[JU], ro[JU], rA[JU], rB[JU];
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const long p = p0 + u * stride;
        const long q = p < g.P ? p : p0;
        rd[u] = V8<T>::load_raw(dout + q * g.lddo + c0);
        ro[u] = V8<T>::load_raw(po + q * ldo + c0);
        rA[u] = V8<T>::load_raw(pa + q * lda + c0);
        rB[u] = V8<T>::load_raw(pb + q * ldb + c0);
      }
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const long p = p0 + u * stride;
        if (p >= g.P) break;
        float v[8], o[8], ua[8], ub[8];
        V8<T>::unpack(rd[u], v); V8<T>::unpack(ro[u], o); V8<T>::unpack(rA[u], ua); V8<T>::unpack(rB[u], ub);
        if (g.dscale != 1.f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = V8<T>::round(v[j] * g.dscale);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (hr && !(o[j] > 0.f)) v[j] = 0.f;
        if (e) V8<T>::store(e + p * g.lde + c0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s0[j] += (A)v[j];
          if (ha) sA[j] += (A)v[j] * (A)(ua[j] - ma[j]);
          if (hbb) sB[j] += (A)v[j] * (A)(ub[j] - mb[j]);
        }
      }
    }
  }
  if (!g.stats_a && !g.stats_b) return;
  // block reduce: red[pl][3][C]
  A* red = reinterpret_cast<A*>(smem);
  const int C = g.C;
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(pl * 3 + 0) * C + c0 + j] = s0[j];
      red[(pl * 3 + 1) * C + c0 + j] = sA[j];
      red[(pl * 3 + 2) * C + c0 + j] = sB[j];
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * C; i += blockDim.x) {
    const int which = i / C, c = i - which * C;
    double s = 0.0;
    for (int q = 0; q < g.NPL; ++q) s += (double)red[(q * 3 + which) * C + c];
    const long row = (long)blockIdx.x * 2 * C;
    if (which == 0) {
      if (g.stats_a) g.stats_a[row + c] = s;
      if (g.stats_b) g.stats_b[row + c] = s;
    } else if (which == 1) {
      if (g.stats_a) g.stats_a[row + C + c] = s;
    } else {
      if (g.stats_b) g.stats_b[row + C + c] = s;
    }
    for (int r = blockIdx.x + gridDim.x; r < TSS_STAT_SLABS; r += gridDim.x) {  // rows nobody owns
      const long z = (long)r * 2 * C + (which == 0 ? c : C + c);
      if (g.stats_a && which != 2) g.stats_a[z] = 0.0;
      if (g.stats_b && which != 1) g.stats_b[z] = 0.0;
    }
  }
}

int join_geometry(JoinArgs& g, int* threads, int* grid, long max_blocks = TSS_STAT_SLABS) {
  if (g.C <= 0 || (g.C % 8) != 0 || g.C > NT * 8) return TSS_ERR_SHAPE;
  g.CV = g.C / 8;
  g.NPL = NT / g.CV;
  *threads = (g.CV * g.NPL + 63) / 64 * 64;
  const long tiles = (g.P + g.NPL - 1) / g.NPL;
  long gsz = tiles < max_blocks ? tiles : max_blocks;          // backward: one statistics slab row per block
  if (gsz < 1) gsz = 1;
  *grid = (int)gsz;
  return TSS_OK;
}

// ------------------------------------------------------------------------------------------ dropout
// y = x * keep / (1 - p); the same kernel serves backward (x := grad).  One Philox call per 4 elements,
// keyed by (seed slot, logical element index / 4) so the mask is independent of the launch geometry.
template <typename T>
__global__ __launch_bounds__(NT) void dropout_kernel(const T* x, long ldx, T* y, long ldy, long P, int C,
                                                     float p, const unsigned long long* seed_slot) {
  const unsigned long long seed = *seed_slot;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 0x5EEDu;
  const float inv = 1.f / (1.f - p);
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const int CV = C / 8;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / CV;
    const int cv = (int)(i - pix * CV);
    float v[8];
    V8<T>::load(x + pix * ldx + cv * 8, v);
    uint32_t r0[4], r1[4];
    const uint64_t ctr = (uint64_t)(pix * C + cv * 8) >> 2;
    philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), k0, k1, r0);
    philox4x32((uint32_t)(ctr + 1), (uint32_t)((ctr + 1) >> 32), k0, k1, r1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (r0[j] >= thresh) ? v[j] * inv : 0.f;
      v[4 + j] = (r1[j] >= thresh) ? v[4 + j] * inv : 0.f;
    }
    V8<T>::store(y + pix * ldy + cv * 8, v);
  }
}

// The mask alone, for consumers that apply nn.Dropout on load (tss_pwconv_fwd_drop / tss_pwconv_bwd_fused_drop): one BYTE per
// (pixel, 8-channel vector), bit j = channel 8 v + j is kept; a pixel's bytes form one 16-byte row (<= 128 channels), so a lane that
// needs the mask of every channel of a pixel gets it with ONE aligned load.  One Philox call per byte, 16 random bits per element (kept <=>
// bits >= round(p * 65536): the probability is exact to 2^-16), keyed by the device-side counter -- which this kernel only READS
// (every block does; the consumer advances it, behind the kernel boundary, so no block can see the next step's value).
__global__ __launch_bounds__(NT) void dropout_mask_kernel(const unsigned long long* counter, uint32_t* mask4, long nwords, float p) {
  const unsigned long long seed = *counter;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ 0x5EEDu;
  const uint32_t thresh = (uint32_t)((double)p * 65536.0 + 0.5);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x) {
    uint32_t word = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t ctr = (uint64_t)i * 4 + u;
      uint32_t r[4];
      philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), k0, k1, r);
      uint32_t byte = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) byte |= (((r[j >> 1] >> (16 * (j & 1))) & 0xffffu) >= thresh ? 1u : 0u) << j;
      word |= byte << (8 * u);
    }
    mask4[i] = word;
  }
}

// ------------------------------------------------------------------------------------------ misc
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* e, long lde, long P, int N, float* out) {
  // out[n] += sum_p e[p][n]; threads = (pixel lane, channel), N <= 64
  __shared__ float red[NT];
  const int nl = threadIdx.x % 64, pl = threadIdx.x / 64;
  float s = 0.f;
  if (nl < N) {
    const long stride = (long)gridDim.x * 4;
    for (long p = (long)blockIdx.x * 4 + pl; p < P; p += stride * 8) {
      T v[8];   // 8 rows in flight per lane (clamped row + select: no predicated loads)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long q = p + u * stride;
        v[u] = e[(q < P ? q : p) * lde + nl];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (p + u * stride < P) ? (float)v[u] : 0.f;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 64 && nl < N) atomicAdd(out + nl, red[nl] + red[64 + nl] + red[128 + nl] + red[192 + nl]);
}

__global__ void permute_w3x3_kernel(const float* w, float* w_tnc, float* w_tcn, int N, int Cin) {
  const long total = (long)N * Cin * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long nc = i / 9;
    const int c = (int)(nc % Cin), n = (int)(nc / Cin);
    const float v = w[i];
    if (w_tnc) w_tnc[((long)tap * N + n) * Cin + c] = v;
    if (w_tcn) w_tcn[((long)tap * Cin + c) * N + n] = v;
  }
}

__global__ void permute_wtaps_kernel(const float* w, float* w_tnc, float* w_tcn, int N, int Cin, int T) {
  const long total = (long)N * Cin * T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % T);
    const long nc = i / T;
    const int c = (int)(nc % Cin), n = (int)(nc / Cin);
    const float v = w[i];
    if (w_tnc) w_tnc[((long)tap * N + n) * Cin + c] = v;
    if (w_tcn) w_tcn[((long)tap * Cin + c) * N + n] = v;
  }
}

// channel_shuffle(x, groups) of TSS/models/lednet.py:183-188: out[:, j * groups + i] = x[:, i * (C / groups) + j].  One lane
// = 8 consecutive OUTPUT channels of one pixel (a 16-byte store); its 8 sources are gathered from the same pixel row.
template <typename T>
__global__ __launch_bounds__(NT) void channel_shuffle_kernel(const T* x, long ldx, T* y, long ldy, long P, int C, int groups) {
  const int CV = C / 8, cpg = C / groups;
  const long total = P * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / CV;
    const int cv = (int)(i - pix * CV);
    const T* row = x + pix * ldx;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = cv * 8 + j;                       // output channel = jj * groups + ii
      const int ii = co % groups, jj = co / groups;
      v[j] = (float)row[ii * cpg + jj];
    }
    V8<T>::store(y + pix * ldy + cv * 8, v);
  }
}

// ------------------------------------------------------------------------------------------ AdamW
// state = {step, bias_correction1, bias_correction2_sqrt}; torch.optim.AdamW arithmetic, single tensor.
__global__ void adamw_tick_kernel(float* state, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float step = state[0] + 1.f;
    state[0] = step;
    state[1] = 1.f - powf(beta1, step);
    state[2] = sqrtf(1.f - powf(beta2, step));
  }
}

__global__ __launch_bounds__(NT) void adamw_kernel(float* p, const float* g, float* m, float* v, long n,
                                                   const float* lr_ptr, float beta1, float beta2, float eps,
                                                   float weight_decay, const float* state, float grad_scale,
                                                   float lr_host, float bc1_host, float bc2s_host) {
  // device-side step counter / learning rate (state, lr_ptr: the step can be replayed from a captured graph), or both as
  // kernel arguments computed on the host (one launch instead of three when the optimizer step is launched eagerly)
  const float lr = state ? *lr_ptr : lr_host;
  const float bc1 = state ? state[1] : bc1_host, bc2s = state ? state[2] : bc2s_host;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * weight_decay);
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

// bf16 shadows of the 1x1 convolution weights, all layers in one launch: job j = (f32 source [N][K], bf16 copy [N][K],
// bf16 transpose [K][N]).  The pointwise kernels stage their weight tiles from these with plain 16-byte copies
// (forward reads the copy, backward-data the transpose) instead of converting f32 per block.
// blockIdx.y == njobs (when zero_n > 0): these blocks clear a float buffer instead -- the flat gradient buffer of the step, whose
// zero-fill would otherwise be a launch of its own at the same point of the step (16-byte stores; zero_n % 4 == 0 or a scalar tail)
__global__ __launch_bounds__(256) void cast_weights_kernel(const long long* table, int njobs, float* zero, long zero_n) {
  const int job = blockIdx.y;
  if (job >= njobs) {
    const long n4 = zero_n >> 2;
    float4* z4 = reinterpret_cast<float4*>(zero);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && (long)threadIdx.x < zero_n - n4 * 4) zero[n4 * 4 + threadIdx.x] = 0.f;
    return;
  }
  const long long* t = table + (long)job * 5;
  const float* src = reinterpret_cast<const float*>(t[0]);
  bf16_t* dst = reinterpret_cast<bf16_t*>(t[1]);
  bf16_t* dstT = reinterpret_cast<bf16_t*>(t[2]);
  const long N = t[3], K = t[4];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < N * K; e += (long)gridDim.x * 256) {
    const bf16_t v = (bf16_t)src[e];
    const long n = e / K, k = e - n * K;
    dst[e] = v;
    dstT[k * N + n] = v;
  }
}

}  // namespace

extern "C" {

int tss_bn_finalize(const double* sums, double count, const float* gamma, float eps,
                    float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* mean_out, float* invstd_out, float* scale, int C, void* stream) {
  TSS_REQUIRE(C > 0 && count >= 1.0, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_FINALIZE, (hipStream_t)stream, 40.0 * C, 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, (hipStream_t)stream, sums, count,
                     gamma, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out, invstd_out,
                     scale, C);
  return tss::check_last("bn_finalize");
}

int tss_slab_reduce(const double* slabs, double count, double* out, int C, void* stream) {
  TSS_REQUIRE(C > 0 && count >= 0.0, TSS_ERR_SHAPE);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, (hipStream_t)stream, slabs, count, out, C);
  return tss::check_last("slab_reduce");
}

int tss_bn_finalize_sync(const double* gsums, const float* gamma, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* mean_out, float* invstd_out,
                         float* scale, int C, void* stream) {
  TSS_REQUIRE(C > 0, TSS_ERR_SHAPE);
  hipLaunchKernelGGL(bn_finalize_sync_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gsums, gamma, eps,
                     momentum, running_mean, running_var, num_batches_tracked, mean_out, invstd_out, scale, C);
  return tss::check_last("bn_finalize_sync");
}

int tss_bn_bwd_finalize_sync(const double* bstats, const double* gsums, const float* invstd, const float* gamma,
                             int accumulate, float* dgamma, float* dbeta, float* ga, float* gb, float* gce, int C,
                             void* stream) {
  TSS_REQUIRE(C > 0, TSS_ERR_SHAPE);
  hipLaunchKernelGGL(bn_bwd_finalize_sync_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_NT), 0, (hipStream_t)stream,
                     bstats, gsums, invstd, gamma, accumulate, dgamma, dbeta, ga, gb, gce, C);
  return tss::check_last("bn_bwd_finalize_sync");
}

int tss_bn_eval_affine(const float* gamma, const float* running_mean, const float* running_var,
                       float eps, float* mean_out, float* invstd_out, float* scale, int C, void* stream) {
  TSS_REQUIRE(C > 0, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_FINALIZE, (hipStream_t)stream, 32.0 * C, 0);
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma,
                     running_mean, running_var, eps, mean_out, invstd_out, scale, C);
  return tss::check_last("bn_eval_affine");
}

int tss_bn_eval_affine_batched(const long long* table, int njobs, int max_channels, void* stream) {
  TSS_REQUIRE(table && njobs > 0 && max_channels > 0, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_FINALIZE, (hipStream_t)stream, 24.0 * max_channels * njobs, 0);
  hipLaunchKernelGGL(bn_eval_affine_batched_kernel, dim3((max_channels + 127) / 128, njobs), dim3(128), 0,
                     (hipStream_t)stream, table, njobs);
  return tss::check_last("bn_eval_affine_batched");
}

int tss_bn_bwd_finalize(const double* bstats, double count, const float* invstd,
                        const float* gamma, int training, int accumulate, float* dgamma, float* dbeta,
                        float* ga, float* gb, float* gce, int C, void* stream) {
  TSS_REQUIRE(C > 0 && count >= 1.0, TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_BWD_FINALIZE, (hipStream_t)stream, 48.0 * C, 0);
  tss_bn_bwd_job j = {bstats, count, invstd, gamma, training, accumulate, dgamma, dbeta, ga, gb, gce, C};      // (xchg_world = 0: local statistics)
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(fin_blocks(C)), dim3(FIN_NT), 0, (hipStream_t)stream, j);
  return tss::check_last("bn_bwd_finalize");
}

int tss_join_fwd(const void* a, long lda, const float* ma, const float* sa, const float* ba,
                 const void* b, long ldb, const float* mb, const float* sb, const float* bb,
                 void* out, long ldo, int relu, float drop_p, const unsigned long long* seed_slot,
                 long P, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((lda % 8) == 0 && lda >= C && (ldo % 8) == 0 && ldo >= C && (!b || ((ldb % 8) == 0 && ldb >= C)), TSS_ERR_SHAPE);
  TSS_REQUIRE(!seed_slot || (relu && drop_p > 0.f && drop_p < 1.f), TSS_ERR_SHAPE);   // the backward relies on out > 0 <=> kept
  TSS_REQUIRE(tss::aligned16(a) && tss::aligned16(out) && tss::aligned16(b), TSS_ERR_ALIGN);
  JoinArgs g = {};
  g.a = a; g.lda = lda; g.ma = ma; g.sa = sa; g.ba = ba; g.b = b; g.ldb = ldb; g.mb = mb; g.sb = sb; g.bb = bb;
  g.out = out; g.ldo = ldo; g.relu = relu; g.P = P; g.C = C;
  g.drop_p = seed_slot ? drop_p : 0.f; g.seed_slot = seed_slot; g.dscale = 1.f;
  int threads, grid;
  // the forward writes no slab rows, so its grid is not tied to their count -- but more blocks are not faster (A/B in round 2:
  // 512 / 1024 / 2048 / 4096 blocks -> 6.315 / 6.330 / 6.340 / 6.352 ms per step)
  static const long fwd_blocks = getenv("TSS_JOIN_FWD_BLOCKS") ? atol(getenv("TSS_JOIN_FWD_BLOCKS")) : 512;
  const int rc = join_geometry(g, &threads, &grid, fwd_blocks);
  if (rc) return rc;
  if (P == 0) return TSS_OK;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, (double)P * C * (b ? 3 : 2) * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(join_fwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(join_fwd_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  return tss::check_last("join_fwd");
}

int tss_join_bwd(const void* dout, long lddo, const void* out, long ldo, int relu,
                 const void* a_raw, long lda, const float* mean_a, double* stats_a,
                 const void* b_raw, long ldb, const float* mean_b, double* stats_b,
                 void* e, long lde, float dout_scale, long P, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE((lddo % 8) == 0 && lddo >= C && (!relu || (out && (ldo % 8) == 0 && ldo >= C)), TSS_ERR_SHAPE);
  TSS_REQUIRE(dout_scale > 0.f && (dout_scale == 1.f || relu), TSS_ERR_SHAPE);
  TSS_REQUIRE((!stats_a || (a_raw && (lda % 8) == 0 && lda >= C)) && (!stats_b || (b_raw && (ldb % 8) == 0 && ldb >= C)), TSS_ERR_SHAPE);
  TSS_REQUIRE(!e || ((lde % 8) == 0 && lde >= C), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dout) && tss::aligned16(e), TSS_ERR_ALIGN);
  JoinArgs g = {};
  g.dout = dout; g.lddo = lddo; g.out = const_cast<void*>(out); g.ldo = ldo; g.relu = relu;
  g.a = a_raw; g.lda = lda; g.ma = mean_a; g.stats_a = stats_a; g.b = b_raw; g.ldb = ldb; g.mb = mean_b; g.stats_b = stats_b;
  g.e = e; g.lde = lde; g.P = P; g.C = C; g.dscale = dout_scale;
  int threads, grid;
  const int rc = join_geometry(g, &threads, &grid);
  if (rc) return rc;
  if (P == 0) return TSS_OK;
  const int nt = 1 + (relu ? 1 : 0) + (e ? 1 : 0) + (stats_a ? 1 : 0) + (stats_b ? 1 : 0);
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)P * C * nt * esz(dtype), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(join_bwd_kernel<bf16_t>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(join_bwd_kernel<float>, dim3(grid), dim3(threads), 0, (hipStream_t)stream, g);
  return tss::check_last("join_bwd");
}

int tss_stat_slabs(void) { return TSS_STAT_SLABS; }

int tss_dropout_tick(unsigned long long* counter, unsigned long long* seed_slot, void* stream) {
  hipLaunchKernelGGL(dropout_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, seed_slot);
  return tss::check_last("dropout_tick");
}

int tss_dropout_mask(const unsigned long long* counter, void* mask, long P, int C, float p, void* stream) {
  TSS_REQUIRE(counter && mask && C > 0 && (C % 8) == 0 && p > 0.f && p < 1.f && tss::aligned16(mask), TSS_ERR_SHAPE);
  if (P == 0) return TSS_OK;
  TSS_REQUIRE(C <= 128, TSS_ERR_SHAPE);
  const long nwords = P * 4;                         // [P][16] bytes
  long grid = (nwords + NT - 1) / NT;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, counter, (uint32_t*)mask, nwords, p);
  return tss::check_last("dropout_mask");
}

int tss_dropout(const void* x, long ldx, void* y, long ldy, long P, int C, float p,
                const unsigned long long* seed_slot, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && (ldx % 8) == 0 && (ldy % 8) == 0 && ldx >= C && ldy >= C && p >= 0.f && p < 1.f, TSS_ERR_SHAPE);
  if (P == 0) return TSS_OK;
  const long total = P * (C / 8);
  long grid = (total + NT - 1) / NT;
  if (grid > 2048) grid = 2048;
  tss::ProfScope prof(TSS_K_DROPOUT, (hipStream_t)stream, 2.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (bf16_t*)y, ldy, P, C, p, seed_slot);
  else
    hipLaunchKernelGGL(dropout_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)x, ldx, (float*)y, ldy, P, C, p, seed_slot);
  return tss::check_last("dropout");
}

int tss_bias_grad(const void* e, long lde, long P, int N, float* dbias, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(N > 0 && N <= 64 && lde >= N, TSS_ERR_SHAPE);
  if (P == 0) return TSS_OK;
  long grid = (P + 3) / 4;
  if (grid > 1024) grid = 1024;
  tss::ProfScope prof(TSS_K_BIAS_GRAD, (hipStream_t)stream, (double)P * N * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)e, lde, P, N, dbias);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const float*)e, lde, P, N, dbias);
  return tss::check_last("bias_grad");
}

int tss_permute_w3x3(const float* w, float* w_tnc, float* w_tcn, int N, int Cin, void* stream) {
  TSS_REQUIRE(N > 0 && Cin > 0, TSS_ERR_SHAPE);
  const long total = (long)N * Cin * 9;
  long grid = (total + NT - 1) / NT;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(permute_w3x3_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, w, w_tnc, w_tcn, N, Cin);
  return tss::check_last("permute_w3x3");
}

int tss_permute_wtaps(const float* w, float* w_tnc, float* w_tcn, int N, int Cin, int T, void* stream) {
  TSS_REQUIRE(N > 0 && Cin > 0 && T > 0, TSS_ERR_SHAPE);
  const long total = (long)N * Cin * T;
  long grid = (total + NT - 1) / NT;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(permute_wtaps_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, w, w_tnc, w_tcn, N, Cin, T);
  return tss::check_last("permute_wtaps");
}

int tss_channel_shuffle(const void* x, long ldx, void* y, long ldy, long P, int C, int groups, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 8) == 0 && groups > 0 && (C % groups) == 0 && ldx >= C && (ldy % 8) == 0 && ldy >= C, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(y), TSS_ERR_ALIGN);
  if (P == 0) return TSS_OK;
  const long total = P * (C / 8);
  long grid = (total + NT - 1) / NT;
  if (grid > 2048) grid = 2048;
  tss::ProfScope prof(TSS_K_COPY, (hipStream_t)stream, 2.0 * P * C * esz(dtype), 0);
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(channel_shuffle_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, P, C, groups);
  else
    hipLaunchKernelGGL(channel_shuffle_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, P, C, groups);
  return tss::check_last("channel_shuffle");
}

int tss_cast_weights(const long long* table, int njobs, int blocks_per_job, float* zero, long zero_n, void* stream) {
  TSS_REQUIRE(njobs >= 0 && blocks_per_job >= 1 && zero_n >= 0 && (zero_n == 0 || (zero && tss::aligned16(zero))), TSS_ERR_SHAPE);
  if (njobs == 0 && zero_n == 0) return TSS_OK;
  hipLaunchKernelGGL(cast_weights_kernel, dim3(blocks_per_job, njobs + (zero_n > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, table,
                     njobs, zero, zero_n);
  return tss::check_last("cast_weights");
}

int tss_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                   const float* lr, float beta1, float beta2, float eps, float weight_decay,
                   float* state /*[3]: step, bc1, sqrt(bc2)*/, float grad_scale, float lr_host, long step_host, void* stream) {
  TSS_REQUIRE(n >= 0 && (state ? lr != nullptr : step_host >= 1), TSS_ERR_SHAPE);
  float bc1 = 1.f, bc2s = 1.f;
  if (state) hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, beta1, beta2);
  else { bc1 = 1.f - powf(beta1, (float)step_host); bc2s = sqrtf(1.f - powf(beta2, (float)step_host)); }   // as adamw_tick_kernel
  if (n == 0) return tss::check_last("adamw_tick");
  long grid = (n + NT - 1) / NT;
  if (grid > 2048) grid = 2048;
  tss::ProfScope prof(TSS_K_ADAMW, (hipStream_t)stream, 28.0 * n, 12.0 * n);
  hipLaunchKernelGGL(adamw_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state, grad_scale, lr_host, bc1, bc2s);
  return tss::check_last("adamw");
}

}  // extern "C"

// Lean bf16 pointwise (1x1) convolution kernel for the performance path: forward and backward-data of every 1x1
// layer whose contraction fits one LDS chunk (K <= 128 channels; fwd: Cin, bwd: Cout).
//
// Same tiling as convgemm_kernel (128 pixels x 128 channels per block, 4 waves x 64x64, D = W x A^T so a lane owns 4
// consecutive channels of one pixel, weights resident in LDS, statistics to slab rows).  What is different is the
// VALU budget: rocprofv3 counters on the generic kernel (profiles/README.md) show ~1900 vector instructions per
// wave per tile, most of them 64-bit address arithmetic, per-vector bounds logic and spilled-SGPR traffic from the
// three operand modes.  Here
//   * the channel vector of a lane is FIXED (lane = (row-in-pass, 8-channel vector)): its BatchNorm coefficients
//     live in registers with the mean folded into the bias ((x-mu)*s+b == x*s + (b-mu*s); bf16 activations only),
//     and its global pointer advances by one constant per pass;
//   * full tiles take a path with no bounds checks at all (only the last tile of a layer is ragged);
//   * the epilogue derives the statistics from the bits it is about to store and writes through 4 row pointers.
// The f32 parity path, K > 128, dense 3x3 and ragged channel counts stay on convgemm_kernel.
#include <cstdlib>
#include "common.h"
#include "wgreduce.h"

namespace {

constexpr int BM = 128, NCH = 128, NT = 256, KMAX = 128, RS = KMAX + 8;

#ifdef TSS_TIMING
__device__ unsigned long long g_pw_timing[8];   // debug builds: [prologue, loop, tail, ..., blocks] cycles of wave 0
#define TSS_T(var) unsigned long long var; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define TSS_T(var)
#endif
typedef bf16_t T;

struct FastArgs {
  long P; int K, N;                    // K = contraction channels, N = output channels
  const T* a0; long lda0; const T* a1; long lda1;
  const float* c0; const float* c1; const float* c2; const float* c3;   // see convgemm.hip GemmArgs
  int a_relu;
  const float* w; int w_trans;         // 0: w[n*K + k]   1: w[k*N + n]
  const T* wb; long ldwb;              // optional bf16 shadow laid out [output channel][contraction] (tss_cast_weights)
  const float* bias;
  T* y; long ldy; double* stats;
  const T* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
  const T* radd; long ldr;              // bwd, input already materialised (no xm): e_in = acc + radd (the skip gradient of a residual block)
  int gslots;
  tss_wg::ReduceArgs red;               // bwd: pending weight-gradient slot reduction of the same layer (nred == 0: none)
  int nred8;                            // its block count rounded up to 8 (keeps blockIdx % 8 == XCD for the main blocks)
  // forward, multi-chunk kernel only: the contraction is the channel CONCATENATION of nsrc tensors of 128 channels each (chunk kc
  // reads source kc; a pitch of 0 broadcasts one row to every pixel), each with its own pending BatchNorm -- the concat never exists
  int nsrc;
  const T* asrc[6]; long ldsrc[6];
  const float* c0s[6]; const float* c1s[6]; const float* c2s[6];
  // forward, DROP instances: nn.Dropout between the pending BatchNorm + ReLU and this layer, applied on load.  dmask: [P][16] bytes, one
  // per (pixel, 8-channel vector), drawn by tss_dropout_mask (bit j = channel 8 v + j kept); dinv = 1 / (1 - p); dcounter: the device-side
  // Philox counter the mask was keyed by -- advanced here, by the consumer, because every block of the mask kernel reads it
  const unsigned char* dmask; float dinv; unsigned long long* dcounter;
  // forward, eval mode (frozen statistics): the BatchNorm BEHIND the layer, the skip of a residual block and the ReLU after the sum are
  // applied in the epilogue -- y = relu?((acc + bias - emean) * escale + ebeta + radd) -- so the block output is written by the layer
  // itself and no join pass exists (escale == NULL: plain output).  radd / ldr as in backward.
  const float* emean; const float* escale; const float* ebeta; int erelu;
  // backward, materialised input that is the OUTPUT of a relu join (the previous block's relu(BN(y3) + skip)): that join's backward --
  // ReLU mask from its output jout, the BatchNorm-backward sums of e and e * (jy - jmean) as slab rows in `stats` -- runs in this
  // epilogue, on the complete gradient (acc + radd), instead of in a pass of its own (ops.JoinFn.backward then only forwards e_in)
  const T* jout; long ldjo; const T* jy; long ldjy; const float* jmean;
};

__device__ __forceinline__ float bits_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bits_lo(uint32_t u) { return __uint_as_float(u << 16); }

// Weight chunk -> LDS as bf16: Ws[n][j] for n in [0, nrows), j in [0, kwp); zero outside [0, ncw) x [0, kw).
//   w_trans == 0: source w[(n0+n)*ldw + kb + j]   (forward: rows are output channels, ldw = K)
//   w_trans == 1: source w[(kb+j)*ldw + n0 + n]   (backward-data: the same [N][K] tensor read transposed, ldw = conv K)
// Fixed 2-D thread maps (no integer division), 8 independent loads in flight per thread, clamp + select instead of
// predicated loads (a predicated load compiles to a branch with its own s_waitcnt: 16 serial L2 round trips).
__device__ __forceinline__ void stage_weights(T* Ws, const float* w, int w_trans, long ldw, int n0, int ncw, int nrows,
                                              int kb, int kw, int kwp, int tid) {
  if (!w_trans) {
    const int tx = tid & 31, ty = tid >> 5;   // float4 column, row within a pass of 8
    const bool cok = tx * 4 < kw;
    for (int rb = 0; rb < nrows; rb += 64) {
      float4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = rb + u * 8 + ty;
        const bool ok = cok && n < ncw;
        wv[u] = *reinterpret_cast<const float4*>(w + (ok ? (long)(n0 + n) * ldw + kb + tx * 4 : 0));
        if (!ok) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = rb + u * 8 + ty;
        if (n < nrows && tx * 4 < kwp) {
          bf16x4 o; o[0] = (T)wv[u].x; o[1] = (T)wv[u].y; o[2] = (T)wv[u].z; o[3] = (T)wv[u].w;
          *reinterpret_cast<bf16x4*>(Ws + n * RS + tx * 4) = o;
        }
      }
    }
  } else {
    const int tj = tid & 127, th = tid >> 7;  // contraction index (conflict-free LDS stores), n-vector within a pass of 2
    const bool jok = tj < kw;
    const int nvt = nrows >> 2;
    for (int vb = 0; vb < nvt; vb += 16) {
      float4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int nv = vb + u * 2 + th;
        const bool ok = jok && nv * 4 < ncw;
        wv[u] = *reinterpret_cast<const float4*>(w + (ok ? (long)(kb + tj) * ldw + n0 + nv * 4 : 0));
        if (!ok) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int nv = vb + u * 2 + th;
        if (nv < nvt && tj < kwp) {
          T* d = Ws + (nv * 4) * RS + tj;
          d[0] = (T)wv[u].x; d[RS] = (T)wv[u].y; d[2 * RS] = (T)wv[u].z; d[3 * RS] = (T)wv[u].w;
        }
      }
    }
  }
}

// Weight chunk from a bf16 shadow wb[(n0+n)*ldwb + kb + j] (already [output][contraction]): plain 16-byte copies,
// one round trip (<= 8 loads per thread for a 128 x 128 chunk), zero outside [0, ncw) x [0, kw).
__device__ __forceinline__ void stage_weights_bf16(T* Ws, const T* wb, long ldwb, int n0, int ncw, int nrows,
                                                   int kb, int kw, int kwp, int tid) {
  const int tx = tid & 15, ty = tid >> 4;     // 16-byte vector within the row, row within a pass of 16
  const bool cok = tx * 8 < kw;               // kw % 8 == 0 (host-checked)
  uint4 wv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int n = u * 16 + ty;
    const bool ok = cok && n < ncw;
    wv[u] = *reinterpret_cast<const uint4*>(wb + (ok ? (long)(n0 + n) * ldwb + kb + tx * 8 : 0));
    if (!ok) wv[u] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int n = u * 16 + ty;
    if (n < nrows && tx * 8 < kwp) *reinterpret_cast<uint4*>(Ws + n * RS + tx * 8) = wv[u];
  }
}

// TM = pixels per tile.  Forward: 128.  Backward-data: 64 -- it stages two tensors (e, y) and needs the producer's raw
// output in the epilogue; with 128-pixel tiles those registers leave no room to keep the next tile's loads in flight
// (an attempt spilled 188 B/lane and lost), with 64-pixel tiles everything is prefetched and nothing spills.
template <bool BWD, int TM, bool DROP = false, bool JB = false>
__global__ __launch_bounds__(NT, 2) void pwfast_kernel(const FastArgs g) {
  static_assert(!JB || BWD, "the join-backward epilogue belongs to backward-data");
  static_assert(!(BWD && DROP), "dropout on load is a forward prologue");
  constexpr int MF = TM / 32;    // 16-pixel MFMA fragments per wave (two waves split the tile's pixels)
  constexpr int NP = TM / 16;    // staging passes (>= 16 rows per pass)
  TSS_T(tq0);
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Ws = Xs + TM * RS;
  float* Ec = reinterpret_cast<float*>(Ws + NCH * RS);   // [3][NCH] epilogue constants of this block's channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // the first nred8 blocks of a backward-data grid sum the weight-gradient workspace of the same layer (wgreduce.h)
  if (BWD && g.nred8 > 0 && (int)blockIdx.x < g.nred8) {
    if ((int)blockIdx.x < g.red.nred) tss_wg::reduce_block(g.red, blockIdx.x, reinterpret_cast<float4*>(smem));
    return;
  }
  const int bidx = BWD ? (int)blockIdx.x - g.nred8 : (int)blockIdx.x;
  if (DROP && blockIdx.x == 0 && threadIdx.x == 0) *g.dcounter += 1ull;     // the mask of this step is drawn: next step, next mask

  const int nchunks = (g.N + NCH - 1) / NCH;
  const int xcd = bidx & 7, slot = bidx >> 3;
  const int nc = slot % nchunks, gslot = slot / nchunks;
  const long ntiles = (g.P + TM - 1) / TM;
  const long per = (ntiles + 7) >> 3;
  const long t_begin = xcd * per + gslot;
  long t_end = xcd * per + per;
  if (t_end > ntiles) t_end = ntiles;

  const int n0 = nc * NCH;
  const int ncw = (g.N - n0 < NCH) ? (g.N - n0) : NCH;
  const int nrows = ((ncw + 15) >> 4) * 16;
  int nfr = (ncw - wn * 64 + 15) >> 4;
  nfr = nfr < 0 ? 0 : (nfr > 4 ? 4 : nfr);

  // The epilogue constants of this block's channels are REQUESTED here and parked in LDS only at the end of the set-up (round 4): their
  // round trip used to be exposed in front of everything else -- loads, wait, LDS stores, barrier, and only then the first tile's
  // requests (2.1 - 2.4 k cycles of the 8 - 10 k-cycle prologue, tools/pw_timing.sh).  Now they fly together with the first tile and
  // the weights: one round trip instead of two.  Three independent loads (null-safe pointers + select), not three branches with a wait each.
  float ec0 = 0.f, ec1 = 0.f, ec2 = 0.f, ec_sc = 0.f, ec_mu = 0.f, ec_be = 0.f;
  bool ec_in = false, ec_hm = false, ec_hs = false, ec_hb = false;
  if (tid < NCH) {
    const int n = n0 + tid;
    const bool in = n < g.N;
    const int nn = in ? n : 0;
    const bool hj = JB && !g.xm && g.jout && g.jmean;
    const bool hm = BWD ? ((g.xm && g.mm) || hj) : (g.bias != nullptr), hs = BWD && g.xm && g.ms, hb = BWD && g.xm && g.mb;
    ec0 = (hm ? (BWD ? (hj ? g.jmean : g.mm) : g.bias) : g.w)[hm ? nn : 0];
    ec1 = (hs ? g.ms : g.w)[hs ? nn : 0];
    ec2 = (hb ? g.mb : g.w)[hb ? nn : 0];
    ec_in = in; ec_hm = hm; ec_hs = hs; ec_hb = hb;
    if (!BWD && g.escale) { ec_sc = g.escale[nn]; ec_mu = g.emean ? g.emean[nn] : 0.f; ec_be = g.ebeta ? g.ebeta[nn] : 0.f; }
  }
  const int K = g.K;
  const int kwp = (K + 31) & ~31;          // MFMA k-steps cover kwp columns; columns >= K are zero in both tiles
  const int nvec = K >> 3;                 // 16-byte vectors per pixel row
  const int nvecp = kwp >> 3;
  const int rpp = NT / nvecp;              // rows staged per pass
  const int npass = (TM + rpp - 1) / rpp;  // <= NP
  const int cv = tid % nvecp, r = tid / nvecp;
  const bool lane_on = r < rpp;            // lanes beyond rpp*nvecp idle during staging
  const bool cv_real = cv < nvec;          // vectors in [nvec, nvecp) are the zero padding of the last k-step

  float st1[4][4], st2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  // epilogue per-lane constants: mask coefficients of the 16 channels this lane owns (4 fragments x 4)
  const int nlane = n0 + wn * 64 + fq * 4;

  // The A-tile loads of tile t+1 are issued right after tile t's registers have been stored to LDS, so the HBM round
  // trip runs under tile t's MFMA + epilogue instead of being exposed at the top of every tile (counters: 76 % of the
  // wave cycles were waits at 2 waves/SIMD).  ra/rb are dead between the LDS store and the next tile: no extra VGPRs.
  uint4 ra[NP], rb[NP];
  uint32_t rm[NP];          // DROP: the mask byte of this lane's vector, per staging pass
  // bwd: the producer's raw output under this lane's outputs (ReLU mask + statistics in the epilogue) travels with the same
  // prefetch, one tile ahead: issued at the top of the tile that needs it, it was 0.5 us old when the epilogue wanted it
  // (LDS store + barrier + 32 MFMAs) against a memory latency of 1.5-2 us -- every tile stalled on it.
  uint2 rxn[4][MF];
  auto issue_loads = [&](long tile_) {
    const long q0 = tile_ * TM;
    const bool full_ = q0 + TM <= g.P;
    if (lane_on) {
      // tile base + this lane's channel vector; rows outside the tile / the tensor re-read row 0 of the tile (always
      // valid) and are zeroed after the load, so no load is ever out of bounds and none is predicated
      const T* pa = g.a0 + q0 * g.lda0 + (cv_real ? cv * 8 : 0);
      const T* pb = BWD ? g.a1 + q0 * g.lda1 + (cv_real ? cv * 8 : 0) : nullptr;
      const int lda = (int)g.lda0, ldb = BWD ? (int)g.lda1 : 0;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        if (ps < npass) {
          const int row = ps * rpp + r;
          const bool ok = row < TM && (full_ || q0 + row < g.P);
          const int rr = ok ? row : 0;
          ra[ps] = *reinterpret_cast<const uint4*>(pa + rr * lda);
          if (BWD) rb[ps] = *reinterpret_cast<const uint4*>(pb + rr * ldb);
          if (DROP) rm[ps] = g.dmask[(q0 + rr) * 16 + (cv_real ? cv : 0)];       // mask rows are 16 bytes per pixel
        }
      }
    }
    if (BWD && g.xm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nfr) {
          const int nn = nlane + i * 16;
          const T* px = g.xm + q0 * g.ldxm + (nn < g.N ? nn : n0);
#pragma unroll
          for (int m = 0; m < MF; ++m) {
            const int row = wm * (TM / 2) + m * 16 + fr;
            rxn[i][m] = *reinterpret_cast<const uint2*>(px + (long)((full_ || q0 + row < g.P) ? row : 0) * g.ldxm);
          }
        }
      }
    }
  };
  TSS_T(tqa);
  if (t_begin < t_end) issue_loads(t_begin);
  TSS_T(tqb);

  // block set-up under the first tile's loads
  // ---- weights -> LDS once per block (resident): Ws[n][k], zero beyond ncw / K
  if (g.wb) stage_weights_bf16(Ws, g.wb, g.ldwb, n0, ncw, nrows, 0, K, kwp, tid);
  else stage_weights(Ws, g.w, g.w_trans, g.w_trans ? (long)g.N : (long)K, n0, ncw, nrows, 0, K, kwp, tid);
  TSS_T(tqc);

  // ---- this lane's prologue coefficients (registers), mean folded into the additive term.  Unconditional 16-byte
  // loads through null-safe pointers: a `ptr ? ptr[i] : c` per element is a branch + wait per load.
  float k0[8], k1[8], kadd[8];
  const float relu_lo = g.a_relu ? 0.f : -TSS_INF;
  {
    const float* safe = g.w;                                   // any readable f32 address for absent vectors
    const int ch = (lane_on && cv_real) ? cv * 8 : 0;
    const float* p0c = g.c0 ? g.c0 + ch : safe; const float* p1c = g.c1 ? g.c1 + ch : safe;
    const float* p2c = g.c2 ? g.c2 + ch : safe; const float* p3c = (BWD && g.c3) ? g.c3 + ch : safe;
    float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(p0c + h, v0 + h); V4<float>::load(p1c + h, v1 + h);
      V4<float>::load(p2c + h, v2 + h); V4<float>::load(p3c + h, v3 + h);
    }
    const bool on = lane_on && cv_real;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float c0v = (on && g.c0) ? v0[j] : 1.f, c1v = (on && g.c1) ? v1[j] : 0.f;
      const float c2v = (on && g.c2) ? v2[j] : 0.f, c3v = (on && BWD && g.c3) ? v3[j] : 0.f;
      if (BWD) {       // g = c0*(e - c2) + c1*(y - c3)
        k0[j] = c0v; k1[j] = c1v; kadd[j] = -(c0v * c2v) - c1v * c3v;
      } else {         // a = (x - c1)*c0 + c2
        k0[j] = c0v; k1[j] = 0.f; kadd[j] = c2v - c1v * c0v;
      }
    }
  }

  if (tid < NCH) {       // the epilogue constants requested at the top
    Ec[tid] = (ec_in && ec_hm) ? ec0 : 0.f;
    Ec[NCH + tid] = (ec_in && ec_hs) ? ec1 : (BWD ? 1.f : 0.f);
    Ec[2 * NCH + tid] = (ec_in && ec_hb) ? ec2 : 0.f;
    if (!BWD && g.escale) {         // eval epilogue: v = acc * scale + shift, shift = beta + (bias - mean) * scale
      Ec[tid] = ec_in ? ec_be + ((ec_hm ? ec0 : 0.f) - ec_mu) * ec_sc : 0.f;
      Ec[NCH + tid] = ec_in ? ec_sc : 0.f;
    }
  }
  __syncthreads();

  TSS_T(tq1);
  for (long tile = t_begin; tile < t_end; tile += g.gslots) {
    const long p0 = tile * TM;
    const bool full = p0 + TM <= g.P;
    __syncthreads();  // previous tile's MFMA reads of Xs are done

    // bwd: the producer's raw output under this lane's outputs for THIS tile: prefetched with the tile (issue_loads)
    uint2 rxm[4][MF];
    if (BWD && g.xm) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < MF; ++m) rxm[i][m] = rxn[i][m];
    }

    // ---- normalise + store the A tile whose loads are in flight
    if (lane_on) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        if (ps < npass) {
          const int row = ps * rpp + r;
          if (row < TM) {
            const bool ok = cv_real && (full || p0 + row < g.P);
            const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra[ps]);
            const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb[ps]);
            float v[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              float lo = bits_lo(ua[h]) * k0[2 * h] + kadd[2 * h];
              float hi = bits_hi(ua[h]) * k0[2 * h + 1] + kadd[2 * h + 1];
              if (BWD) { lo += bits_lo(ub[h]) * k1[2 * h]; hi += bits_hi(ub[h]) * k1[2 * h + 1]; }
              v[2 * h] = lo; v[2 * h + 1] = hi;
            }
            if (!BWD) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], relu_lo);
            }
            if (DROP) {     // kept or zero; the factor 1 / (1 - p) is applied to the (few) outputs, in the epilogue
#pragma unroll
              for (int j = 0; j < 8; ++j) if (!((rm[ps] >> j) & 1u)) v[j] = 0.f;
            }
            if (!ok) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            V8<T>::store(Xs + row * RS + cv * 8, v);
          }
        }
      }
    }
    if (tile + g.gslots < t_end) issue_loads(tile + g.gslots);
    __syncthreads();

    // ---- MFMA: D[n][p] += W[n][k] * A[p][k]
    f32x4 acc[MF][4];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nfr > 0) {
      const T* xrow = Xs + (wm * (TM / 2) + fr) * RS + fq * 8;
      const T* wrow = Ws + (wn * 64 + fr) * RS + fq * 8;
      const int nks = kwp >> 5;
      for (int ks = 0; ks < nks; ++ks) {
        bf16x8 xf[MF];
#pragma unroll
        for (int m = 0; m < MF; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xrow + m * 16 * RS + ks * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < nfr) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + i * 16 * RS + ks * 32);
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[m][i], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue
    T* yrow = g.y + (p0 + wm * (TM / 2) + fr) * g.ldy + nlane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nfr) {
        const int n = nlane + i * 16;
        const bool nin = n < g.N;     // N % 4 == 0 (host-checked): a 4-channel group is in or out as a whole
        // per-channel epilogue constants come from LDS (Ec): a global load here would make every store of the tile
        // wait for vmcnt(0), i.e. for the prefetched loads of the next tile
        const int nl = wn * 64 + i * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(Ec + nl);
        const float4 e1 = *reinterpret_cast<const float4*>(Ec + NCH + nl);
        const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * NCH + nl);
        const float bs[4] = {e0.x, e0.y, e0.z, e0.w};                                  // fwd: bias
        const float cmm[4] = {e0.x, e0.y, e0.z, e0.w};                                 // bwd: mean / scale / bias
        const float cms[4] = {e1.x, e1.y, e1.z, e1.w}, cmb[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const bool pin = full || (p0 + wm * (TM / 2) + m * 16 + fr < g.P);
          if (pin && nin) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = BWD ? acc[m][i][q] : (DROP ? acc[m][i][q] * g.dinv + bs[q] : acc[m][i][q] + bs[q]);
            if (!BWD && g.escale) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q] * cms[q] + bs[q];
            }
            if (BWD && g.xm) {
              const uint2 xr = rxm[i][m];
              const float xc[4] = {bits_lo(xr.x) - cmm[0], bits_hi(xr.x) - cmm[1], bits_lo(xr.y) - cmm[2], bits_hi(xr.y) - cmm[3]};
              if (g.m_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(xc[q] * cms[q] + cmb[q] > 0.f)) v[q] = 0.f;
              }
              bf16x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * xc[q]; }
              *reinterpret_cast<bf16x4*>(yrow + (long)m * 16 * g.ldy + i * 16) = o;
            } else {
              if (g.radd) {     // bwd: fan-in of a residual block (the other gradient of this tensor); fwd, eval epilogue: the block's skip
                const long pr = p0 + wm * (TM / 2) + m * 16 + fr;
                const uint2 rr = *reinterpret_cast<const uint2*>(g.radd + pr * g.ldr + nlane + i * 16);
                v[0] += bits_lo(rr.x); v[1] += bits_hi(rr.x); v[2] += bits_lo(rr.y); v[3] += bits_hi(rr.y);
              }
              if (!BWD && g.erelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
              }
              bf16x4 o;
              if (JB && g.jout) {     // the backward of the join that produced this layer's input
                const long pr = p0 + wm * (TM / 2) + m * 16 + fr;
                const uint2 ov = *reinterpret_cast<const uint2*>(g.jout + pr * g.ldjo + nlane + i * 16);
                const uint2 yv = *reinterpret_cast<const uint2*>(g.jy + pr * g.ldjy + nlane + i * 16);
                const float oo[4] = {bits_lo(ov.x), bits_hi(ov.x), bits_lo(ov.y), bits_hi(ov.y)};
                const float yc[4] = {bits_lo(yv.x) - cmm[0], bits_hi(yv.x) - cmm[1], bits_lo(yv.y) - cmm[2], bits_hi(yv.y) - cmm[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (T)(oo[q] > 0.f ? v[q] : 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * yc[q]; }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
              }
              *reinterpret_cast<bf16x4*>(yrow + (long)m * 16 * g.ldy + i * 16) = o;
            }
          }
        }
      }
    }
  }

#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TSS_T(tq2);
  // ---- statistics slab row of this block (see convgemm.hip)
  if (g.stats) {
    __syncthreads();
    const int row = xcd + 8 * gslot, rows_used = 8 * g.gslots;
    float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][NCH], aliases Xs
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
        if (fr == 0) {
          red[(wm * 2 + 0) * NCH + wn * 64 + i * 16 + fq * 4 + q] = u;
          red[(wm * 2 + 1) * NCH + wn * 64 + i * 16 + fq * 4 + q] = w2;
        }
      }
    __syncthreads();
    if (tid < ncw) {
      const double a = (double)red[0 * NCH + tid] + (double)red[2 * NCH + tid];
      const double b = (double)red[1 * NCH + tid] + (double)red[3 * NCH + tid];
      g.stats[(long)row * 2 * g.N + n0 + tid] = a;
      g.stats[(long)row * 2 * g.N + g.N + n0 + tid] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * g.N + n0 + tid] = 0.0;
        g.stats[(long)rr * 2 * g.N + g.N + n0 + tid] = 0.0;
      }
    }
  }
#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSS_T(tq3);
  if (threadIdx.x == 0) { atomicAdd(&g_pw_timing[0], tq1 - tq0); atomicAdd(&g_pw_timing[1], tq2 - tq1); atomicAdd(&g_pw_timing[2], tq3 - tq2); atomicAdd(&g_pw_timing[3], tqa - tq0); atomicAdd(&g_pw_timing[4], tqb - tqa); atomicAdd(&g_pw_timing[5], tqc - tqb); atomicAdd(&g_pw_timing[6], tq1 - tqc); atomicAdd(&g_pw_timing[7], 1ull); }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Multi-chunk variant: contraction longer than 128 channels (forward of the 1x1 "project" convs, K = 192..768;
// backward-data of the "expand" convs, contraction over N = 192..768).  Same tiles and epilogue; the contraction is
// walked in 128-channel chunks with the accumulators kept in registers, each chunk's weights restaged from L2 and
// the folded BatchNorm constants of all K channels resident in LDS.
// (Round 2: requesting the epilogue's x values before the chunk loop, and the next chunk's A tile under the current chunk's
// MFMAs, cost 18-37 registers, the third wave per SIMD of the 32-pixel instance and 2-5 us per launch: measured, reverted.)
constexpr int KTOT = 768;
// TM = pixels per tile: 128, or 64 when a layer has fewer than 256 tiles of 128 (the 1/32-resolution project / expand
// layers: 128 blocks of 128 pixels left half the CUs idle behind a five-chunk serial chain).
template <bool BWD, int TM, bool JB = false>
__global__ __launch_bounds__(NT, 2) void pwfast_mc_kernel(const FastArgs g) {
  static_assert(!JB || BWD, "the join-backward epilogue belongs to backward-data");
  constexpr int MF = TM / 32;    // 16-pixel MFMA fragments per wave (two waves split the tile's pixels)
  constexpr int NP = TM / 16;    // staging passes (>= 16 rows per pass)
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Ws = Xs + TM * RS;
  float* Ck = reinterpret_cast<float*>(Ws + NCH * RS);   // [3][KTOT]: k0, k1, kadd
  float* Ec = Ck + 3 * KTOT;                              // [3][NCH] epilogue constants

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // the first nred8 blocks of a backward-data grid sum the weight-gradient workspace of the same layer (wgreduce.h)
  if (BWD && g.nred8 > 0 && (int)blockIdx.x < g.nred8) {
    if ((int)blockIdx.x < g.red.nred) tss_wg::reduce_block(g.red, blockIdx.x, reinterpret_cast<float4*>(smem));
    return;
  }
  const int bidx = BWD ? (int)blockIdx.x - g.nred8 : (int)blockIdx.x;

  const int nchunks = (g.N + NCH - 1) / NCH;
  const int xcd = bidx & 7, slot = bidx >> 3;
  const int nc = slot % nchunks, gslot = slot / nchunks;
  const long ntiles = (g.P + TM - 1) / TM;
  const long per = (ntiles + 7) >> 3;
  const long t_begin = xcd * per + gslot;
  long t_end = xcd * per + per;
  if (t_end > ntiles) t_end = ntiles;

  const int n0 = nc * NCH;
  const int ncw = (g.N - n0 < NCH) ? (g.N - n0) : NCH;
  const int nrows = ((ncw + 15) >> 4) * 16;
  int nfr = (ncw - wn * 64 + 15) >> 4;
  nfr = nfr < 0 ? 0 : (nfr > 4 ? 4 : nfr);
  // The epilogue constants of this block's channels are REQUESTED here and parked in LDS only at the end of the set-up (round 4): their
  // round trip used to be exposed in front of everything else -- loads, wait, LDS stores, barrier, and only then the first tile's
  // requests (2.1 - 2.4 k cycles of the 8 - 10 k-cycle prologue, tools/pw_timing.sh).  Now they fly together with the first tile and
  // the weights: one round trip instead of two.  Three independent loads (null-safe pointers + select), not three branches with a wait each.
  float ec0 = 0.f, ec1 = 0.f, ec2 = 0.f, ec_sc = 0.f, ec_mu = 0.f, ec_be = 0.f;
  bool ec_in = false, ec_hm = false, ec_hs = false, ec_hb = false;
  if (tid < NCH) {
    const int n = n0 + tid;
    const bool in = n < g.N;
    const int nn = in ? n : 0;
    const bool hj = JB && !g.xm && g.jout && g.jmean;
    const bool hm = BWD ? ((g.xm && g.mm) || hj) : (g.bias != nullptr), hs = BWD && g.xm && g.ms, hb = BWD && g.xm && g.mb;
    ec0 = (hm ? (BWD ? (hj ? g.jmean : g.mm) : g.bias) : g.w)[hm ? nn : 0];
    ec1 = (hs ? g.ms : g.w)[hs ? nn : 0];
    ec2 = (hb ? g.mb : g.w)[hb ? nn : 0];
    ec_in = in; ec_hm = hm; ec_hs = hs; ec_hb = hb;
    if (!BWD && g.escale) { ec_sc = g.escale[nn]; ec_mu = g.emean ? g.emean[nn] : 0.f; ec_be = g.ebeta ? g.ebeta[nn] : 0.f; }
  }
  const int K = g.K;
  const int nkc = (K + KMAX - 1) / KMAX;
  const float relu_lo = g.a_relu ? 0.f : -TSS_INF;

  for (int ch = tid; ch < K; ch += NT) {   // folded constants of every contraction channel, once per block
    const float* q0 = g.c0; const float* q1 = g.c1; const float* q2 = g.c2;
    int cc = ch;
    if (!BWD && g.nsrc > 0) { const int sI = ch >> 7; cc = ch & 127; q0 = g.c0s[sI]; q1 = g.c1s[sI]; q2 = g.c2s[sI]; }
    const float v0 = (q0 ? q0 : g.w)[q0 ? cc : 0], v1 = (q1 ? q1 : g.w)[q1 ? cc : 0];
    const float v2 = (q2 ? q2 : g.w)[q2 ? cc : 0], v3 = (g.c3 ? g.c3 : g.w)[g.c3 ? ch : 0];
    const float c0v = q0 ? v0 : 1.f, c1v = q1 ? v1 : 0.f, c2v = q2 ? v2 : 0.f, c3v = g.c3 ? v3 : 0.f;
    if (BWD) {       // g = c0*(e - c2) + c1*(y - c3)
      Ck[ch] = c0v; Ck[KTOT + ch] = c1v; Ck[2 * KTOT + ch] = -(c0v * c2v) - c1v * c3v;
    } else {         // a = (x - c1)*c0 + c2
      Ck[ch] = c0v; Ck[KTOT + ch] = 0.f; Ck[2 * KTOT + ch] = c2v - c1v * c0v;
    }
  }
  if (tid < NCH) {       // the epilogue constants requested at the top: their round trip ran under the loop above
    Ec[tid] = (ec_in && ec_hm) ? ec0 : 0.f;
    Ec[NCH + tid] = (ec_in && ec_hs) ? ec1 : (BWD ? 1.f : 0.f);
    Ec[2 * NCH + tid] = (ec_in && ec_hb) ? ec2 : 0.f;
    if (!BWD && g.escale) {         // eval epilogue: v = acc * scale + shift, shift = beta + (bias - mean) * scale
      Ec[tid] = ec_in ? ec_be + ((ec_hm ? ec0 : 0.f) - ec_mu) * ec_sc : 0.f;
      Ec[NCH + tid] = ec_in ? ec_sc : 0.f;
    }
  }

  float st1[4][4], st2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }
  const int nlane = n0 + wn * 64 + fq * 4;

  for (long tile = t_begin; tile < t_end; tile += g.gslots) {
    const long p0 = tile * TM;
    const bool full = p0 + TM <= g.P;
    f32x4 acc[MF][4];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nkc; ++kc) {
      const int kb = kc * KMAX;
      const int kw = (K - kb < KMAX) ? (K - kb) : KMAX;
      const int kwp = (kw + 31) & ~31;
      const int nvec = kw >> 3, nvecp = kwp >> 3;
      const int rpp = NT / nvecp, npass = (TM + rpp - 1) / rpp;
      const int cv = tid % nvecp, r = tid / nvecp;
      const bool lane_on = r < rpp, cv_real = cv < nvec;
      __syncthreads();  // previous chunk's MFMA reads are done (and Ck is visible on the first pass)

      // A-tile loads first (the HBM round trip), then the weight chunk from L2, then normalise + store
      uint4 ra[NP], rb[NP];
      if (lane_on) {
        const bool multi = !BWD && g.nsrc > 0;
        const int lda = multi ? (int)g.ldsrc[kc] : (int)g.lda0, ldb = BWD ? (int)g.lda1 : 0;
        const T* pa = (multi ? g.asrc[kc] + p0 * lda : g.a0 + p0 * g.lda0 + kb) + (cv_real ? cv * 8 : 0);
        const T* pb = BWD ? g.a1 + p0 * g.lda1 + kb + (cv_real ? cv * 8 : 0) : nullptr;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
          if (ps < npass) {
            const int row = ps * rpp + r;
            const bool ok = row < TM && (full || p0 + row < g.P);
            const int rr = ok ? row : 0;
            ra[ps] = *reinterpret_cast<const uint4*>(pa + rr * lda);
            if (BWD) rb[ps] = *reinterpret_cast<const uint4*>(pb + rr * ldb);
          }
        }
      }
      if (g.wb) stage_weights_bf16(Ws, g.wb, g.ldwb, n0, ncw, nrows, kb, kw, kwp, tid);
      else stage_weights(Ws, g.w, g.w_trans, g.w_trans ? (long)g.N : (long)K, n0, ncw, nrows, kb, kw, kwp, tid);
      if (lane_on) {
        float k0[8], k1[8], kadd[8];
        const int cb = kb + (cv_real ? cv * 8 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = Ck[cb + j]; k1[j] = Ck[KTOT + cb + j]; kadd[j] = Ck[2 * KTOT + cb + j]; }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
          if (ps < npass) {
            const int row = ps * rpp + r;
            if (row < TM) {
              const bool ok = cv_real && (full || p0 + row < g.P);
              const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra[ps]);
              const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb[ps]);
              float v[8];
#pragma unroll
              for (int h = 0; h < 4; ++h) {
                float lo = bits_lo(ua[h]) * k0[2 * h] + kadd[2 * h];
                float hi = bits_hi(ua[h]) * k0[2 * h + 1] + kadd[2 * h + 1];
                if (BWD) { lo += bits_lo(ub[h]) * k1[2 * h]; hi += bits_hi(ub[h]) * k1[2 * h + 1]; }
                v[2 * h] = lo; v[2 * h + 1] = hi;
              }
              if (!BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], relu_lo);
              }
              if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
              }
              V8<T>::store(Xs + row * RS + cv * 8, v);
            }
          }
        }
      }
      __syncthreads();

      if (nfr > 0) {
        const T* xrow = Xs + (wm * (TM / 2) + fr) * RS + fq * 8;
        const T* wrow = Ws + (wn * 64 + fr) * RS + fq * 8;
        const int nks = kwp >> 5;
        for (int ks = 0; ks < nks; ++ks) {
          bf16x8 xf[MF];
#pragma unroll
          for (int m = 0; m < MF; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xrow + m * 16 * RS + ks * 32);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < nfr) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + i * 16 * RS + ks * 32);
#pragma unroll
              for (int m = 0; m < MF; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[m][i], 0, 0, 0);
            }
          }
        }
      }
    }

    // ---- epilogue (identical to pwfast_kernel)
    T* yrow = g.y + (p0 + wm * (TM / 2) + fr) * g.ldy + nlane;
    const T* xrow_m = BWD && g.xm ? g.xm + (p0 + wm * (TM / 2) + fr) * g.ldxm + nlane : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nfr) {
        const int n = nlane + i * 16;
        const bool nin = n < g.N;
        // per-channel epilogue constants come from LDS (Ec): a global load here would make every store of the tile
        // wait for vmcnt(0), i.e. for the prefetched loads of the next tile
        const int nl = wn * 64 + i * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(Ec + nl);
        const float4 e1 = *reinterpret_cast<const float4*>(Ec + NCH + nl);
        const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * NCH + nl);
        const float bs[4] = {e0.x, e0.y, e0.z, e0.w};                                  // fwd: bias
        const float cmm[4] = {e0.x, e0.y, e0.z, e0.w};                                 // bwd: mean / scale / bias
        const float cms[4] = {e1.x, e1.y, e1.z, e1.w}, cmb[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const bool pin = full || (p0 + wm * (TM / 2) + m * 16 + fr < g.P);
          if (pin && nin) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = BWD ? acc[m][i][q] : acc[m][i][q] + bs[q];
            if (!BWD && g.escale) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q] * cms[q] + bs[q];
            }
            if (BWD && g.xm) {
              const uint2 xr = *reinterpret_cast<const uint2*>(xrow_m + (long)m * 16 * g.ldxm + i * 16);
              const float xc[4] = {bits_lo(xr.x) - cmm[0], bits_hi(xr.x) - cmm[1], bits_lo(xr.y) - cmm[2], bits_hi(xr.y) - cmm[3]};
              if (g.m_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(xc[q] * cms[q] + cmb[q] > 0.f)) v[q] = 0.f;
              }
              bf16x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * xc[q]; }
              *reinterpret_cast<bf16x4*>(yrow + (long)m * 16 * g.ldy + i * 16) = o;
            } else {
              if (g.radd) {     // bwd: fan-in of a residual block (the other gradient of this tensor); fwd, eval epilogue: the block's skip
                const long pr = p0 + wm * (TM / 2) + m * 16 + fr;
                const uint2 rr = *reinterpret_cast<const uint2*>(g.radd + pr * g.ldr + nlane + i * 16);
                v[0] += bits_lo(rr.x); v[1] += bits_hi(rr.x); v[2] += bits_lo(rr.y); v[3] += bits_hi(rr.y);
              }
              if (!BWD && g.erelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
              }
              bf16x4 o;
              if (JB && g.jout) {     // the backward of the join that produced this layer's input
                const long pr = p0 + wm * (TM / 2) + m * 16 + fr;
                const uint2 ov = *reinterpret_cast<const uint2*>(g.jout + pr * g.ldjo + nlane + i * 16);
                const uint2 yv = *reinterpret_cast<const uint2*>(g.jy + pr * g.ldjy + nlane + i * 16);
                const float oo[4] = {bits_lo(ov.x), bits_hi(ov.x), bits_lo(ov.y), bits_hi(ov.y)};
                const float yc[4] = {bits_lo(yv.x) - cmm[0], bits_hi(yv.x) - cmm[1], bits_lo(yv.y) - cmm[2], bits_hi(yv.y) - cmm[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (T)(oo[q] > 0.f ? v[q] : 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * yc[q]; }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
              }
              *reinterpret_cast<bf16x4*>(yrow + (long)m * 16 * g.ldy + i * 16) = o;
            }
          }
        }
      }
    }
  }

  if (g.stats) {
    __syncthreads();
    const int row = xcd + 8 * gslot, rows_used = 8 * g.gslots;
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
        if (fr == 0) {
          red[(wm * 2 + 0) * NCH + wn * 64 + i * 16 + fq * 4 + q] = u;
          red[(wm * 2 + 1) * NCH + wn * 64 + i * 16 + fq * 4 + q] = w2;
        }
      }
    __syncthreads();
    if (tid < ncw) {
      const double a = (double)red[0 * NCH + tid] + (double)red[2 * NCH + tid];
      const double b = (double)red[1 * NCH + tid] + (double)red[3 * NCH + tid];
      g.stats[(long)row * 2 * g.N + n0 + tid] = a;
      g.stats[(long)row * 2 * g.N + g.N + n0 + tid] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * g.N + n0 + tid] = 0.0;
        g.stats[(long)rr * 2 * g.N + g.N + n0 + tid] = 0.0;
      }
    }
  }
}

template <bool BWD, int TM, bool JB = false>
void launch_fast_mc_tm(FastArgs& g, hipStream_t stream) {
  constexpr size_t smem = (size_t)(TM + NCH) * RS * sizeof(T) + 3 * KTOT * sizeof(float) + 3 * NCH * sizeof(float);
  const int nchunks = (g.N + NCH - 1) / NCH;
  const long ntiles = (g.P + TM - 1) / TM;
  long gs = (ntiles + 7) / 8;
  long cap = 64 / nchunks;
  if (cap < 1) cap = 1;
  if (gs > cap) gs = cap;
  g.gslots = (int)gs;
  const int grid = 8 * nchunks * (int)gs + g.nred8;
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwfast_mc_kernel<BWD, TM, JB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL((pwfast_mc_kernel<BWD, TM, JB>), dim3(grid), dim3(NT), smem, stream, g);
}

template <bool BWD, bool JB = false>
void launch_fast_mc(FastArgs& g, hipStream_t stream) {
  const int nchunks = (g.N + NCH - 1) / NCH;
  // fewer than 256 block-tiles of 128 pixels: halve the tile so that every CU gets a block
  const long t128 = (g.P + BM - 1) / BM * nchunks;
  static const long thr32 = getenv("TSS_PW_MC_SMALL") ? atol(getenv("TSS_PW_MC_SMALL")) : 192;
  if (t128 < thr32) launch_fast_mc_tm<BWD, 32, JB>(g, stream);
  else if (t128 < 256) launch_fast_mc_tm<BWD, 64, JB>(g, stream);
  else launch_fast_mc_tm<BWD, BM, JB>(g, stream);
}

template <bool BWD, int TM, bool DROP = false, bool JB = false>
void launch_fast(FastArgs& g, hipStream_t stream) {
  constexpr size_t smem = (size_t)(TM + NCH) * RS * sizeof(T) + 3 * NCH * sizeof(float);
  const int nchunks = (g.N + NCH - 1) / NCH;
  const long ntiles = (g.P + TM - 1) / TM;
  long gs = (ntiles + 7) / 8;
  long cap = ((BWD ? TM == 32 : TM == 64) ? 96 : 64) / nchunks;   // the small-tile variants run 3 blocks per CU
  if (cap > 64) cap = 64;                                          // 8 * gslots slab rows
  if (cap < 1) cap = 1;
  if (gs > cap) gs = cap;
  g.gslots = (int)gs;
  const int grid = 8 * nchunks * (int)gs + g.nred8;
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwfast_kernel<BWD, TM, DROP, JB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL((pwfast_kernel<BWD, TM, DROP, JB>), dim3(grid), dim3(NT), smem, stream, g);
}

}  // namespace

#ifdef TSS_TIMING
extern "C" int tss_debug_pw_timing(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pw_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_timing), z, sizeof(z)); }
  return 0;
}
#endif

int g_tss_disable_fast = 0;   // tss_set_option(TSS_OPT_DISABLE_FAST_PATHS, 1): A/B switch for tests

// forward: y = act(x) W^T (+bias).  Returns false when the shape is not covered (caller uses convgemm_kernel).
bool tss_pwfast_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w, const void* w_bf16, const float* bias, void* y, long ldy, double* stats, long P, int K, int N,
                    hipStream_t stream) {
  // a ragged N (the 19-class classifier) is fine when there are no statistics and the row pitch has room for the
  // zero the last 4-channel group writes into the padding
  const bool n_ok = (N % 4) == 0 || (!stats && ldy >= (N + 3) / 4 * 4);
  if (g_tss_disable_fast || K > KTOT || (K % 8) != 0 || !n_ok || P <= 0) return false;
  FastArgs g = {};
  g.P = P; g.K = K; g.N = N;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w; g.w_trans = 0; g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  if (w_bf16 && (K % 8) == 0 && tss::aligned16(w_bf16)) { g.wb = (const T*)w_bf16; g.ldwb = K; }   // rows of K contraction channels
  if (K <= KMAX) {
    // 64-pixel tiles (3 blocks per CU instead of 2) unless the layer is long enough to amortise the big tile: the per-tile
    // chain of memory latencies, not bandwidth, bounds all but the longest layers (tools/pw_timing.sh, tools/ab_pw_tiles.sh)
    static const long thr = getenv("TSS_PW_FWD_SMALL") ? atol(getenv("TSS_PW_FWD_SMALL")) : 4200;
    const long t128 = (P + 127) / 128 * ((N + NCH - 1) / NCH);
    if (t128 < thr) launch_fast<false, 64>(g, stream); else launch_fast<false, 128>(g, stream);
  } else launch_fast_mc<false>(g, stream);
  return true;
}

// forward with the eval-mode BatchNorm behind the layer, the skip and the ReLU after the sum in the epilogue (see FastArgs::escale)
extern "C" int tss_pwconv_fwd_joined(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                     const float* w, const void* w_bf16, const float* bias,
                                     const float* out_mean, const float* out_scale, const float* out_beta,
                                     const void* residual, long ldr, int out_relu, void* y, long ldy,
                                     long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(!g_tss_disable_fast && P > 0 && K >= 8 && K <= KTOT && (K % 8) == 0 && N >= 4 && (N % 4) == 0 && x && w && y && out_scale,
              TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= K && (ldy % 4) == 0 && ldy >= N && (!residual || ((ldr % 4) == 0 && ldr >= N)), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && ((uintptr_t)y & 7u) == 0 && (!residual || ((uintptr_t)residual & 7u) == 0), TSS_ERR_ALIGN);
  FastArgs g = {};
  g.P = P; g.K = K; g.N = N;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w; g.w_trans = 0; g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = nullptr;
  if (w_bf16 && tss::aligned16(w_bf16)) { g.wb = (const T*)w_bf16; g.ldwb = K; }
  g.emean = out_mean; g.escale = out_scale; g.ebeta = out_beta; g.erelu = out_relu;
  g.radd = (const T*)residual; g.ldr = ldr;
  tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, (double)P * (K + N * (residual ? 2 : 1)) * 2.0, 2.0 * (double)P * K * N);
  if (K <= KMAX) {
    static const long thr = getenv("TSS_PW_FWD_SMALL") ? atol(getenv("TSS_PW_FWD_SMALL")) : 4200;
    const long t128 = (P + 127) / 128 * ((N + NCH - 1) / NCH);
    if (t128 < thr) launch_fast<false, 64>(g, (hipStream_t)stream); else launch_fast<false, 128>(g, (hipStream_t)stream);
  } else launch_fast_mc<false>(g, (hipStream_t)stream);
  return tss::check_last("pwconv_fwd_joined");
}

// forward with nn.Dropout applied on load: y = dropout(act(x)) W^T (+bias), mask bytes from tss_dropout_mask (pointwise.hip)
extern "C" int tss_pwconv_drop_supported(long P, int K, int N, int dtype) {
  return (!g_tss_disable_fast && dtype == TSS_BF16 && P > 0 && K >= 8 && K <= KMAX && (K % 8) == 0 && N >= 1 && N <= NCH) ? 1 : 0;
}

extern "C" int tss_pwconv_fwd_drop(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                   const float* w, const void* w_bf16, const float* bias, void* y, long ldy,
                                   const void* mask, float drop_p, unsigned long long* counter,
                                   long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(tss_pwconv_drop_supported(P, K, N, dtype) && drop_p > 0.f && drop_p < 1.f && mask && counter && x && w && y, TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= K && (ldy % 4) == 0 && ldy >= (N + 3) / 4 * 4, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && ((uintptr_t)y & 7u) == 0, TSS_ERR_ALIGN);
  FastArgs g = {};
  g.P = P; g.K = K; g.N = N;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w; g.w_trans = 0; g.bias = bias; g.y = (T*)y; g.ldy = ldy; g.stats = nullptr;
  if (w_bf16 && tss::aligned16(w_bf16)) { g.wb = (const T*)w_bf16; g.ldwb = K; }
  g.dmask = (const unsigned char*)mask; g.dinv = 1.f / (1.f - drop_p); g.dcounter = counter;
  tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, (double)P * (K + N) * 2.0 + (double)P * (K / 8), 2.0 * (double)P * K * N);
  static const long thr = getenv("TSS_PW_FWD_SMALL") ? atol(getenv("TSS_PW_FWD_SMALL")) : 4200;
  const long t128 = (P + 127) / 128;
  if (t128 < thr) launch_fast<false, 64, true>(g, (hipStream_t)stream); else launch_fast<false, 128, true>(g, (hipStream_t)stream);
  return tss::check_last("pwconv_fwd_drop");
}

// forward over the channel concatenation of nsrc 128-channel tensors (see FastArgs::nsrc): the 1x1 `project` layer of an ASPP head
// reading its five branches in place (models/aspp.py), eval mode.  lds[i] == 0: source i is ONE row (a [1, 128, 1, 1] map)
// broadcast to every pixel -- the image-pooling branch without its upsampled copy.
extern "C" int tss_pwconv_fwd_multi(const void* const* srcs, const long* lds, const float* const* means, const float* const* scales,
                                    const float* const* biases, int nsrc, int in_relu, const float* w, const void* w_bf16,
                                    const float* bias, void* y, long ldy, long P, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(nsrc >= 2 && nsrc <= 6 && P > 0 && N > 0 && (N % 4) == 0 && (ldy % 4) == 0 && ldy >= N && w && srcs && lds && y,
              TSS_ERR_SHAPE);
  FastArgs g = {};
  g.P = P; g.K = nsrc * KMAX; g.N = N; g.nsrc = nsrc; g.a_relu = in_relu;
  for (int i = 0; i < nsrc; ++i) {
    TSS_REQUIRE(srcs[i] && tss::aligned16(srcs[i]) && (lds[i] == 0 || (lds[i] >= KMAX && (lds[i] % 8) == 0)) && lds[i] < (1L << 30), TSS_ERR_SHAPE);
    g.asrc[i] = (const T*)srcs[i]; g.ldsrc[i] = lds[i];
    g.c0s[i] = scales ? scales[i] : nullptr; g.c1s[i] = means ? means[i] : nullptr; g.c2s[i] = biases ? biases[i] : nullptr;
  }
  g.a0 = g.asrc[0]; g.lda0 = g.ldsrc[0];
  g.w = w; g.w_trans = 0; g.bias = bias; g.y = (T*)y; g.ldy = ldy;
  if (w_bf16 && tss::aligned16(w_bf16)) { g.wb = (const T*)w_bf16; g.ldwb = g.K; }
  tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, (double)P * (g.K + N) * 2.0, 2.0 * (double)P * g.K * N);
  launch_fast_mc<false>(g, (hipStream_t)stream);
  return tss::check_last("pwconv_fwd_multi");
}

// backward-data: e_in = relu'(act(x)) * (g W), g = ga*(e-gce) + gb*(yraw-gmu); contraction over N (must be <= 128)
bool tss_pwfast_bwd_data(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                         const float* gce, const float* gmu, const float* w, const void* wT_bf16, const void* xraw, long ldx,
                         const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats, const float* red_ws, float* red_dw, long red_P, int red_K, int red_N,
                         long P, int K, int N, hipStream_t stream, const void* radd, long ldr,
                         const void* jout, long ldjo, const void* jy, long ldjy, const float* jmean, double* jstats) {
  if (g_tss_disable_fast || !yraw || N > KTOT || (N % 8) != 0 || (K % 4) != 0 || P <= 0) return false;
  if (radd && (xraw || (ldr % 4) != 0 || ldr < K)) return false;     // the add is only folded into the unmasked epilogue
  if (jout && (xraw || !jy || !jstats || (ldjo % 4) != 0 || ldjo < K || (ldjy % 4) != 0 || ldjy < K)) return false;
  FastArgs g = {};
  if (red_ws && red_dw) {   // weight-gradient slots (of this layer or of one further up the backward pass) are summed by the first blocks of this launch
    g.red = tss_wg::reduce_args(red_ws, red_dw, red_P, red_K, red_N);
    g.nred8 = (g.red.nred + 7) & ~7;
  }
  g.P = P; g.K = N; g.N = K;
  g.a0 = (const T*)e; g.lda0 = lde; g.a1 = (const T*)yraw; g.lda1 = ldyr; g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu;
  g.w = w; g.w_trans = 1; g.y = (T*)e_in; g.ldy = ldei; g.stats = bstats;
  if (wT_bf16 && tss::aligned16(wT_bf16)) { g.wb = (const T*)wT_bf16; g.ldwb = N; }   // transpose [conv K][conv N]: rows of N contraction channels
  g.xm = (const T*)xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  g.radd = (const T*)radd; g.ldr = ldr;
  if (jout) { g.jout = (const T*)jout; g.ldjo = ldjo; g.jy = (const T*)jy; g.ldjy = ldjy; g.jmean = jmean; g.stats = jstats; }
  if (N <= KMAX) {
    static const long thr = getenv("TSS_PW_BWD_SMALL") ? atol(getenv("TSS_PW_BWD_SMALL")) : 4200;   // 32-pixel tiles, as above
    const long t64 = (P + 63) / 64 * ((K + NCH - 1) / NCH);
    if (jout) { if (t64 < thr) launch_fast<true, 32, false, true>(g, stream); else launch_fast<true, 64, false, true>(g, stream); }
    else if (t64 < thr) launch_fast<true, 32>(g, stream); else launch_fast<true, 64>(g, stream);
  } else if (jout) launch_fast_mc<true, true>(g, stream);
  else launch_fast_mc<true>(g, stream);
  return true;
}

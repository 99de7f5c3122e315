// MFMA contraction kernel shared by every dense (non-depthwise) convolution of the hot path:
//
//   out[p][n] = sum_k A(p, k) * Wt(n, k)            p = output pixel (NHWC row), n = output channel
//
//   * pointwise 1x1 forward        A = relu?((x-mean)*scale+bias)          (deferred BN of the producer)
//   * pointwise 1x1 backward-data  A = ga*(e-ce) + gb*(y-mu)               (deferred BN-backward of this conv)
//   * dense 3x3 forward/backward   the same two prologues, gathered over 9 taps (im2col-free)
//   * 3-channel stem 3x3 stride 2  A gathered from the NCHW image, K = Cin*9 zero-padded to the MFMA step
//
// The product is computed transposed (D = W_tile x A_tile^T) so that one lane ends up owning 4
// consecutive channels of one pixel: NHWC stores are 8/16 B per lane and per-channel statistics are a
// 16-lane reduction.  Blocks are persistent over 128-pixel tiles; per-channel sum / sum-of-squares
// (forward) or sum(e) / sum(e*x) (backward) are accumulated in registers across tiles and leave the
// block as one f64 atomic per channel.  bf16 uses v_mfma_f32_16x16x32_bf16, f32 (parity path) uses
// v_mfma_f32_16x16x4_f32 which is an exact fmaf chain.
#include "common.h"

namespace {

constexpr int BM = 128;   // pixels per tile (4 waves x 32)
constexpr int NCH = 128;  // output channels per block
constexpr int NT = 256;

enum { A_PW = 0, A_TAPS = 1, A_STEM = 2 };

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KC = 128, KSTEP = 32, RS = KC + 8;
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag ld(const bf16_t* row, int ks, int q) {
    return *reinterpret_cast<const bf16x8*>(row + ks * 32 + q * 8);
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KC = 64, KSTEP = 4, RS = KC + 4;
  typedef float Frag;
  static __device__ __forceinline__ Frag ld(const float* row, int ks, int q) { return row[ks * 4 + q]; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

struct GemmArgs {
  long P;                 // output rows (pixels)
  int KD, ND, ntaps, mode;
  // A operand
  const void* a0; long lda0;
  const void* a1; long lda1;
  // one source : A = (a0 - c1) * c0 + c2            (c0 scale, c1 mean, c2 bias; NULL -> 1, 0, 0)
  // two sources: A = c0 * (a0 - c2) + c1 * (a1 - c3)   (c0 ga, c2 ce, c1 gb, c3 mu)
  const float* c0; const float* c1; const float* c2; const float* c3;
  int a_relu, a0_f32;
  int Hin, Win, Hout, Wout, stride, dil, tap_sign, Cin;
  int tap0, tstep1;       // A_TAPS: local tap t is tap (tap0 + t * (tstep1 + 1)) of the 3x3 grid -- (0, 0): all nine; (3, 0): the 1x3 row; (1, 2): the 3x1 column
  int gkh, gkw;           // A_TAPS, gkw > 0: a gkh x gkw tap grid instead (odd sides, centred: tap t = row t / gkw, column t % gkw)
  int tstride;            // A_TAPS, > 1: transposed gather (backward-data of a strided convolution): the source coordinate is
                          //   (p - off(tap)) / tstride where that division is exact, else the tap contributes nothing
  // weights: element (n, k, tap) at w[n*wrs + k*wcs + tap*wts]
  const float* w; long wrs, wcs, wts;
  const float* bias;
  // output + forward statistics
  void* y; long ldy; double* stats;
  // backward epilogue: mask / second moment source
  const void* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
  int gslots;             // tile slots per XCD per channel chunk
};

// MODE (A_PW / A_TAPS / A_STEM) and HAS_A1 (second operand = backward prologue) are template parameters: with them
// runtime, the dead paths alone cost ~150 spilled SGPRs (v_readlane/v_writelane traffic inside the hot loops).
template <typename T, bool HAS_A1, int MODE>
__global__ __launch_bounds__(NT, 2) void convgemm_kernel(const GemmArgs g) {
  typedef Mma<T> M;
  constexpr int KC = M::KC, RS = M::RS, KSTEP = M::KSTEP;
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Ws = Xs + BM * RS;
  float* Cs = reinterpret_cast<float*>(Ws + NCH * RS);  // [4][KC]
  double* Sacc = reinterpret_cast<double*>(Cs + 4 * KC);  // [NCH][2] exact statistics (f32 path)
  constexpr bool EXACT_STATS = sizeof(T) == 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;  // wave tile: pixels wm*64..+63, channels wn*64..+63 of the block tile

  const int nchunks = (g.ND + NCH - 1) / NCH;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nc = slot % nchunks, gslot = slot / nchunks;
  const long ntiles = (g.P + BM - 1) / BM;
  const long per = (ntiles + 7) >> 3;
  const long t_begin = xcd * per + gslot;
  long t_end = xcd * per + per;
  if (t_end > ntiles) t_end = ntiles;

  const int n0 = nc * NCH;
  const int ncw = (g.ND - n0 < NCH) ? (g.ND - n0) : NCH;
  const int nrows = ((ncw + 15) >> 4) * 16;
  int nfr = (ncw - wn * 64 + 15) >> 4;   // 16-channel fragments this wave owns (0..4)
  nfr = nfr < 0 ? 0 : (nfr > 4 ? 4 : nfr);

  const int kcpt = (g.KD + KC - 1) / KC;  // k-chunks per tap
  const int nkc = g.ntaps * kcpt;
  const bool w_resident = (nkc == 1);
  bool w_loaded = false;

  float st1[4][4], st2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[i][r] = 0.f; st2[i][r] = 0.f; }

  const long HWo = (long)g.Hout * g.Wout;
  const float relu_lo = g.a_relu ? 0.f : -TSS_INF;  // relu as one v_max with a per-kernel constant
  if (EXACT_STATS) {
    for (int i = tid; i < NCH * 2; i += NT) Sacc[i] = 0.0;  // first use is after the tile loop's barriers
  }

  // epilogue constants of the 16 channels this lane owns, loaded once (null-safe pointers, clamped channel, selects):
  // fetched inside the epilogue each was a predicated load in front of the stores (a branch and a wait apiece)
  float e_bias[4][4], e_mm[4][4], e_ms[4][4], e_mb[4][4];
  {
    const float* pb = g.bias ? g.bias : g.w; const float* pm = g.mm ? g.mm : g.w;
    const float* ps = g.ms ? g.ms : g.w; const float* pq = g.mb ? g.mb : g.w;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + fq * 4 + r;
        const bool in = n < g.ND;
        const int nb = (in && g.bias) ? n : 0, nm = (in && g.mm) ? n : 0, ns = (in && g.ms) ? n : 0, nq = (in && g.mb) ? n : 0;
        const float vb = pb[nb], vm = pm[nm], vs = ps[ns], vq = pq[nq];
        e_bias[i][r] = (in && g.bias) ? vb : 0.f;
        e_mm[i][r] = (in && g.mm) ? vm : 0.f;
        e_ms[i][r] = (in && g.ms) ? vs : 1.f;
        e_mb[i][r] = (in && g.mb) ? vq : 0.f;
      }
  }

  for (long tile = t_begin; tile < t_end; tile += g.gslots) {
    const long p0 = tile * BM;
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nkc; ++kc) {
      const int tap = kc / kcpt;
      const int k0 = (kc - tap * kcpt) * KC;
      const int kw = (g.KD - k0 < KC) ? (g.KD - k0) : KC;
      const int kwp = (kw + KSTEP - 1) / KSTEP * KSTEP;
      __syncthreads();  // all MFMA reads of the previous chunk are done

      // ---- 1. issue this thread's A-tile loads FIRST (raw, into registers): they are the only HBM round trip of
      //         the chunk and must overlap the coefficient / weight staging below, not queue behind it
      constexpr int MAXV = BM * (KC / 8) / NT;  // 16-byte vectors per thread per chunk (8 for bf16, 4 for f32)
      typename V8<T>::Raw r0[MAXV], r1[HAS_A1 ? MAXV : 1];
      bool vok[MAXV];
      const int nvec = (kwp + 7) >> 3;
      // (row, cv) of this thread's i-th vector.  When nvec divides 256 (16/8/4/2/1: every hot-path layer but the
      // 48/96-channel ones) cv is fixed per thread and row advances by 256/nvec: one division per chunk, not 16.
      const bool pow_map = (NT % nvec) == 0;
      const int row0 = tid / nvec, cv0 = tid - row0 * nvec, rstep = pow_map ? NT / nvec : 0;
      if (MODE != A_STEM) {
        int dy, dx;       // this tap's offset from the centre, in units of the dilation
        if (g.gkw > 0) { const int ty = tap / g.gkw; dy = ty - (g.gkh >> 1); dx = (tap - ty * g.gkw) - (g.gkw >> 1); }
        else { const int tgrid = g.tap0 + tap * (g.tstep1 + 1); const int ky = tgrid / 3; dy = ky - 1; dx = tgrid - ky * 3 - 1; }
        dy *= g.tap_sign * g.dil; dx *= g.tap_sign * g.dil;
        const T* a0 = reinterpret_cast<const T*>(g.a0);
        const T* a1 = reinterpret_cast<const T*>(g.a1);
        (void)a1;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int idx = tid + i * NT;
          int row, cv;
          if (pow_map) { row = row0 + i * rstep; cv = cv0; } else { row = idx / nvec; cv = idx - row * nvec; }
          const long p = p0 + row;
          long q = -1;
          if (idx < BM * nvec && p < g.P && cv * 8 < kw) {
            if (MODE == A_PW) {
              q = p;
            } else {
              const long b = p / HWo; const long rem = p - b * HWo;
              const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
              int iy = oy * g.stride + dy;
              int ix = ox * g.stride + dx;
              bool okq = iy >= 0 && ix >= 0;
              if (g.tstride > 1) {
                const int qy = iy / g.tstride, qx = ix / g.tstride;
                okq = okq && qy * g.tstride == iy && qx * g.tstride == ix;
                iy = qy; ix = qx;
              }
              if (okq && iy < g.Hin && ix < g.Win) q = (b * g.Hin + iy) * (long)g.Win + ix;
            }
          }
          vok[i] = q >= 0;
          const long qc = q >= 0 ? q : 0;
          const int cvc = (cv * 8 < kw) ? cv : 0;
          r0[i] = V8<T>::load_raw(a0 + qc * g.lda0 + k0 + cvc * 8);
          if (HAS_A1) r1[i] = V8<T>::load_raw(a1 + qc * g.lda1 + k0 + cvc * 8);
        }
      }

      // ---- 2. coefficients and (when not resident) weights
      if (MODE != A_STEM) {
        for (int j = tid; j < kw; j += NT) {   // four independent loads through null-safe pointers, then selects
          const float v0 = (g.c0 ? g.c0 + k0 + j : g.w)[0], v1 = (g.c1 ? g.c1 + k0 + j : g.w)[0];
          const float v2 = (g.c2 ? g.c2 + k0 + j : g.w)[0], v3 = (g.c3 ? g.c3 + k0 + j : g.w)[0];
          Cs[j] = g.c0 ? v0 : 1.f;
          Cs[KC + j] = g.c1 ? v1 : 0.f;
          Cs[2 * KC + j] = g.c2 ? v2 : 0.f;
          Cs[3 * KC + j] = g.c3 ? v3 : 0.f;
        }
      }
      if (!w_resident || !w_loaded) {
        const float* wt = g.w + tap * g.wts;
        if (g.wcs == 1 && (kw & 3) == 0 && (g.wrs & 3) == 0) {
          const int vpr = kwp >> 2;  // float4 per row
          const int total = nrows * vpr;
          for (int base = 0; base < total; base += NT * 8) {   // 8 independent float4 loads in flight per lane
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int idx = base + tid + u * NT;
              const int n = idx / vpr, jv = idx - n * vpr;
              // clamp + select, not a predicated load (a branch with its own wait per load: 16 serial L2 round trips)
              const bool ok = idx < total && n < ncw && jv * 4 < kw;
              wv[u] = *reinterpret_cast<const float4*>(wt + (ok ? (long)(n0 + n) * g.wrs + k0 + jv * 4 : 0));
              if (!ok) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int idx = base + tid + u * NT;
              if (idx < total) {
                const int n = idx / vpr, jv = idx - n * vpr;
                T* d = Ws + n * RS + jv * 4;
                d[0] = (T)wv[u].x; d[1] = (T)wv[u].y; d[2] = (T)wv[u].z; d[3] = (T)wv[u].w;
              }
            }
          }
        } else if (g.wrs == 1 && (ncw & 3) == 0 && (g.wcs & 3) == 0) {
          // transposed source (backward-data): w[(k0+j)*wcs + n].  One float4 = 4 output rows of one contraction
          // column j.  Lanes run over j (NOT over n): the four 2-byte LDS stores of a lane then land in one row
          // each with consecutive lanes on consecutive columns -> conflict-free; with lanes over n every store
          // of a wave would hit 2 banks (row pitch 272 B, 4-row stride).  The 16-byte global reads are L2 hits.
          const int vpc = nrows >> 2;  // float4 groups of rows
          const int total = kwp * vpc;
          for (int base = 0; base < total; base += NT * 8) {
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int idx = base + tid + u * NT;
              const int nv = idx / kwp, j = idx - nv * kwp;
              const bool ok = idx < total && j < kw && nv * 4 < ncw;
              wv[u] = *reinterpret_cast<const float4*>(wt + (ok ? (long)(k0 + j) * g.wcs + n0 + nv * 4 : 0));
              if (!ok) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int idx = base + tid + u * NT;
              if (idx < total) {
                const int nv = idx / kwp, j = idx - nv * kwp;
                T* d = Ws + (nv * 4) * RS + j;
                d[0] = (T)wv[u].x; d[RS] = (T)wv[u].y; d[2 * RS] = (T)wv[u].z; d[3 * RS] = (T)wv[u].w;
              }
            }
          }
        } else {
          for (int idx = tid; idx < nrows * kwp; idx += NT) {
            const int n = idx / kwp, j = idx - n * kwp;
            float v = 0.f;
            if (n < ncw && j < kw) v = wt[(long)(n0 + n) * g.wrs + (long)(k0 + j) * g.wcs];
            Ws[n * RS + j] = (T)v;
          }
        }
        w_loaded = true;
      }
      __syncthreads();  // coefficients (and weights) visible

      // ---- 3. normalise the loaded vectors and write the A tile: Xs[row][j] = A(p0+row, k0+j), zero beyond kw / P / borders
      if (MODE == A_STEM) {
        const float* src32 = reinterpret_cast<const float*>(g.a0);
        const T* srcT = reinterpret_cast<const T*>(g.a0);
        for (int idx = tid; idx < BM * kwp; idx += NT) {
          const int row = idx & (BM - 1), j = idx >> 7;  // BM == 128: pixels fastest -> row reads coalesce
          const long p = p0 + row;
          float v = 0.f;
          if (p < g.P && j < kw) {
            const int c = j / 9, t9 = j - c * 9, ky = t9 / 3, kx = t9 - ky * 3;
            const long b = p / HWo; const long rem = p - b * HWo;
            const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
            const int iy = oy * g.stride + (ky - 1) * g.dil, ix = ox * g.stride + (kx - 1) * g.dil;
            if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) {
              const long off = ((b * g.Cin + c) * g.Hin + iy) * (long)g.Win + ix;
              v = g.a0_f32 ? src32[off] : (float)srcT[off];
            }
          }
          Xs[row * RS + j] = (T)v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int idx = tid + i * NT;
          if (idx < BM * nvec) {
            int row, cv;
            if (pow_map) { row = row0 + i * rstep; cv = cv0; } else { row = idx / nvec; cv = idx - row * nvec; }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (vok[i]) {
              float x0[8];
              V8<T>::unpack(r0[i], x0);
              const float* cc = Cs + cv * 8;
              if (HAS_A1) {
                float x1[8];
                V8<T>::unpack(r1[HAS_A1 ? i : 0], x1);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                  v[j] = cc[j] * (x0[j] - cc[2 * KC + j]) + cc[KC + j] * (x1[j] - cc[3 * KC + j]);
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (x0[j] - cc[KC + j]) * cc[j] + cc[2 * KC + j];
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], relu_lo);
              if (cv * 8 + 8 > kw) {   // only the last, partial vector of a chunk
#pragma unroll
                for (int j = 0; j < 8; ++j) if (cv * 8 + j >= kw) v[j] = 0.f;
              }
            }
            V8<T>::store(Xs + row * RS + cv * 8, v);
          }
        }
      }
      __syncthreads();

      // ---- MFMA: D[n][p] += W[n][k] * A[p][k]
      if (nfr > 0) {
        const T* xrow = Xs + (wm * 64 + fr) * RS;
        const T* wrow = Ws + (wn * 64 + fr) * RS;
        const int nks = kwp / KSTEP;
        for (int ks = 0; ks < nks; ++ks) {
          typename M::Frag xf[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) xf[m] = M::ld(xrow + m * 16 * RS, ks, fq);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < nfr) {
              const typename M::Frag wf = M::ld(wrow + i * 16 * RS, ks, fq);
#pragma unroll
              for (int m = 0; m < 4; ++m) acc[m][i] = M::mma(wf, xf[m], acc[m][i]);
            }
          }
        }
      }
    }

    // ---- epilogue: lane owns pixel (fr) x channels n..n+3
    T* yo = reinterpret_cast<T*>(g.y);
    const T* xm = reinterpret_cast<const T*>(g.xm);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nfr) {
        const int n = n0 + wn * 64 + i * 16 + fq * 4;
        double d1[4] = {0.0, 0.0, 0.0, 0.0}, d2[4] = {0.0, 0.0, 0.0, 0.0};  // f32 parity path only
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const long p = p0 + wm * 64 + m * 16 + fr;
          if (p < g.P && n < g.ND) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[m][i][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += e_bias[i][r];
            if (xm) {
              float xr[4];
              V4<T>::load(xm + p * g.ldxm + n, xr);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float xc = xr[r] - e_mm[i][r];
                if (g.m_relu) {
                  const float a = g.ms ? (xc * e_ms[i][r] + e_mb[i][r]) : xc;
                  if (!(a > 0.f)) v[r] = 0.f;
                }
                v[r] = V8<T>::round(v[r]);
                if (EXACT_STATS) { d1[r] += (double)v[r]; d2[r] += (double)v[r] * (double)xc; }
                else { st1[i][r] += v[r]; st2[i][r] += v[r] * xc; }
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = V8<T>::round(v[r]);
                if (EXACT_STATS) { d1[r] += (double)v[r]; d2[r] += (double)v[r] * (double)v[r]; }
                else { st1[i][r] += v[r]; st2[i][r] += v[r] * v[r]; }
              }
            }
            V4<T>::store(yo + p * g.ldy + n, v);
          }
        }
        if (EXACT_STATS && g.stats) {
          // f32 parity path: products and sums in f64, so var = E[y^2] - mean^2 does not cancel in f32
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double a = d1[r], b = d2[r];
#pragma unroll
            for (int mk = 1; mk <= 8; mk <<= 1) { a += __shfl_xor(a, mk, 64); b += __shfl_xor(b, mk, 64); }
            if (fr == 0) {
              atomicAdd(&Sacc[(wn * 64 + i * 16 + fq * 4 + r) * 2 + 0], a);
              atomicAdd(&Sacc[(wn * 64 + i * 16 + fq * 4 + r) * 2 + 1], b);
            }
          }
        }
      }
    }
  }

  // ---- per-channel statistics: this block's partial sums go to slab row (xcd + 8*gslot), columns of its chunk;
  //      rows that no block owns are zeroed here (the caller never clears the buffer)
  if (g.stats) {
    __syncthreads();
    const int row = xcd + 8 * gslot, rows_used = 8 * g.gslots;
    double a = 0.0, b = 0.0;
    if (EXACT_STATS) {
      if (tid < ncw) { a = Sacc[tid * 2 + 0]; b = Sacc[tid * 2 + 1]; }
    } else {
      // bf16 path: f32 partials per lane, 16-lane reduce, then the two pixel-halves (wm) through LDS
      float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][NCH], aliases Xs
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float u = st1[i][r], w = st2[i][r];
#pragma unroll
          for (int mk = 1; mk <= 8; mk <<= 1) { u += __shfl_xor(u, mk, 64); w += __shfl_xor(w, mk, 64); }
          if (fr == 0) {
            red[(wm * 2 + 0) * NCH + wn * 64 + i * 16 + fq * 4 + r] = u;
            red[(wm * 2 + 1) * NCH + wn * 64 + i * 16 + fq * 4 + r] = w;
          }
        }
      __syncthreads();
      if (tid < ncw) {
#pragma unroll
        for (int wv = 0; wv < 2; ++wv) { a += red[(wv * 2 + 0) * NCH + tid]; b += red[(wv * 2 + 1) * NCH + tid]; }
      }
    }
    if (tid < ncw) {
      g.stats[(long)row * 2 * g.ND + n0 + tid] = a;
      g.stats[(long)row * 2 * g.ND + g.ND + n0 + tid] = b;
      for (int r = row + rows_used; r < TSS_STAT_SLABS; r += rows_used) {
        g.stats[(long)r * 2 * g.ND + n0 + tid] = 0.0;
        g.stats[(long)r * 2 * g.ND + g.ND + n0 + tid] = 0.0;
      }
    }
  }
}

template <typename T> size_t smem_bytes() {
  return (size_t)(BM + NCH) * Mma<T>::RS * sizeof(T) + 4 * Mma<T>::KC * sizeof(float) + NCH * 2 * sizeof(double);
}

template <typename T, bool HAS_A1, int MODE>
void launch_one(const GemmArgs& g, int grid, hipStream_t stream) {
  static tss::DevOnce attr;  // raise the dynamic-LDS limit once per instantiation
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convgemm_kernel<T, HAS_A1, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<T>());
  }
  hipLaunchKernelGGL((convgemm_kernel<T, HAS_A1, MODE>), dim3(grid), dim3(NT), smem_bytes<T>(), stream, g);
}

template <typename T>
void launch_typed(const GemmArgs& g, int grid, hipStream_t stream) {
  const bool a1 = g.a1 != nullptr;
  if (g.mode == A_PW) { if (a1) launch_one<T, true, A_PW>(g, grid, stream); else launch_one<T, false, A_PW>(g, grid, stream); }
  else if (g.mode == A_TAPS) { if (a1) launch_one<T, true, A_TAPS>(g, grid, stream); else launch_one<T, false, A_TAPS>(g, grid, stream); }
  else launch_one<T, false, A_STEM>(g, grid, stream);
}

void launch_inst(const GemmArgs& g, int dtype, int grid, hipStream_t stream) {
  if (dtype == TSS_BF16) launch_typed<bf16_t>(g, grid, stream); else launch_typed<float>(g, grid, stream);
}

int launch(GemmArgs& g, int dtype, int kernel_id, hipStream_t stream, double alg_bytes) {
  if (g.P <= 0) return TSS_OK;
  const int nchunks = (g.ND + NCH - 1) / NCH;
  const long ntiles = (g.P + BM - 1) / BM;
  long gs = (ntiles + 7) / 8;
  long cap = 64 / nchunks;  // <= 512 blocks: 2 per CU, matching the 2-blocks/CU LDS budget
  if (cap < 1) cap = 1;
  if (gs > cap) gs = cap;
  g.gslots = (int)gs;
  const int grid = 8 * nchunks * (int)gs;
  const double flops = 2.0 * (double)g.P * g.KD * g.ntaps * g.ND;
  tss::ProfScope prof(kernel_id, stream, alg_bytes, flops);
  launch_inst(g, dtype, grid, stream);
  return tss::check_last("convgemm");
}

inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

extern int g_tss_disable_fast;   // pwfast.hip
// fcg.hip: lean bf16 kernels of the 64-channel five-tap and 128-channel three-tap layers (false: shape not covered)
bool tss_fcg_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                 const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                 int B, int H, int W, int Cin, int N, int T, int axis, int dil, hipStream_t stream);
bool tss_fcg_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                      const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                      const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                      void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int T, int axis, int dil, hipStream_t stream);
// sconv.hip: lean bf16 kernels of the stride-2 dense 3x3 (false: shape not covered)
bool tss_sconv_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                   const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                   int B, int Hin, int Win, int Cin, int N, hipStream_t stream);
bool tss_sconv_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                        const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                        const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                        void* e_in, long ldei, double* bstats, int B, int Hin, int Win, int Cin, int N, hipStream_t stream);
bool tss_sconv_transposed_fwd(const void* x, long ldx, const float* w_tcn, const float* bias, void* y, long ldy,
                              int B, int Hout, int Wout, int Cout, int Cin_t, hipStream_t stream);
// fc1d.hip: lean bf16 kernels of the three-tap layers (false: shape not covered)
bool tss_fc1d_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                  const float* w_tnc, int torch_layout, const float* bias, void* y, long ldy, double* stats,
                  int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream);
bool tss_fc1d_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                       const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn, int torch_layout,
                       const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                       void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int axis, int dil, hipStream_t stream);
// conv3x3.hip
bool tss_conv3x3_lean_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          const void* w9, void* y, long ldy, double* stats, int B, int H, int W, int Cin, int N,
                          int stride, int dil, hipStream_t stream);
bool tss_conv3x3_lean_bwd_data(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                               const float* gce, const float* gmu, const void* w9t, const void* xraw, long ldx,
                               const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                               void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int dil,
                               hipStream_t stream);
bool tss_pwfast_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w, const void* w_bf16, const float* bias, void* y, long ldy, double* stats, long P, int K, int N,
                    hipStream_t stream);                                                                    // pwfast.hip
bool tss_pwfast_bwd_data(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                         const float* gce, const float* gmu, const float* w, const void* wT_bf16, const void* xraw, long ldx,
                         const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats, const float* red_ws, float* red_dw, long red_P, int red_K, int red_N,
                         long P, int K, int N, hipStream_t stream, const void* radd = nullptr, long ldr = 0,
                         const void* jout = nullptr, long ldjo = 0, const void* jy = nullptr, long ldjy = 0,
                         const float* jmean = nullptr, double* jstats = nullptr);  // pwfast.hip
void tss_wg_reduce_standalone(const float* ws, float* dw, long P, int K, int N, hipStream_t stream);  // wgrad.hip
bool tss_conv3x3_wstat_fwd(const void* x, long ldx, const float* in_scale, int in_relu, const void* w9, void* y, long ldy,
                           double* stats, int B, int H, int W, int Cin, int N, int stride, int dil, hipStream_t stream);
bool tss_conv3x3_stream_fwd(const void* x, long ldx, const float* in_scale, int in_relu, const void* w9, void* y, long ldy,
                            double* stats, int B, int H, int W, int Cin, int N, int stride, int dil, hipStream_t stream);  // atrous.hip
bool tss_stem_direct_fwd(const void* x_nchw, int x_is_f32, const float* w, void* y, long ldy, double* stats,
                         int B, int Cin, int Hin, int Win, int N, int stride, int dtype, hipStream_t stream);  // stem.hip

extern "C" {

int tss_pwconv_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                   const float* w, const void* w_bf16, const float* bias, void* y, long ldy, double* stats,
                   long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(K > 0 && N > 0 && (K % 8) == 0 && (ldx % 8) == 0 && (ldy % 4) == 0 && ldx >= K && ldy >= (N + 3) / 4 * 4,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y) && tss::aligned16(w), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.P = P; g.KD = K; g.ND = N; g.ntaps = 1; g.mode = A_PW;
  g.a0 = x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.Hout = 1; g.Wout = 1;
  g.w = w; g.wrs = K; g.wcs = 1; g.wts = 0; g.bias = bias;
  g.y = y; g.ldy = ldy; g.stats = stats;
  if (dtype == TSS_BF16 && !g_tss_disable_fast && K <= 768) {   // lean bf16 kernels (pwfast.hip)
    tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, (double)P * (K + N) * 2, 2.0 * (double)P * K * N);
    if (tss_pwfast_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w, w_bf16, bias, y, ldy, stats, P, K, N, (hipStream_t)stream))
      return tss::check_last("pwfast_fwd");
  }
  return launch(g, dtype, TSS_K_PWCONV_FWD, (hipStream_t)stream, (double)P * (K + N) * esz(dtype));
}

int tss_pwconv_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                        const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                        const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                        void* e_in, long ldei, double* bstats, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                        long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(K > 0 && N > 0 && (K % 8) == 0 && (lde % 8) == 0 && lde >= (N + 7) / 8 * 8 && (ldei % 4) == 0 && ldei >= K,
              TSS_ERR_SHAPE);
  if (wg_P <= 0) { wg_P = P; wg_K = K; wg_N = N; }      // the slots of this layer's own weight gradient
  TSS_REQUIRE(!(wg_ws && wg_dw) || (wg_K > 0 && wg_N > 0 && (wg_K % 8) == 0 && (wg_N % 8) == 0), TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= (N + 7) / 8 * 8), TSS_ERR_SHAPE);
  TSS_REQUIRE(!xraw || ((ldx % 4) == 0 && ldx >= K), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || xraw, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(e_in) && tss::aligned16(w), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.P = P; g.KD = N; g.ND = K; g.ntaps = 1; g.mode = A_PW;
  g.a0 = e; g.lda0 = lde; g.a1 = yraw; g.lda1 = ldyr;
  if (yraw) { g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu; } else { g.c0 = ga; }
  g.Hout = 1; g.Wout = 1;
  g.w = w; g.wrs = 1; g.wcs = K; g.wts = 0;  // Wt(k_out, n) = w[n*K + k_out]
  g.y = e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  const double bytes = (double)P * (N * (yraw ? 2 : 1) + K * (xraw ? 2 : 1)) * esz(dtype);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && yraw && N <= 768 && (N % 8) == 0 && (K % 4) == 0) {
    tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream, bytes, 2.0 * (double)P * K * N);
    if (tss_pwfast_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, wT_bf16, xraw, ldx, in_mean, in_scale, in_bias, in_relu,
                            e_in, ldei, bstats, wg_ws, wg_dw, wg_P, wg_K, wg_N, P, K, N, (hipStream_t)stream))
      return tss::check_last("pwfast_bwd_data");
  }
  if (wg_ws && wg_dw) tss_wg_reduce_standalone(wg_ws, wg_dw, wg_P, wg_K, wg_N, (hipStream_t)stream);   // nobody else will
  return launch(g, dtype, TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream, bytes);
}

// tss_pwconv_bwd_data with the fan-in of a residual block folded in: e_in = g W + radd (bf16 [P][K], pitch ldr).  Only on the lean
// bf16 path and only for a materialised input (no xraw): tss_pwconv_bwd_data_radd_supported says whether the call will be taken;
// otherwise the caller adds the two gradients itself.
int tss_pwconv_bwd_data_radd_supported(long P, int K, int N, int dtype) {
  return dtype == TSS_BF16 && !g_tss_disable_fast && N <= 768 && (N % 8) == 0 && (K % 8) == 0 && P > 0;
}

int tss_pwconv_bwd_data_radd(const void* e, long lde, const void* yraw, long ldyr,
                             const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                             void* e_in, long ldei, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                             const void* radd, long ldr, long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(tss_pwconv_bwd_data_radd_supported(P, K, N, dtype) && yraw && radd, TSS_ERR_SHAPE);
  if (wg_P <= 0) { wg_P = P; wg_K = K; wg_N = N; }
  TSS_REQUIRE(!(wg_ws && wg_dw) || (wg_K > 0 && wg_N > 0 && (wg_K % 8) == 0 && (wg_N % 8) == 0), TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= N && (ldyr % 8) == 0 && ldyr >= N && (ldei % 4) == 0 && ldei >= K && (ldr % 4) == 0 && ldr >= K, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(e_in) && tss::aligned16(w) && ((uintptr_t)radd & 7u) == 0, TSS_ERR_ALIGN);
  const double bytes = (double)P * (N * 2 + K * 2) * 2.0;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream, bytes, 2.0 * (double)P * K * N);
  if (!tss_pwfast_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, wT_bf16, nullptr, 0, nullptr, nullptr, nullptr, 0,
                           e_in, ldei, nullptr, wg_ws, wg_dw, wg_P, wg_K, wg_N, P, K, N, (hipStream_t)stream, radd, ldr))
    return TSS_ERR_SHAPE;
  return tss::check_last("pwfast_bwd_data_radd");
}

// tss_pwconv_bwd_data_radd (radd optional here) for a layer whose materialised input is the output of a relu join (a block output):
// the backward of THAT join runs in this launch's epilogue -- e_in = (g W + radd) masked by join_out > 0, and join_bstats receives the
// slab rows of sum(e_in), sum(e_in * (join_y - join_mean)) for the BatchNorm behind join_y.  Valid only when e_in is the complete
// gradient of that join's output (the caller checks; the join's own backward kernel remains the fallback).
int tss_pwconv_bwd_data_joined(const void* e, long lde, const void* yraw, long ldyr,
                               const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                               void* e_in, long ldei, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                               const void* radd, long ldr, const void* join_out, long ldjo, const void* join_y, long ldjy,
                               const float* join_mean, double* join_bstats, long P, int K, int N, int dtype, void* stream) {
  TSS_REQUIRE(tss_pwconv_bwd_data_radd_supported(P, K, N, dtype) && yraw && join_out && join_y && join_bstats, TSS_ERR_SHAPE);
  if (wg_P <= 0) { wg_P = P; wg_K = K; wg_N = N; }
  TSS_REQUIRE(!(wg_ws && wg_dw) || (wg_K > 0 && wg_N > 0 && (wg_K % 8) == 0 && (wg_N % 8) == 0), TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= N && (ldyr % 8) == 0 && ldyr >= N && (ldei % 4) == 0 && ldei >= K
              && (!radd || ((ldr % 4) == 0 && ldr >= K)) && (ldjo % 4) == 0 && ldjo >= K && (ldjy % 4) == 0 && ldjy >= K, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(e_in) && tss::aligned16(w) && (!radd || ((uintptr_t)radd & 7u) == 0)
              && ((uintptr_t)join_out & 7u) == 0 && ((uintptr_t)join_y & 7u) == 0, TSS_ERR_ALIGN);
  const double bytes = (double)P * (N * 2 + K * (radd ? 4 : 3)) * 2.0;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream, bytes, 2.0 * (double)P * K * N);
  if (!tss_pwfast_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w, wT_bf16, nullptr, 0, nullptr, nullptr, nullptr, 0,
                           e_in, ldei, nullptr, wg_ws, wg_dw, wg_P, wg_K, wg_N, P, K, N, (hipStream_t)stream, radd, ldr,
                           join_out, ldjo, join_y, ldjy, join_mean, join_bstats))
    return TSS_ERR_SHAPE;
  return tss::check_last("pwfast_bwd_data_joined");
}

int tss_conv3x3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const void* w_tnc_bf16, void* y, long ldy, double* stats,
                    int B, int Hin, int Win, int Cin, int N, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (ldx % 8) == 0 && ldx >= Cin && (ldy % 4) == 0 && ldy >= N && stride >= 1 && dil >= 1,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = dil; g.tap_sign = 1; g.Cin = Cin;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;  // pad = dil, k = 3
  g.P = (long)B * g.Hout * g.Wout; g.KD = Cin; g.ND = N; g.ntaps = 9; g.mode = A_TAPS;
  g.a0 = x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.wrs = Cin; g.wcs = 1; g.wts = (long)N * Cin;  // [tap][n][c]
  g.y = y; g.ldy = ldy; g.stats = stats;
  const double bytes = ((double)B * Hin * Win * Cin + (double)g.P * N) * esz(dtype);
  if (dtype == TSS_BF16 && w_tnc_bf16 && !g_tss_disable_fast && tss::aligned16(w_tnc_bf16)) {
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, bytes, 18.0 * (double)g.P * Cin * N);
    // a materialised input (no BatchNorm / ReLU pending): activations streamed into the MFMA operand registers, any dilation
    // 128 -> 128 without statistics: weights stationary in registers, activation rows walked through an LDS ring (wstat.hip)
    if (tss_conv3x3_wstat_fwd(x, ldx, in_scale, in_relu, w_tnc_bf16, y, ldy, stats, B, Hin, Win, Cin, N, stride, dil,
                              (hipStream_t)stream))
      return tss::check_last("conv3x3_wstat_fwd");
    if (tss_conv3x3_stream_fwd(x, ldx, in_scale, in_relu, w_tnc_bf16, y, ldy, stats, B, Hin, Win, Cin, N, stride, dil,
                               (hipStream_t)stream))
      return tss::check_last("conv3x3_stream_fwd");
    if (tss_conv3x3_lean_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc_bf16, y, ldy, stats, B, Hin, Win, Cin, N,
                             stride, dil, (hipStream_t)stream))
      return tss::check_last("conv3x3_lean_fwd");
  }
  TSS_REQUIRE(w_tnc, TSS_ERR_SHAPE);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && stride == 2 && dil == 1 && Cin <= 64 && N <= 128) {   // sconv.hip
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, bytes, 18.0 * (double)g.P * Cin * N);
    if (tss_sconv_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc, nullptr, y, ldy, stats, B, Hin, Win, Cin, N, (hipStream_t)stream))
      return tss::check_last("sconv_fwd");
  }
  return launch(g, dtype, TSS_K_CONV3X3_FWD, (hipStream_t)stream, bytes);
}

int tss_conv3x3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* w_tcn_bf16,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int H, int W, int Cin, int N, int dil, int dtype, void* stream) {
  // stride-1 dense 3x3 only (the hot path has no strided dense conv with Cin > 3)
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (N % 8) == 0 && (Cin % 4) == 0 && (lde % 8) == 0 && lde >= N && (ldei % 4) == 0 && ldei >= Cin,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || xraw, TSS_ERR_SHAPE);
  GemmArgs g = {};
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1; g.dil = dil; g.tap_sign = -1; g.Cin = N;
  g.P = (long)B * H * W; g.KD = N; g.ND = Cin; g.ntaps = 9; g.mode = A_TAPS;
  g.a0 = e; g.lda0 = lde; g.a1 = yraw; g.lda1 = ldyr;
  if (yraw) { g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu; } else { g.c0 = ga; }
  g.w = w_tcn; g.wrs = N; g.wcs = 1; g.wts = (long)Cin * N;  // [tap][ci][co]
  g.y = e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  const double bytes = (double)g.P * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * esz(dtype);
  if (dtype == TSS_BF16 && w_tcn_bf16 && !g_tss_disable_fast && tss::aligned16(w_tcn_bf16) && tss::aligned16(e) &&
      tss::aligned16(e_in)) {
    tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, bytes, 18.0 * (double)g.P * Cin * N);
    if (tss_conv3x3_lean_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w_tcn_bf16, xraw, ldx, in_mean, in_scale, in_bias,
                                  in_relu, e_in, ldei, bstats, B, H, W, Cin, N, dil, (hipStream_t)stream))
      return tss::check_last("conv3x3_lean_bwd_data");
  }
  TSS_REQUIRE(w_tcn, TSS_ERR_SHAPE);
  return launch(g, dtype, TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, bytes);
}

// ---- factorized (1-D) dense convolutions: nn.Conv2d(C, C, (1,3) | (3,1), padding = (0,d) | (d,0), dilation = (1,d) | (d,1)) of
// FactorizedConvBlock, TSS/models/lednet.py:157-180 (and esnet.py:83-166).  Same implicit-GEMM kernel as the dense 3x3 with three
// of its nine taps: axis 0 = along W (the 1x3 conv: taps 3, 4, 5 of the grid), axis 1 = along H (3x1: taps 1, 4, 7).
int tss_conv1d3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                    int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (ldx % 8) == 0 && ldx >= Cin && (ldy % 4) == 0 && ldy >= N && dil >= 1 &&
              (axis == 0 || axis == 1) && w_tnc, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin == N && (N == 16 || N == 32 || N == 64)) {   // lean kernel (fc1d.hip)
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, (double)B * H * W * (Cin + N) * 2.0, 2.0 * B * H * W * 3.0 * Cin * N);
    if (tss_fc1d_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc, 0, bias, y, ldy, stats, B, H, W, Cin, N, axis, dil, (hipStream_t)stream))
      return tss::check_last("fc1d_fwd");
  }
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin == N && N == 128) {   // fcg.hip
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, (double)B * H * W * (Cin + N) * 2.0, 2.0 * B * H * W * 3.0 * Cin * N);
    if (tss_fcg_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc, bias, y, ldy, stats, B, H, W, Cin, N, 3, axis, dil, (hipStream_t)stream))
      return tss::check_last("fcg_fwd");
  }
  GemmArgs g = {};
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1; g.dil = dil; g.tap_sign = 1; g.Cin = Cin;
  g.tap0 = axis == 0 ? 3 : 1; g.tstep1 = axis == 0 ? 0 : 2;
  g.P = (long)B * H * W; g.KD = Cin; g.ND = N; g.ntaps = 3; g.mode = A_TAPS;
  g.a0 = x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.wrs = Cin; g.wcs = 1; g.wts = (long)N * Cin; g.bias = bias;
  g.y = y; g.ldy = ldy; g.stats = stats;
  return launch(g, dtype, TSS_K_CONV3X3_FWD, (hipStream_t)stream, (double)g.P * (Cin + N) * esz(dtype));
}

int tss_conv1d3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (N % 8) == 0 && (Cin % 4) == 0 && (lde % 8) == 0 && lde >= N && (ldei % 4) == 0 && ldei >= Cin &&
              (axis == 0 || axis == 1) && w_tcn, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || xraw, TSS_ERR_SHAPE);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin == N && (N == 16 || N == 32 || N == 64)) {
    tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, (double)B * H * W * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * 2.0,
                        2.0 * B * H * W * 3.0 * Cin * N);
    if (tss_fc1d_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w_tcn, 0, xraw, ldx, in_mean, in_scale, in_bias, in_relu, e_in, ldei, bstats,
                          B, H, W, Cin, N, axis, dil, (hipStream_t)stream))
      return tss::check_last("fc1d_bwd_data");
  }
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin == N && N == 128) {
    tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, (double)B * H * W * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * 2.0,
                        2.0 * B * H * W * 3.0 * Cin * N);
    if (tss_fcg_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w_tcn, xraw, ldx, in_mean, in_scale, in_bias, in_relu, e_in, ldei, bstats,
                         B, H, W, Cin, N, 3, axis, dil, (hipStream_t)stream))
      return tss::check_last("fcg_bwd_data");
  }
  GemmArgs g = {};
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.stride = 1; g.dil = dil; g.tap_sign = -1; g.Cin = N;
  g.tap0 = axis == 0 ? 3 : 1; g.tstep1 = axis == 0 ? 0 : 2;
  g.P = (long)B * H * W; g.KD = N; g.ND = Cin; g.ntaps = 3; g.mode = A_TAPS;
  g.a0 = e; g.lda0 = lde; g.a1 = yraw; g.lda1 = ldyr;
  if (yraw) { g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu; } else { g.c0 = ga; }
  g.w = w_tcn; g.wrs = N; g.wcs = 1; g.wts = (long)Cin * N;
  g.y = e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  return launch(g, dtype, TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream,
                (double)g.P * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * esz(dtype));
}

// ---- general dense convolution: kh x kw taps (odd sides), padding = dilation * (k - 1) / 2 per axis, any stride, optional bias.
// The strided 3x3 / 5x5 / 7x7 layers of LEDNet's APN decoder (TSS/models/lednet.py:62-64), the strided 3x3 of the
// DownsamplingBlocks (lednet.py:130-131, esnet.py:54-56) and ESNet's 1x5 / 5x1 factorized layers (esnet.py:83-113): small layers
// off the benchmarked path, on the generic implicit-GEMM kernel with a kh x kw tap grid.  Weights as [kh*kw][N][Cin] (fwd) and
// [kh*kw][Cin][N] (bwd_data) from tss_permute_wtaps(T = kh*kw).
int tss_convkxk_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                    int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (ldx % 8) == 0 && ldx >= Cin && (ldy % 4) == 0 && ldy >= N && stride >= 1 && dil >= 1 &&
              kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && kh * kw <= 81 && w_tnc, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = dil; g.tap_sign = 1; g.Cin = Cin; g.gkh = kh; g.gkw = kw;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  g.P = (long)B * g.Hout * g.Wout; g.KD = Cin; g.ND = N; g.ntaps = kh * kw; g.mode = A_TAPS;
  g.a0 = x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w = w_tnc; g.wrs = Cin; g.wcs = 1; g.wts = (long)N * Cin; g.bias = bias;
  g.y = y; g.ldy = ldy; g.stats = stats;
  if (dtype == TSS_BF16 && !g_tss_disable_fast && stride == 1 && Cin == N && N == 64 && ((kh == 1 && kw == 5) || (kh == 5 && kw == 1))) {   // fcg.hip
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, (double)g.P * (Cin + N) * 2.0, 2.0 * (double)g.P * 5.0 * Cin * N);
    if (tss_fcg_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc, bias, y, ldy, stats, B, Hin, Win, Cin, N, 5, kh == 1 ? 0 : 1, dil,
                    (hipStream_t)stream))
      return tss::check_last("fcg_fwd");
  }
  if (dtype == TSS_BF16 && !g_tss_disable_fast && kh == 3 && kw == 3 && stride == 2 && dil == 1 && Cin <= 64 && N <= 128) {   // sconv.hip
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, ((double)B * Hin * Win * Cin + (double)g.P * N) * 2.0, 18.0 * (double)g.P * Cin * N);
    if (tss_sconv_fwd(x, ldx, in_mean, in_scale, in_bias, in_relu, w_tnc, bias, y, ldy, stats, B, Hin, Win, Cin, N, (hipStream_t)stream))
      return tss::check_last("sconv_fwd");
  }
  return launch(g, dtype, TSS_K_CONV3X3_FWD, (hipStream_t)stream, ((double)B * Hin * Win * Cin + (double)g.P * N) * esz(dtype));
}

// e: [B][Hout][Wout][N] (Hout = (Hin - 1) / stride + 1), e_in: [B][Hin][Win][Cin]
int tss_convkxk_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (N % 8) == 0 && (Cin % 4) == 0 && (lde % 8) == 0 && lde >= N && (ldei % 4) == 0 && ldei >= Cin &&
              stride >= 1 && dil >= 1 && kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && kh * kw <= 81 && w_tcn, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || xraw, TSS_ERR_SHAPE);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && stride == 1 && Cin == N && N == 64 && ((kh == 1 && kw == 5) || (kh == 5 && kw == 1))) {
    const double px = (double)B * Hin * Win;
    tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, px * (N * (yraw ? 2 : 1) + Cin * (xraw ? 2 : 1)) * 2.0, 2.0 * px * 5.0 * Cin * N);
    if (tss_fcg_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w_tcn, xraw, ldx, in_mean, in_scale, in_bias, in_relu, e_in, ldei, bstats,
                         B, Hin, Win, Cin, N, 5, kh == 1 ? 0 : 1, dil, (hipStream_t)stream))
      return tss::check_last("fcg_bwd_data");
  }
  if (dtype == TSS_BF16 && !g_tss_disable_fast && kh == 3 && kw == 3 && stride == 2 && dil == 1 && ((Cin == N && (N == 32 || N == 64)) || (Cin == 16 && N == 48))) {
    const double po = (double)B * ((Hin - 1) / 2 + 1) * ((Win - 1) / 2 + 1);
    tss::ProfScope prof(TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream, (po * N * (yraw ? 2 : 1) + (double)B * Hin * Win * Cin * (xraw ? 2 : 1)) * 2.0,
                        18.0 * po * Cin * N);
    if (tss_sconv_bwd_data(e, lde, yraw, ldyr, ga, gb, gce, gmu, w_tcn, xraw, ldx, in_mean, in_scale, in_bias, in_relu, e_in, ldei, bstats,
                           B, Hin, Win, Cin, N, (hipStream_t)stream))
      return tss::check_last("sconv_bwd_data");
  }
  GemmArgs g = {};
  g.Hin = (Hin - 1) / stride + 1; g.Win = (Win - 1) / stride + 1;      // the SOURCE grid of the gather: the layer's output
  g.Hout = Hin; g.Wout = Win; g.stride = 1; g.tstride = stride; g.dil = dil; g.tap_sign = -1; g.Cin = N; g.gkh = kh; g.gkw = kw;
  g.P = (long)B * Hin * Win; g.KD = N; g.ND = Cin; g.ntaps = kh * kw; g.mode = A_TAPS;
  g.a0 = e; g.lda0 = lde; g.a1 = yraw; g.lda1 = ldyr;
  if (yraw) { g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu; } else { g.c0 = ga; }
  g.w = w_tcn; g.wrs = N; g.wcs = 1; g.wts = (long)Cin * N;
  g.y = e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  return launch(g, dtype, TSS_K_CONV3X3_BWD_DATA, (hipStream_t)stream,
                ((double)B * g.Hin * g.Win * N * (yraw ? 2 : 1) + (double)g.P * Cin * (xraw ? 2 : 1)) * esz(dtype));
}

// nn.ConvTranspose2d(Cin_t, Cout, k, stride, padding = (k - 1) / 2, output_padding = stride - 1) (UpsamplingBlock, TSS/models/esnet.py:71-80):
// its forward IS the input-gradient gather of the strided convolution with the same weight tensor ([Cin_t][Cout][kh][kw] read as
// [N][Cin][taps]): x [B][Hout/stride][Wout/stride][Cin_t] -> y [B][Hout][Wout][Cout], w_tcn = [taps][Cout][Cin_t], bias added.
int tss_convkxk_transposed_fwd(const void* x, long ldx, const float* w_tcn, const float* bias, void* y, long ldy,
                               int B, int Hout, int Wout, int Cout, int Cin_t, int kh, int kw, int stride, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cout > 0 && Cin_t > 0 && (Cin_t % 8) == 0 && (Cout % 4) == 0 && (ldx % 8) == 0 && ldx >= Cin_t && (ldy % 4) == 0 && ldy >= Cout &&
              stride >= 1 && (Hout % stride) == 0 && (Wout % stride) == 0 && kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && kh * kw <= 81 && w_tcn,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && tss::aligned16(y), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.Hin = Hout / stride; g.Win = Wout / stride;
  g.Hout = Hout; g.Wout = Wout; g.stride = 1; g.tstride = stride; g.dil = 1; g.tap_sign = -1; g.Cin = Cin_t; g.gkh = kh; g.gkw = kw;
  g.P = (long)B * Hout * Wout; g.KD = Cin_t; g.ND = Cout; g.ntaps = kh * kw; g.mode = A_TAPS;
  g.a0 = x; g.lda0 = ldx;
  g.w = w_tcn; g.wrs = Cin_t; g.wcs = 1; g.wts = (long)Cout * Cin_t; g.bias = bias;
  g.y = y; g.ldy = ldy;
  if (dtype == TSS_BF16 && !g_tss_disable_fast && kh == 3 && kw == 3 && stride == 2) {      // parity-class kernel of sconv.hip (small channel counts)
    tss::ProfScope prof(TSS_K_CONV3X3_FWD, (hipStream_t)stream, ((double)B * g.Hin * g.Win * Cin_t + (double)g.P * Cout) * 2.0,
                        2.0 * (double)g.P * 2.25 * Cin_t * Cout);
    if (tss_sconv_transposed_fwd(x, ldx, w_tcn, bias, y, ldy, B, Hout, Wout, Cout, Cin_t, (hipStream_t)stream))
      return tss::check_last("sconv_transposed_fwd");
  }
  return launch(g, dtype, TSS_K_CONV3X3_FWD, (hipStream_t)stream, ((double)B * g.Hin * g.Win * Cin_t + (double)g.P * Cout) * esz(dtype));
}

int tss_get_option(int key) { return key == TSS_OPT_DISABLE_FAST_PATHS ? g_tss_disable_fast : -1; }

int tss_set_option(int key, int value) {
  if (key == TSS_OPT_DISABLE_FAST_PATHS) { g_tss_disable_fast = value; return TSS_OK; }
  return TSS_ERR_SHAPE;
}

int tss_stem3x3_fwd(const void* x_nchw, int x_is_f32, const float* w, void* y, long ldy, double* stats,
                    int B, int Cin, int Hin, int Win, int N, int stride, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin >= 1 && Cin * 9 <= Mma<float>::KC && N > 0 && (ldy % 4) == 0 && ldy >= N && stride >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(y), TSS_ERR_ALIGN);
  GemmArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = 1; g.tap_sign = 1; g.Cin = Cin;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  g.P = (long)B * g.Hout * g.Wout; g.KD = Cin * 9; g.ND = N; g.ntaps = 1; g.mode = A_STEM;
  g.a0 = x_nchw; g.a0_f32 = x_is_f32;
  g.w = w; g.wrs = (long)Cin * 9; g.wcs = 1; g.wts = 0;
  g.y = y; g.ldy = ldy; g.stats = stats;
  const double bytes = (double)B * Cin * Hin * Win * (x_is_f32 ? 4 : esz(dtype)) + (double)g.P * N * esz(dtype);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin <= 3 && N == 32) {   // performance path: direct VALU kernel (stem.hip)
    tss::ProfScope prof(TSS_K_STEM_FWD, (hipStream_t)stream, bytes, 2.0 * (double)g.P * g.KD * N);
    if (tss_stem_direct_fwd(x_nchw, x_is_f32, w, y, ldy, stats, B, Cin, Hin, Win, N, stride, dtype, (hipStream_t)stream))
      return tss::check_last("stem_direct_fwd");
  }
  return launch(g, dtype, TSS_K_STEM_FWD, (hipStream_t)stream, bytes);
}

}  // extern "C"

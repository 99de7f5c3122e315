// Lean bf16 dense 3x3 convolution (stride 1, padding = dilation <= 18): forward and backward-data of
// ConvBlock(128, 128, 3, padding=1) at the end of ContextNet's context branch (TSS/models/contextnet.py:55) and of the
// atrous branches (rates 6 / 12 / 18) of the ASPP head of BASELINE config 5 (models/aspp.py).
//
// The generic implicit-GEMM kernel (convgemm.hip, A_TAPS) walks the 9 taps as 9 dependent chunks: per tap it gathers a
// shifted 128-pixel tile from global memory, restages 64 KB of f32 weights and meets at three barriers -- ~11 us per
// tap on a 16 k-pixel map (110-130 us per launch for 2.4 GMAC).  Here
//   * a tile is 64 consecutive pixels of ONE image row; its 3 x 66-pixel halo is normalised (deferred BatchNorm + ReLU,
//     or the BatchNorm-backward combination of e and y) and written to LDS ONCE, zero outside the image -- the nine
//     taps are nine shifted views of that LDS tile, no further global reads of activations;
//   * weights come as bf16 [tap][output channel][contraction] (tss_permute_w3x3_bf16): a tap is 32 KB, staged with
//     16-byte copies into one of two LDS buffers while the previous tap's MFMAs run -- one barrier per tap;
//   * epilogue as in pwfast.hip: statistics from the bits that are stored, one slab row per block.
// K (contraction) in {32, 64, 128}; N (outputs) a multiple of 16 up to 128.  Everything else stays on convgemm_kernel.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256, TW = 64, RS = 128 + 8, NCH = 128;
constexpr int HALFPASS = 7;   // halo passes per batch at 16 pixels per pass (K = 128): dilation 1: ceil(198 / 16) = 13 <= 2 * 7;
                              // dilation D <= 18 (the ASPP rates 6 / 12 / 18): 3 (64 + 2 D) <= 300 pixels = 19 passes <= 3 * 7
constexpr int MAXDIL = 18;    // 3 x 100 halo pixels + two 128 x 128 weight taps = 152.7 KB of the 160 KB of LDS

struct C3Args {
  int B, H, W, K, N, D;                              // D = dilation (padding = dilation)
  const T* a0; long lda0; const T* a1; long lda1;    // fwd: x (a1 unused)   bwd: e, yraw
  const float* c0; const float* c1; const float* c2; const float* c3; int a_relu;
  const T* w9;                                       // [9][N][K]
  int tap_sign;                                      // +1: source pixel p + off(tap)   -1: p - off(tap)
  T* y; long ldy; double* stats;
  const T* xm; long ldxm; const float* mm; const float* ms; const float* mb; int m_relu;
};

__device__ __forceinline__ float bits_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bits_lo(uint32_t u) { return __uint_as_float(u << 16); }

template <bool BWD, int NB>
__global__ __launch_bounds__(NT, 1) void conv3x3_lean_kernel(const C3Args g) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int D = g.D, HC = TW + 2 * D, HP = 3 * HC;           // halo: 3 rows (y - D, y, y + D) x (64 + 2 D) pixels
  T* Xs = reinterpret_cast<T*>(smem);                        // [HP][RS]
  T* Ws = Xs + HP * RS;                                      // [2][NCH][RS]
  float* Ec = reinterpret_cast<float*>(Ws + 2 * NCH * RS);   // [3][NCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;                   // wave tile: 32 pixels x 64 channels
  const int K = g.K, N = g.N;
  const int nvec = K >> 3;                                   // 16-byte vectors per pixel / per weight row: 4, 8 or 16
  const int rpp = NT / nvec;                                 // pixels (or weight rows) per pass
  const int cv = tid & (nvec - 1), r = tid / nvec;
  const int npass = (HP + rpp - 1) / rpp;
  const int nks = K >> 5;
  int nfr = (N - wn * 64 + 15) >> 4;
  nfr = nfr < 0 ? 0 : (nfr > 4 ? 4 : nfr);
  const int tpr = (g.W + TW - 1) / TW;                       // tiles per image row
  const long ntiles = (long)g.B * g.H * tpr;

  if (tid < NCH) {
    const bool in = tid < N;
    const int nn = in ? tid : 0;
    const bool hm = BWD && g.xm && g.mm, hs = BWD && g.xm && g.ms, hb = BWD && g.xm && g.mb;
    const float* safe = reinterpret_cast<const float*>(g.w9);   // any readable, 16-byte aligned address
    const float e0 = (hm ? g.mm : safe)[hm ? nn : 0], e1 = (hs ? g.ms : safe)[hs ? nn : 0], e2 = (hb ? g.mb : safe)[hb ? nn : 0];
    Ec[tid] = (in && hm) ? e0 : 0.f;
    Ec[NCH + tid] = (in && hs) ? e1 : 1.f;
    Ec[2 * NCH + tid] = (in && hb) ? e2 : 0.f;
  }

  // this lane's prologue coefficients (its channel vector is fixed), mean folded into the additive term
  float k0[8], k1[8], kadd[8];
  const float relu_lo = g.a_relu ? 0.f : -TSS_INF;
  {
    const float* safe = reinterpret_cast<const float*>(g.w9);
    const float* p0c = g.c0 ? g.c0 + cv * 8 : safe;
    const float* p1c = g.c1 ? g.c1 + cv * 8 : safe; const float* p2c = g.c2 ? g.c2 + cv * 8 : safe;
    const float* p3c = (BWD && g.c3) ? g.c3 + cv * 8 : safe;
    float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(p0c + h, v0 + h); V4<float>::load(p1c + h, v1 + h);
      V4<float>::load(p2c + h, v2 + h); V4<float>::load(p3c + h, v3 + h);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float c0v = g.c0 ? v0[j] : 1.f, c1v = g.c1 ? v1[j] : 0.f, c2v = g.c2 ? v2[j] : 0.f, c3v = (BWD && g.c3) ? v3[j] : 0.f;
      if (BWD) { k0[j] = c0v; k1[j] = c1v; kadd[j] = -(c0v * c2v) - c1v * c3v; }   // g = c0*(e - c2) + c1*(y - c3)
      else     { k0[j] = c0v; k1[j] = 0.f; kadd[j] = c2v - c1v * c0v; }            // a = (x - c1)*c0 + c2
    }
  }

  float st1[4][4], st2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  // weight tap -> registers (8 rows per thread, 16-byte vectors; rows >= N re-read row 0 and are never stored)
  uint4 wreg[8];
  auto load_tap = [&](int tap) {
    const T* src = g.w9 + (long)tap * N * K + cv * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = u * rpp + r;
      wreg[u] = *reinterpret_cast<const uint4*>(src + (n < N ? n : 0) * K);
    }
  };
  auto store_tap = [&](T* dst) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = u * rpp + r;
      if (n < N) *reinterpret_cast<uint4*>(dst + n * RS + cv * 8) = wreg[u];
    }
  };

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = (int)(tile % tpr);
    const long by = tile / tpr;                              // b * H + y
    const int y = (int)(by % g.H);
    const long brow = by - y;                                // b * H
    const int x0 = tx * TW;
    const long pc = by * g.W + x0;                           // first pixel of the tile: always in the image
    __syncthreads();                                         // previous tile: MFMA reads of Xs / Ws, epilogue's Ec

    // ---- halo in two batches of <= 7 passes: loads first (clamped source, never out of bounds, never predicated), then
    // transform + store.  (One batch of 13 x 2 vectors does not fit the 256 architectural VGPRs next to the rest.)
    load_tap(0);
    // bwd: the producer's raw output under this lane's outputs (ReLU mask and statistics in the epilogue)
    uint2 rxm[4][2];
    if (BWD && g.xm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nfr) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int px = wm * 32 + m * 16 + fr;
            const long p = pc + ((x0 + px < g.W) ? px : 0);
            rxm[i][m] = *reinterpret_cast<const uint2*>(g.xm + p * g.ldxm + wn * 64 + i * 16 + fq * 4);
          }
        }
      }
    }
#pragma unroll
    for (int half = 0; half < NB; ++half) {
      uint4 ra[HALFPASS], rb[BWD ? HALFPASS : 1];
      bool okp[HALFPASS];
#pragma unroll
      for (int u = 0; u < HALFPASS; ++u) {
        const int ps = half * HALFPASS + u;
        if (ps < npass) {
          const int hp = ps * rpp + r;
          const int hr = hp / HC, hc = hp - hr * HC;
          const int iy = y + (hr - 1) * D, ix = x0 + hc - D;
          const bool ok = hp < HP && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
          okp[u] = ok;
          const long q = ok ? (brow + iy) * g.W + ix : pc;
          ra[u] = *reinterpret_cast<const uint4*>(g.a0 + q * g.lda0 + cv * 8);
          if (BWD) rb[u] = *reinterpret_cast<const uint4*>(g.a1 + q * g.lda1 + cv * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < HALFPASS; ++u) {
        const int ps = half * HALFPASS + u;
        if (ps < npass) {
          const int hp = ps * rpp + r;
          if (hp < HP) {
            const uint32_t* ua = reinterpret_cast<const uint32_t*>(&ra[u]);
            const uint32_t* ub = reinterpret_cast<const uint32_t*>(&rb[BWD ? u : 0]);
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
            if (!okp[u]) {                                   // zero padding applies to the ACTIVATED tensor
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            V8<T>::store(Xs + hp * RS + cv * 8, v);
          }
        }
      }
    }
    store_tap(Ws);
    __syncthreads();

    // ---- 9 taps: D[n][p] += W_tap[n][k] * A[p + off(tap)][k]
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) load_tap(tap + 1);                        // lands under this tap's MFMAs
      const int dy = g.tap_sign * (tap / 3 - 1), dx = g.tap_sign * (tap % 3 - 1);
      if (nfr > 0) {
        const T* xrow = Xs + ((dy + 1) * HC + wm * 32 + fr + (dx + 1) * D) * RS + fq * 8;
        const T* wrow = Ws + (tap & 1) * NCH * RS + (wn * 64 + fr) * RS + fq * 8;
        for (int ks = 0; ks < nks; ++ks) {
          bf16x8 xf[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xrow + m * 16 * RS + ks * 32);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < nfr) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + i * 16 * RS + ks * 32);
#pragma unroll
              for (int m = 0; m < 2; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[m][i], 0, 0, 0);
            }
          }
        }
      }
      // the other buffer was last read by tap - 1, and every wave has passed the barrier that closed it
      if (tap < 8) store_tap(Ws + ((tap + 1) & 1) * NCH * RS);
      __syncthreads();
    }

    // ---- epilogue: lane owns pixel (wm*32 + m*16 + fr) x channels (wn*64 + i*16 + fq*4 .. +3)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nfr) {
        const int nl = wn * 64 + i * 16 + fq * 4;
        const float4 e0 = *reinterpret_cast<const float4*>(Ec + nl);
        const float4 e1 = *reinterpret_cast<const float4*>(Ec + NCH + nl);
        const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * NCH + nl);
        const float cmm[4] = {e0.x, e0.y, e0.z, e0.w}, cms[4] = {e1.x, e1.y, e1.z, e1.w}, cmb[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int px = wm * 32 + m * 16 + fr;
          if (x0 + px < g.W) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q];
            bf16x4 o;
            if (BWD && g.xm) {
              const uint2 xr = rxm[i][m];
              const float xc[4] = {bits_lo(xr.x) - cmm[0], bits_hi(xr.x) - cmm[1], bits_lo(xr.y) - cmm[2], bits_hi(xr.y) - cmm[3]};
              if (g.m_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(xc[q] * cms[q] + cmb[q] > 0.f)) v[q] = 0.f;
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * xc[q]; }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
              for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
            }
            *reinterpret_cast<bf16x4*>(g.y + (pc + px) * g.ldy + nl) = o;
          }
        }
      }
    }
  }

  // ---- statistics: one slab row per block; rows no block owns are zeroed here (the caller never clears the buffer)
  if (g.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);             // [2 (wm)][2][NCH], aliases Xs
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
    if (tid < N) {
      const double a = (double)red[0 * NCH + tid] + (double)red[2 * NCH + tid];
      const double b = (double)red[1 * NCH + tid] + (double)red[3 * NCH + tid];
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

inline size_t smem_bytes(int dil) { return (size_t)(3 * (TW + 2 * dil) + 2 * NCH) * RS * sizeof(T) + 3 * NCH * sizeof(float); }

__global__ __launch_bounds__(256) void permute_w3x3_bf16_kernel(const float* w, T* w_tnc, T* w_tcn, int N, int Cin) {
  const long total = (long)N * Cin * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long nc = i / 9;
    const int c = (int)(nc % Cin), n = (int)(nc / Cin);
    const T v = (T)w[i];
    if (w_tnc) w_tnc[((long)tap * N + n) * Cin + c] = v;
    if (w_tcn) w_tcn[((long)tap * Cin + c) * N + n] = v;
  }
}

// Weight gradient of the dense 3x3: the activated input is unfolded once into [P][Cin*9] bf16 with column c*9 + tap
// (zero where the tap leaves the image), which makes dW[n][c][tap] = sum_p g[p][n] * col[p][c*9 + tap] a plain pointwise
// weight gradient with K = 9*Cin whose output layout IS torch's [N][Cin][3][3] -- the pipelined MFMA kernel of wgrad.hip
// does the rest (general tap loop: 228 us on the 8x128x32x64 map; unfold 37 MB + wgfast: see profiles/README.md).
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* x, long ldx, const float* mean, const float* scale,
                                                        const float* bias, int relu, T* col, int B, int H, int W, int C, int dil) {
  const long total = (long)B * H * W * C;
  const float lo = relu ? 0.f : -TSS_INF;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long p = i / C;
    const long pix = p;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long b = p / H;
    const float mu = scale ? mean[c] : 0.f, sc = scale ? scale[c] : 1.f, be = scale ? bias[c] : 0.f;
    T v[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {           // nine independent 2-byte loads (lanes run along the channels: coalesced)
      const int iy = yy + (t / 3 - 1) * dil, ix = xx + (t % 3 - 1) * dil;
      ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const long q = ok[t] ? (b * H + iy) * (long)W + ix : pix;
      v[t] = x[q * ldx + c];
    }
    T* dst = col + pix * ((long)C * 9) + (long)c * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) dst[t] = ok[t] ? (T)fmaxf(((float)v[t] - mu) * sc + be, lo) : (T)0.f;
  }
}

template <bool BWD, int NB>
void launch_lean_nb(const C3Args& g, hipStream_t stream) {
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_lean_kernel<BWD, NB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes(NB == 2 ? 1 : MAXDIL));
  }
  const long ntiles = (long)g.B * g.H * ((g.W + TW - 1) / TW);
  const int grid = (int)(ntiles < TSS_STAT_SLABS ? ntiles : TSS_STAT_SLABS);
  hipLaunchKernelGGL((conv3x3_lean_kernel<BWD, NB>), dim3(grid), dim3(NT), smem_bytes(g.D), stream, g);
}

template <bool BWD>
void launch_lean(const C3Args& g, hipStream_t stream) {
  if (g.D == 1) launch_lean_nb<BWD, 2>(g, stream); else launch_lean_nb<BWD, 3>(g, stream);
}

bool shape_ok(int K, int N, int stride, int dil) {
  return stride == 1 && dil >= 1 && dil <= MAXDIL && (K == 32 || K == 64 || K == 128) && N >= 16 && N <= NCH && (N % 16) == 0;
}

}  // namespace

// Called by tss_conv3x3_fwd / tss_conv3x3_bwd_data (convgemm.hip) when a bf16 tap-major weight copy is supplied.
// Return false: shape outside the lean kernel's domain, nothing launched.
bool tss_conv3x3_lean_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          const void* w9, void* y, long ldy, double* stats, int B, int H, int W, int Cin, int N,
                          int stride, int dil, hipStream_t stream) {
  if (!shape_ok(Cin, N, stride, dil) || (ldx % 8) != 0 || (ldy % 4) != 0 || (long)B * H * W == 0) return false;
  C3Args g = {};
  g.B = B; g.H = H; g.W = W; g.K = Cin; g.N = N; g.D = dil;
  g.a0 = (const T*)x; g.lda0 = ldx; g.c0 = in_scale; g.c1 = in_mean; g.c2 = in_bias; g.a_relu = in_relu;
  g.w9 = (const T*)w9; g.tap_sign = 1; g.y = (T*)y; g.ldy = ldy; g.stats = stats;
  launch_lean<false>(g, stream);
  return true;
}

bool tss_conv3x3_lean_bwd_data(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                               const float* gce, const float* gmu, const void* w9t, const void* xraw, long ldx,
                               const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                               void* e_in, long ldei, double* bstats, int B, int H, int W, int Cin, int N, int dil,
                               hipStream_t stream) {
  // contraction over the N channels of e, outputs = the Cin channels of e_in
  if (!shape_ok(N, Cin, 1, dil) || !yraw || !ga || (lde % 8) != 0 || (ldyr % 8) != 0 || (ldei % 4) != 0 ||
      (xraw && (ldx % 4) != 0) || (long)B * H * W == 0)
    return false;
  C3Args g = {};
  g.B = B; g.H = H; g.W = W; g.K = N; g.N = Cin; g.D = dil;
  g.a0 = (const T*)e; g.lda0 = lde; g.a1 = (const T*)yraw; g.lda1 = ldyr;
  g.c0 = ga; g.c1 = gb; g.c2 = gce; g.c3 = gmu;
  g.w9 = (const T*)w9t; g.tap_sign = -1; g.y = (T*)e_in; g.ldy = ldei; g.stats = bstats;
  g.xm = (const T*)xraw; g.ldxm = ldx; g.mm = in_mean; g.ms = in_scale; g.mb = in_bias; g.m_relu = in_relu;
  launch_lean<true>(g, stream);
  return true;
}

extern "C" int tss_im2col3x3(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                              void* col, int B, int H, int W, int C, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && ldx >= C && dil >= 1 && x && col, TSS_ERR_SHAPE);
  TSS_REQUIRE((in_mean != nullptr) == (in_scale != nullptr) && (in_bias != nullptr) == (in_scale != nullptr), TSS_ERR_SHAPE);
  const long total = (long)B * H * W * C;
  if (total == 0) return TSS_OK;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, in_mean, in_scale,
                     in_bias, in_relu, (T*)col, B, H, W, C, dil);
  return tss::check_last("im2col3x3");
}

extern "C" int tss_permute_w3x3_bf16(const float* w, void* w_tnc, void* w_tcn, int N, int Cin, void* stream) {
  TSS_REQUIRE(N > 0 && Cin > 0, TSS_ERR_SHAPE);
  const long total = (long)N * Cin * 9;
  long grid = (total + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(permute_w3x3_bf16_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, w, (T*)w_tnc, (T*)w_tcn, N, Cin);
  return tss::check_last("permute_w3x3_bf16");
}

// Weight-gradient contraction for the dense (non-depthwise) convolutions:
//
//   dW[n][k] += sum_p G(p, n) * A(p, k)
//     G(p, n) = ga[n]*(e[p][n]-ce[n]) + gb[n]*(y[p][n]-mu[n]) (deferred BN-backward of this conv's output)
//     A(p, k) = relu?((x[q(p)][k]-mean[k])*scale[k]+bias[k]) (deferred BN of the producer; q = p for 1x1,
//                                                             the tap-shifted pixel for 3x3, the NCHW gather
//                                                             for the 3-channel stem)
//
// The contraction index is the pixel, so both operands are needed channel-major.  Tiles are read
// NHWC-coalesced (16 B per lane) and transposed while they are written to LDS ([channel][pixel]
// rows of 128 B + 16 B pad, 16-byte chunks XOR-swizzled by channel group so that neither the 2/4-byte
// transposing stores nor the 16-byte MFMA operand reads pile up on one bank).  Each of the 4 waves
// owns a 64x64 block of the 128x128 dW tile (16 MFMA accumulators); blocks split the pixel range and
// add their partial tile to dW with f32 atomics (dW must be zero-initialised or hold a value to
// accumulate onto, exactly like autograd's .grad).
#include "common.h"
#include "wgreduce.h"
#include "bnfin.h"

extern int g_tss_disable_fast;   // pwfast.hip

namespace {

constexpr int TN = 128, TK = 128, NT = 256;
enum { A_PW = 0, A_TAPS = 1, A_STEM = 2 };

template <typename T> struct WMma;
template <> struct WMma<bf16_t> {
  static constexpr int PT = 64, KSTEP = 32;
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag ld(const unsigned char* row, int swz, int ks, int q) {
    return *reinterpret_cast<const bf16x8*>(row + (((ks * 4 + q) ^ swz) << 4));
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct WMma<float> {
  static constexpr int PT = 32, KSTEP = 4;
  typedef float Frag;
  static __device__ __forceinline__ Frag ld(const unsigned char* row, int swz, int ks, int q) {
    return *reinterpret_cast<const float*>(row + ((ks ^ swz) << 4) + q * 4);
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
constexpr int ROWB = 144;  // bytes per LDS row: 128 B of pixels + 16 B pad (both dtypes)

struct WgradArgs {
  long P;
  int ND, KD, ntaps, mode;
  const void* e; long lde; const void* yraw; long ldyr;
  const float* ga; const float* gb; const float* gce; const float* gmu;
  const void* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu, x_f32;
  int Hin, Win, Hout, Wout, stride, dil, Cin;
  int tap0, tstep1;               // as in convgemm.hip GemmArgs: which taps of the 3x3 grid the ntaps local taps are
  int gkh, gkw;                   // gkw > 0: a gkh x gkw tap grid instead (odd sides, centred)
  float* dw; long drs, dcs, dts;  // dW element (n, k, tap) at dw[n*drs + k*dcs + tap*dts]
  int nsplit;
  float* ws;                       // wgfast: partial tiles [tile][nsplit][ws_dim(ND)*ws_dim(KD)] (NULL: atomics onto dw)
  // wgfast: the first fin_blocks blocks of the grid finalize the BatchNorm backward of ANOTHER layer (the one the backward
  // pass reaches next; its slab rows are complete, its coefficients are not needed by this launch): no launch of its own
  tss_bn_bwd_job fin; int fin_blocks;
};

// transposing store of a 4-pixel x 8-channel unit: rows ch0..ch0+7, columns 4*pg .. 4*pg+3 (one 8/16-byte
// store per channel instead of four 2/4-byte ones)
template <typename T> struct Pack4;
template <> struct Pack4<bf16_t> {
  static __device__ __forceinline__ void put(unsigned char* dst, float a, float b, float c, float d) {
    bf16x4 o; o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
    *reinterpret_cast<bf16x4*>(dst) = o;
  }
};
template <> struct Pack4<float> {
  static __device__ __forceinline__ void put(unsigned char* dst, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
  }
};
template <typename T>
__device__ __forceinline__ unsigned char* unit_addr(unsigned char* tile, int row, int pg) {
  const int boff = pg * 4 * (int)sizeof(T);
  return tile + row * ROWB + ((((boff >> 4)) ^ ((row >> 3) & 7)) << 4) + (boff & 15);
}
template <typename T>
__device__ __forceinline__ void put_unit(unsigned char* tile, int ch0, int pg, const float v[4][8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) Pack4<T>::put(unit_addr<T>(tile, ch0 + j, pg), v[0][j], v[1][j], v[2][j], v[3][j]);
}

template <typename T>
__global__ __launch_bounds__(NT, 2) void wgrad_kernel(const WgradArgs g) {
  typedef WMma<T> M;
  constexpr int PT = M::PT, NPG = PT / 4;
  constexpr bool FOLD = sizeof(T) == 2;   // bf16: BatchNorm constants folded into one FMA per element
  __shared__ __align__(16) unsigned char Gt[TN * ROWB];
  __shared__ __align__(16) unsigned char At[TK * ROWB];
  __shared__ __align__(16) float Cg[4][TN];  // ga, gb, gce, gmu of this N chunk
  __shared__ __align__(16) float Ca[3][TK];  // mean, scale, bias of this K chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;

  const int nchn = (g.ND + TN - 1) / TN, nchk = (g.KD + TK - 1) / TK;
  int bid = blockIdx.x;
  const int split = bid % g.nsplit; bid /= g.nsplit;
  const int nc = bid % nchn; bid /= nchn;
  const int kc = bid % nchk; bid /= nchk;
  const int tap = bid;

  const int n0 = nc * TN, k0 = kc * TK;
  const int ncw = (g.ND - n0 < TN) ? (g.ND - n0) : TN;
  const int kcw = (g.KD - k0 < TK) ? (g.KD - k0) : TK;
  const int nvn = (ncw + 7) >> 3, nvk = (kcw + 7) >> 3;
  // MFMA fragments this wave actually owns (its 64x64 quadrant may lie partly or wholly outside the tile)
  int nfn = (ncw - wn * 64 + 15) >> 4; nfn = nfn < 0 ? 0 : (nfn > 4 ? 4 : nfn);
  int nfk = (kcw - wk * 64 + 15) >> 4; nfk = nfk < 0 ? 0 : (nfk > 4 ? 4 : nfk);

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + g.nsplit - 1) / g.nsplit;
  const long s_begin = split * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // zero both tiles once (rows / chunks that are never staged must read as 0) and stage the coefficients
  for (int i = tid; i < TN * ROWB / 16; i += NT) {
    reinterpret_cast<uint4*>(Gt)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(At)[i] = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < TN; i += NT) {
    const bool in = i < ncw;
    Cg[0][i] = (in && g.ga) ? g.ga[n0 + i] : 1.f;
    Cg[1][i] = (in && g.yraw) ? g.gb[n0 + i] : 0.f;
    Cg[2][i] = (in && g.yraw) ? g.gce[n0 + i] : 0.f;
    Cg[3][i] = (in && g.yraw) ? g.gmu[n0 + i] : 0.f;
    const bool ik = i < kcw && g.mode != A_STEM;
    Ca[0][i] = (ik && g.xs && g.xm) ? g.xm[k0 + i] : 0.f;
    Ca[1][i] = (ik && g.xs) ? g.xs[k0 + i] : 1.f;
    Ca[2][i] = (ik && g.xs && g.xb) ? g.xb[k0 + i] : 0.f;
  }

  const T* e = reinterpret_cast<const T*>(g.e);
  const T* yr = reinterpret_cast<const T*>(g.yraw);
  const T* x = reinterpret_cast<const T*>(g.x);
  const long HWo = (long)g.Hout * g.Wout;
  int tdy, tdx;                   // this tap's offset from the centre, in input pixels
  if (g.gkw > 0) { const int ty = tap / g.gkw; tdy = ty - (g.gkh >> 1); tdx = (tap - ty * g.gkw) - (g.gkw >> 1); }
  else { const int tgrid = g.tap0 + tap * (g.tstep1 + 1); const int ky = tgrid / 3; tdy = ky - 1; tdx = tgrid - ky * 3 - 1; }
  tdy *= g.dil; tdx *= g.dil;

  for (long s = s_begin; s < s_end; ++s) {
    const long p0 = s * PT;
    __syncthreads();
    // ---- G tile: units of 4 pixels x 8 channels
    for (int u = tid; u < NPG * nvn; u += NT) {
      const int pg = u / nvn, cv = u - pg * nvn;
      const int ch = n0 + cv * 8;
      float v[4][8];
      float ca[8], cb[8], cc[8], cm[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ca[j] = Cg[0][cv * 8 + j]; cb[j] = Cg[1][cv * 8 + j]; cc[j] = Cg[2][cv * 8 + j]; cm[j] = Cg[3][cv * 8 + j];
        if (FOLD) cc[j] = -(ca[j] * cc[j]) - cb[j] * cm[j];   // bf16: g = ga*e + gb*y + folded constant
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + pg * 4 + i;
        const bool ok = p < g.P;
        const long pc = ok ? p : 0;
        float ev[8];
        V8<T>::load(e + pc * g.lde + ch, ev);
        if (yr) {
          float yv[8];
          V8<T>::load(yr + pc * g.ldyr + ch, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[i][j] = FOLD ? ca[j] * ev[j] + (cb[j] * yv[j] + cc[j]) : ca[j] * (ev[j] - cc[j]) + cb[j] * (yv[j] - cm[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = ca[j] * ev[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!ok || ch + j >= g.ND) v[i][j] = 0.f;
      }
      put_unit<T>(Gt, cv * 8, pg, v);
    }
    // ---- A tile
    if (g.mode == A_STEM) {
      const float* src32 = reinterpret_cast<const float*>(g.x);
      for (int u = tid; u < NPG * kcw; u += NT) {
        const int pg = u % NPG, j = u / NPG;
        const int c = j / 9, t9 = j - c * 9, sy = t9 / 3, sx = t9 - sy * 3;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long p = p0 + pg * 4 + i;
          v[i] = 0.f;
          if (p < g.P) {
            const long b = p / HWo; const long rem = p - b * HWo;
            const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
            const int iy = oy * g.stride + (sy - 1) * g.dil, ix = ox * g.stride + (sx - 1) * g.dil;
            if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) {
              const long off = ((b * g.Cin + c) * g.Hin + iy) * (long)g.Win + ix;
              v[i] = g.x_f32 ? src32[off] : (float)x[off];
            }
          }
        }
        Pack4<T>::put(unit_addr<T>(At, j, pg), v[0], v[1], v[2], v[3]);
      }
    } else {
      for (int u = tid; u < NPG * nvk; u += NT) {
        const int pg = u / nvk, cv = u - pg * nvk;
        const int ch = k0 + cv * 8;
        float v[4][8];
        float cm[8], cs[8], cb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          cm[j] = Ca[0][cv * 8 + j]; cs[j] = Ca[1][cv * 8 + j]; cb[j] = Ca[2][cv * 8 + j];
          if (FOLD) cb[j] -= cm[j] * cs[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long p = p0 + pg * 4 + i;
          long q = -1;
          if (p < g.P) {
            if (g.mode == A_PW) {
              q = p;
            } else {
              const long b = p / HWo; const long rem = p - b * HWo;
              const int oy = (int)(rem / g.Wout), ox = (int)(rem - (long)oy * g.Wout);
              const int iy = oy * g.stride + tdy, ix = ox * g.stride + tdx;
              if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) q = (b * g.Hin + iy) * (long)g.Win + ix;
            }
          }
          float xv[8];
          V8<T>::load(x + (q >= 0 ? q : 0) * g.ldx + ch, xv);
          const float vlo = q >= 0 ? (g.x_relu ? 0.f : -TSS_INF) : 0.f, vhi = q >= 0 ? TSS_INF : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[i][j] = clamp3(FOLD ? xv[j] * cs[j] + cb[j] : (xv[j] - cm[j]) * cs[j] + cb[j], vlo, vhi);
            if (ch + j >= g.KD) v[i][j] = 0.f;
          }
        }
        put_unit<T>(At, cv * 8, pg, v);
      }
    }
    __syncthreads();
    // ---- MFMA over the PT pixels of this stage
    if (nfn > 0 && nfk > 0) {
      const unsigned char* grow = Gt + (wn * 64 + fr) * ROWB;
      const unsigned char* arow = At + (wk * 64 + fr) * ROWB;
#pragma unroll
      for (int ks = 0; ks < PT / M::KSTEP; ++ks) {
        typename M::Frag gf[4], af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // swizzle key = (row >> 3) & 7 with row = w*64 + i*16 + fr
          gf[i] = M::ld(grow + i * 16 * ROWB, ((wn * 64 + i * 16 + fr) >> 3) & 7, ks, fq);
          af[i] = M::ld(arow + i * 16 * ROWB, ((wk * 64 + i * 16 + fr) >> 3) & 7, ks, fq);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < nfn && j < nfk) acc[i][j] = M::mma(gf[i], af[j], acc[i][j]);
      }
    }
  }

  // ---- add the partial tile: D[row = n][col = k]
  float* dwt = g.dw + tap * g.dts;
  if (s_begin < s_end) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i < nfn && j < nfk) {
          const int k = k0 + wk * 64 + j * 16 + fr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4 + r;
            if (n < g.ND && k < g.KD) atomicAdd(dwt + (long)n * g.drs + (long)k * g.dcs, acc[i][j][r]);
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Lean bf16 kernel for the 1x1 layers (the performance path).  Same contraction and LDS layout as wgrad_kernel, but
//   * software-pipelined: the 16-byte loads of stage s+1 (e, y, x: one 4-pixel x 8-channel unit of each per thread)
//     are issued right after stage s went to LDS and land under its MFMAs; the generic kernel exposes two memory
//     round trips and two barriers per 64 pixels (rocprofv3: ~77 % of its wave cycles are waits);
//   * the unit of a thread is fixed, so its folded BatchNorm constants stay in registers and its addresses advance by
//     a constant per stage;
//   * the four waves are assigned by tile shape: a tile narrower than 5 fragments in n or k is not split along that
//     axis (those waves would idle) but along the pixel axis instead (each wave takes half of the k-steps);
//   * PT = 128-pixel stages when both chunk widths are <= 64 channels, so that the staging still uses every thread.
using tss_wg::ws_dim;

#ifdef TSS_TIMING
// phase timing of wgfast_kernel (debug builds only: python -m ...build with TSS_TIMING=1): cycles summed over wave 0 of
// every block for [barrier-in, G half (load wait + transform), A half, barrier-out, MFMA, prologue, atomics tail]
__device__ unsigned long long g_wg_timing[8];
#define TSS_T(var) unsigned long long var; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define TSS_T(var)
#endif
template <int PT>
__global__ __launch_bounds__(NT, 2) void wgfast_kernel(const WgradArgs g) {
  typedef bf16_t T;
  constexpr int NPG = PT / 4, ROW = PT * 2 + 16, NKS = PT / 32;
  TSS_T(tks);
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* Gt = smem;
  unsigned char* At = smem + TN * ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  if ((int)blockIdx.x < g.fin_blocks) { tss_fin::bn_bwd_finalize_block(g.fin, blockIdx.x); return; }
  const int nchn = (g.ND + TN - 1) / TN;
  int bid = (int)blockIdx.x - g.fin_blocks;
  const int split = bid % g.nsplit; bid /= g.nsplit;
  const int nc = bid % nchn; bid /= nchn;
  const int kc = bid;
  const int n0 = nc * TN, k0 = kc * TK;
  const int ncw = (g.ND - n0 < TN) ? (g.ND - n0) : TN;   // multiples of 8 (host-checked)
  const int kcw = (g.KD - k0 < TK) ? (g.KD - k0) : TK;
  const int nvn = ncw >> 3, nvk = kcw >> 3;
  const int FN = (ncw + 15) >> 4, FK = (kcw + 15) >> 4;

  // ---- role of this wave: fragment window [ib, ib+cn) x [jb, jb+ck), k-steps [ks0, ks1)
  int ib = 0, jb = 0, cn = 4, ck = 4, ks0 = 0, ks1 = NKS;
  if (FN > 4 && FK > 4) { ib = 4 * (wave >> 1); jb = 4 * (wave & 1); }
  else {
    ks0 = (wave & 1) * (NKS / 2); ks1 = ks0 + NKS / 2;
    if (FN > 4) ib = 4 * (wave >> 1);
    else if (FK > 4) jb = 4 * (wave >> 1);
    else if (FN > 2) { ib = 2 * (wave >> 1); cn = 2; }
    else { jb = 2 * (wave >> 1); ck = 2; }
  }
  cn = FN - ib < cn ? FN - ib : cn; cn = cn < 0 ? 0 : cn;
  ck = FK - jb < ck ? FK - jb : ck; ck = ck < 0 ? 0 : ck;

  const long nstage = (g.P + PT - 1) / PT;
  const long per = (nstage + g.nsplit - 1) / g.nsplit;
  const long s_begin = split * per;
  long s_end = s_begin + per;
  if (s_end > nstage) s_end = nstage;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- this thread's units: one 4-pixel x 8-channel unit of each operand per stage
  const int pgG = tid / nvn, cvG = tid - pgG * nvn;
  const int pgA = tid / nvk, cvA = tid - pgA * nvk;
  const bool onG = pgG < NPG, onA = pgA < NPG;
  const T* eg = reinterpret_cast<const T*>(g.e) + n0 + (onG ? cvG * 8 : 0);
  const T* yg = reinterpret_cast<const T*>(g.yraw) + n0 + (onG ? cvG * 8 : 0);
  const T* xg = reinterpret_cast<const T*>(g.x) + k0 + (onA ? cvA * 8 : 0);
  uint4 re[4], ry[4], rx[4];
  auto issue_g = [&](long s) {
    const long p0 = s * PT;
    if (onG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + pgG * 4 + i;
        const long pc = p < g.P ? p : 0;       // clamp + zero afterwards: no predicated loads
        re[i] = *reinterpret_cast<const uint4*>(eg + pc * g.lde);
        ry[i] = *reinterpret_cast<const uint4*>(yg + pc * g.ldyr);
      }
    }
  };
  auto issue_a = [&](long s) {
    const long p0 = s * PT;
    if (onA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + pgA * 4 + i;
        const long pc = p < g.P ? p : 0;
        rx[i] = *reinterpret_cast<const uint4*>(xg + pc * g.ldx);
      }
    }
  };
  auto unit_ptr = [&](unsigned char* tile, int row, int pg) -> unsigned char* {
    const int boff = pg * 8;
    return tile + row * ROW + ((((boff >> 4)) ^ ((row >> 3) & 7)) << 4) + (boff & 15);
  };
  auto blo = [](uint32_t u) { return __uint_as_float(u << 16); };
  auto bhi = [](uint32_t u) { return __uint_as_float(u & 0xffff0000u); };

#ifdef TSS_TIMING
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  // first stage's loads go out before anything else: the block's set-up (LDS clear, constants) runs under them
  if (s_begin < s_end) { issue_g(s_begin); issue_a(s_begin); }

  for (int i = tid; i < TN * ROW / 16; i += NT) {   // rows that are never staged must read as 0
    reinterpret_cast<uint4*>(Gt)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(At)[i] = make_uint4(0, 0, 0, 0);
  }
  // folded constants of this thread's channel vectors: unconditional 16-byte loads (a `ptr ? ptr[i] : c` per element
  // compiles to a branch + wait per load: ~15 us of dependent round trips per block before this was hoisted)
  float ca[8], cb[8], cc[8], as[8], ab[8];
  {
    const int chn = n0 + (onG ? cvG * 8 : 0), chk = k0 + (onA ? cvA * 8 : 0);
    const float* pxs = g.xs ? g.xs : g.ga;      // any readable address when the input has no pending BatchNorm
    const float* pxb = (g.xs && g.xb) ? g.xb : g.ga;
    const float* pxm = (g.xs && g.xm) ? g.xm : g.ga;
    const int cks = g.xs ? chk : 0;
    float vga[8], vgb[8], vce[8], vmu[8], vs[8], vb[8], vm[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      V4<float>::load(g.ga + chn + h, vga + h); V4<float>::load(g.gb + chn + h, vgb + h);
      V4<float>::load(g.gce + chn + h, vce + h); V4<float>::load(g.gmu + chn + h, vmu + h);
      V4<float>::load(pxs + cks + h, vs + h); V4<float>::load(pxb + cks + h, vb + h); V4<float>::load(pxm + cks + h, vm + h);
    }
    const bool has_s = g.xs != nullptr, has_b = has_s && g.xb != nullptr, has_m = has_s && g.xm != nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ca[j] = vga[j]; cb[j] = vgb[j]; cc[j] = -(vga[j] * vce[j]) - vgb[j] * vmu[j];      // g = ga*e + gb*y + cc
      const float sc = has_s ? vs[j] : 1.f;
      as[j] = sc; ab[j] = (has_b ? vb[j] : 0.f) - (has_m ? vm[j] : 0.f) * sc;             // a = relu?(x*as + ab)
    }
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;
  TSS_T(tk1);
#ifdef TSS_TIMING
  tph[5] += tk1 - tks;
#endif
  for (long s = s_begin; s < s_end; ++s) {
    const long p0 = s * PT;
    TSS_T(t0);
    __syncthreads();   // MFMAs of the previous stage have read the tiles (first pass: the zero fill is complete)
    TSS_T(t1);
    if (onG) {
      float v[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = p0 + pgG * 4 + i < g.P;
        const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re[i]);
        const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry[i]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[i][2 * h] = ca[2 * h] * blo(ue[h]) + (cb[2 * h] * blo(uy[h]) + cc[2 * h]);
          v[i][2 * h + 1] = ca[2 * h + 1] * bhi(ue[h]) + (cb[2 * h + 1] * bhi(uy[h]) + cc[2 * h + 1]);
        }
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Pack4<T>::put(unit_ptr(Gt, cvG * 8 + j, pgG), v[0][j], v[1][j], v[2][j], v[3][j]);
    }
    TSS_T(t2);
    // the G registers are free again: their next-stage loads go out before the A half is processed, so the memory
    // pipe is never idle for this block (each half's loads fly under the other half's arithmetic and the MFMAs)
    if (s + 1 < s_end) issue_g(s + 1);
    if (onA) {
      float v[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = p0 + pgA * 4 + i < g.P;
        const uint32_t* ux = reinterpret_cast<const uint32_t*>(&rx[i]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[i][2 * h] = fmaxf(blo(ux[h]) * as[2 * h] + ab[2 * h], relu_lo);
          v[i][2 * h + 1] = fmaxf(bhi(ux[h]) * as[2 * h + 1] + ab[2 * h + 1], relu_lo);
        }
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Pack4<T>::put(unit_ptr(At, cvA * 8 + j, pgA), v[0][j], v[1][j], v[2][j], v[3][j]);
    }
    if (s + 1 < s_end) issue_a(s + 1);
    TSS_T(t3);
    __syncthreads();
    TSS_T(t4);

    if (cn > 0 && ck > 0) {
      const int rg = ib * 16 + fr, rk = jb * 16 + fr;
      const unsigned char* grow = Gt + rg * ROW;
      const unsigned char* arow = At + rk * ROW;
      for (int ks = ks0; ks < ks1; ++ks) {
        bf16x8 gf[4], af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < cn) gf[i] = *reinterpret_cast<const bf16x8*>(grow + i * 16 * ROW + (((ks * 4 + fq) ^ (((rg + i * 16) >> 3) & 7)) << 4));
          if (i < ck) af[i] = *reinterpret_cast<const bf16x8*>(arow + i * 16 * ROW + (((ks * 4 + fq) ^ (((rk + i * 16) >> 3) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < cn && j < ck) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], af[j], acc[i][j], 0, 0, 0);
      }
    }
    TSS_T(t5);
#ifdef TSS_TIMING
    tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; tph[4] += t5 - t4;
#endif
  }

  TSS_T(tkl);
  if (g.ws) {
    // partial tile -> this block's workspace slot, one 16x16 MFMA fragment after the other (plain stores; blocks without stages
    // write their zeros); wg_reduce_kernel sums the slots.  Measured on the 128x128 layer at 1/8 resolution: the f32
    // atomics of 512 blocks onto one 64 KB tile cost 42 us of an 86 us kernel whose streaming loop already runs at
    // HBM speed.  Waves that split the pixels of a stage (same fragment window, other k-steps) first meet in LDS.
    const bool psplit = !(FN > 4 && FK > 4);
    if (psplit) {
      __syncthreads();                                   // the tiles are dead: reuse them as the exchange buffer
      f32x4* xch = reinterpret_cast<f32x4*>(smem) + (wave >> 1) * 16 * 64;
      if (wave & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < cn && j < ck) xch[(i * 4 + j) * 64 + lane] = acc[i][j];
      }
      __syncthreads();
      if (!(wave & 1)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < cn && j < ck) acc[i][j] += xch[(i * 4 + j) * 64 + lane];
      }
    }
    if (!psplit || !(wave & 1)) {
      const int tile = kc * nchn + nc;
      const int TNe = ws_dim(g.ND), TKe = ws_dim(g.KD);     // slot = the largest tile of this layer, 16-padded
      float* wt = g.ws + ((long)tile * g.nsplit + split) * (TNe * TKe);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (i < cn && j < ck) {   // fragment-major slot layout (wgreduce.h): every store instruction writes 256 contiguous bytes
            float* wf = wt + ((ib + i) * (TKe >> 4) + (jb + j)) * 256 + fq * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) wf[r * 64] = acc[i][j][r];
          }
        }
    }
  } else if (s_begin < s_end) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i < cn && j < ck) {
          const int k = k0 + (jb + j) * 16 + fr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + (ib + i) * 16 + fq * 4 + r;
            if (n < g.ND && k < g.KD) atomicAdd(g.dw + (long)n * g.drs + (long)k * g.dcs, acc[i][j][r]);
          }
        }
      }
  }
#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSS_T(tk2);
  tph[6] = tk2 - tkl;
  if (threadIdx.x == 0) {
    for (int q = 0; q < 7; ++q) atomicAdd(&g_wg_timing[q], tph[q]);
    atomicAdd(&g_wg_timing[7], 1ull);
  }
#endif
}

// stand-alone form of the slot reduction (wgreduce.h); normally these blocks ride on the layer's backward-data launch
__global__ __launch_bounds__(NT) void wg_reduce_kernel(const tss_wg::ReduceArgs r) {
  __shared__ float4 part[4 * 64];
  tss_wg::reduce_block(r, blockIdx.x, part);
}

using tss_wg::reduce_args;

template <int PT>
void launch_fast_pt(WgradArgs& g, hipStream_t stream, bool defer_reduce) {
  const tss_wg::Split sp = tss_wg::split_for(g.P, g.KD, g.ND);
  g.nsplit = sp.nsplit;
  constexpr int smem = 2 * TN * (PT * 2 + 16);
  static tss::DevOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgfast_kernel<PT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  }
  hipLaunchKernelGGL(wgfast_kernel<PT>, dim3(sp.tiles * g.nsplit + g.fin_blocks), dim3(NT), smem, stream, g);
  if (g.ws && !defer_reduce) {
    const tss_wg::ReduceArgs r = reduce_args(g.ws, g.dw, g.P, g.KD, g.ND);
    hipLaunchKernelGGL(wg_reduce_kernel, dim3(r.nred), dim3(NT), 0, stream, r);
  }
}

int launch(WgradArgs& g, int dtype, int kernel_id, hipStream_t stream, double alg_bytes) {
  if (g.P <= 0) return TSS_OK;
  const int nchn = (g.ND + TN - 1) / TN, nchk = (g.KD + TK - 1) / TK;
  const int tiles = nchn * nchk * g.ntaps;
  const int PT = dtype == TSS_BF16 ? WMma<bf16_t>::PT : WMma<float>::PT;
  const long nstage = (g.P + PT - 1) / PT;
  // blocks per output tile: enough to fill the chip, but every block must amortise its fixed cost (LDS clear,
  // 16 K f32 atomics for a full tile) over at least MIN_STAGES pixel stages
  constexpr long MIN_STAGES = 8;
  long ns = 1024 / tiles;
  if (ns < 1) ns = 1;
  if (ns > (nstage + MIN_STAGES - 1) / MIN_STAGES) ns = (nstage + MIN_STAGES - 1) / MIN_STAGES;
  if (ns < 1) ns = 1;
  g.nsplit = (int)ns;
  const int grid = tiles * (int)ns;
  tss::ProfScope prof(kernel_id, stream, alg_bytes, 2.0 * (double)g.P * g.ND * g.KD * g.ntaps);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(wgrad_kernel<bf16_t>, dim3(grid), dim3(NT), 0, stream, g);
  else hipLaunchKernelGGL(wgrad_kernel<float>, dim3(grid), dim3(NT), 0, stream, g);
  return tss::check_last("wgrad");
}

inline size_t esz(int dtype) { return dtype == TSS_BF16 ? 2 : 4; }

}  // namespace

bool tss_stem_direct_wgrad(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb,
                           const float* gce, const float* gmu, const void* x_nchw, int x_is_f32, float* dw, float* ws,
                           int B, int Cin, int Hin, int Win, int N, int stride, int dtype, hipStream_t stream);  // stem.hip

// pending slot reduction of a layer, launched on its own (pwfast.hip calls this when its kernels cannot carry it)
void tss_wg_reduce_standalone(const float* ws, float* dw, long P, int K, int N, hipStream_t stream) {
  const tss_wg::ReduceArgs r = reduce_args(ws, dw, P, K, N);
  hipLaunchKernelGGL(wg_reduce_kernel, dim3(r.nred), dim3(NT), 0, stream, r);
}

extern "C" {

int tss_pwconv_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                          const float* ga, const float* gb, const float* gce, const float* gmu,
                          const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          float* dw, float* ws, int defer_reduce, long P, int K, int N, int dtype, const tss_bn_bwd_job* fin,
                          void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(!fin || (fin->C > 0 && fin->count >= 1.0 && fin->bstats && fin->invstd && fin->ga && fin->gb && fin->gce), TSS_ERR_SHAPE);
  TSS_REQUIRE(K > 0 && N > 0 && (K % 8) == 0 && (lde % 8) == 0 && lde >= (N + 7) / 8 * 8 && (ldx % 8) == 0 && ldx >= K, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= (N + 7) / 8 * 8 && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(xraw), TSS_ERR_ALIGN);
  WgradArgs g = {};
  g.P = P; g.ND = N; g.KD = K; g.ntaps = 1; g.mode = A_PW;
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.Hout = 1; g.Wout = 1;
  g.dw = dw; g.drs = K; g.dcs = 1; g.dts = 0;
  const double bytes = (double)P * (N * (yraw ? 2 : 1) + K) * esz(dtype);
  if (dtype == TSS_BF16 && !g_tss_disable_fast && yraw && (N % 8) == 0 && P > 0) {   // lean pipelined kernel
    tss::ProfScope prof(TSS_K_PWCONV_BWD_WEIGHT, (hipStream_t)stream, bytes, 2.0 * (double)P * N * K);
    g.ws = ws;
    if (fin) { g.fin = *fin; g.fin_blocks = tss_fin::fin_blocks(fin->C); }
    if (tss_wg::split_for(P, K, N).pt == 128) launch_fast_pt<128>(g, (hipStream_t)stream, ws && defer_reduce);
    else launch_fast_pt<64>(g, (hipStream_t)stream, ws && defer_reduce);
    return tss::check_last("wgfast");
  }
  if (fin) {   // the general kernel carries nothing: the finalize runs as the launch of its own it would otherwise have been
    const int rc = tss_bn_bwd_finalize(fin->bstats, fin->count, fin->invstd, fin->gamma, fin->training, fin->accumulate, fin->dgamma,
                                       fin->dbeta, fin->ga, fin->gb, fin->gce, fin->C, stream);
    if (rc != TSS_OK) return rc;
  }
  return launch(g, dtype, TSS_K_PWCONV_BWD_WEIGHT, (hipStream_t)stream, bytes);
}

int tss_pwconv_wg_reduce(const float* ws, float* dw, long P, int K, int N, void* stream) {
  TSS_REQUIRE(ws && dw && P > 0 && K > 0 && N > 0 && (K % 8) == 0 && (N % 8) == 0, TSS_ERR_SHAPE);
  tss_wg_reduce_standalone(ws, dw, P, K, N, (hipStream_t)stream);
  return tss::check_last("wg_reduce");
}

long tss_pwconv_bwd_weight_ws(long P, int K, int N, int dtype) {
  if (dtype != TSS_BF16 || g_tss_disable_fast || (N % 8) != 0 || (K % 8) != 0 || P <= 0) return 0;
  const tss_wg::Split sp = tss_wg::split_for(P, K, N);
  return (long)sp.tiles * sp.nsplit * ws_dim(N) * ws_dim(K);
}

int tss_conv3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int Hin, int Win, int Cin, int N, int stride, int dil,
                           int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (N % 8) == 0 && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin,
              TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  WgradArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = dil; g.Cin = Cin;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  g.P = (long)B * g.Hout * g.Wout; g.ND = N; g.KD = Cin; g.ntaps = 9; g.mode = A_TAPS;
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.dw = dw; g.drs = (long)Cin * 9; g.dcs = 9; g.dts = 1;  // torch layout [N][Cin][3][3]
  return launch(g, dtype, TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream,
                ((double)g.P * N * (yraw ? 2 : 1) + (double)B * Hin * Win * Cin) * esz(dtype));
}

int tss_conv1d3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (N % 8) == 0 && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin &&
              (axis == 0 || axis == 1) && dil >= 1, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  WgradArgs g = {};
  g.Hin = H; g.Win = W; g.stride = 1; g.dil = dil; g.Cin = Cin; g.Hout = H; g.Wout = W;
  g.tap0 = axis == 0 ? 3 : 1; g.tstep1 = axis == 0 ? 0 : 2;
  g.P = (long)B * H * W; g.ND = N; g.KD = Cin; g.ntaps = 3; g.mode = A_TAPS;
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.dw = dw; g.drs = (long)Cin * 3; g.dcs = 3; g.dts = 1;  // torch layout [N][Cin][1][3] / [N][Cin][3][1]
  return launch(g, dtype, TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream,
                ((double)g.P * N * (yraw ? 2 : 1) + (double)g.P * Cin) * esz(dtype));
}

// general dense convolution (tss_convkxk_fwd): dw in the torch layout [N][Cin][kh][kw]
int tss_convkxk_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil,
                           int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin > 0 && N > 0 && (Cin % 8) == 0 && (N % 8) == 0 && (lde % 8) == 0 && lde >= N && (ldx % 8) == 0 && ldx >= Cin &&
              stride >= 1 && dil >= 1 && kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && kh * kw <= 81, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  WgradArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = dil; g.Cin = Cin; g.gkh = kh; g.gkw = kw;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  g.P = (long)B * g.Hout * g.Wout; g.ND = N; g.KD = Cin; g.ntaps = kh * kw; g.mode = A_TAPS;
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = xraw; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu;
  g.dw = dw; g.drs = (long)Cin * kh * kw; g.dcs = kh * kw; g.dts = 1;
  return launch(g, dtype, TSS_K_CONV3X3_BWD_WEIGHT, (hipStream_t)stream,
                ((double)g.P * N * (yraw ? 2 : 1) + (double)B * Hin * Win * Cin) * esz(dtype));
}

int tss_stem3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* x_nchw, int x_is_f32, float* dw, float* ws,
                           int B, int Cin, int Hin, int Win, int N, int stride, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(Cin >= 1 && Cin * 9 <= TK && N > 0 && (N % 8) == 0 && (lde % 8) == 0 && lde >= N, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= N && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  WgradArgs g = {};
  g.Hin = Hin; g.Win = Win; g.stride = stride; g.dil = 1; g.Cin = Cin;
  g.Hout = (Hin - 1) / stride + 1; g.Wout = (Win - 1) / stride + 1;
  g.P = (long)B * g.Hout * g.Wout; g.ND = N; g.KD = Cin * 9; g.ntaps = 1; g.mode = A_STEM;
  g.e = e; g.lde = lde; g.yraw = yraw; g.ldyr = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.x = x_nchw; g.x_f32 = x_is_f32;
  g.dw = dw; g.drs = (long)Cin * 9; g.dcs = 1; g.dts = 0;
  const double bytes = (double)g.P * N * (yraw ? 2 : 1) * esz(dtype) + (double)B * Cin * Hin * Win * (x_is_f32 ? 4 : esz(dtype));
  if (dtype == TSS_BF16 && !g_tss_disable_fast && Cin <= 3 && N == 32 && ws) {   // performance path: direct kernel (stem.hip)
    tss::ProfScope prof(TSS_K_STEM_BWD_WEIGHT, (hipStream_t)stream, bytes, 2.0 * (double)g.P * N * Cin * 9);
    if (tss_stem_direct_wgrad(e, lde, yraw, ldyr, ga, gb, gce, gmu, x_nchw, x_is_f32, dw, ws, B, Cin, Hin, Win, N, stride,
                              dtype, (hipStream_t)stream))
      return tss::check_last("stem_direct_wgrad");
  }
  return launch(g, dtype, TSS_K_STEM_BWD_WEIGHT, (hipStream_t)stream, bytes);
}

#ifdef TSS_TIMING
int tss_debug_wg_timing(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wg_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_timing), z, sizeof(z)); }
  return 0;
}
#endif

}  // extern "C"

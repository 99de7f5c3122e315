// Backward of a 1x1 (pointwise) layer in ONE sweep: input gradient AND weight gradient from a single pass over (e, y, x).
//
// pwfast_kernel<true> (input gradient) and wgfast_kernel (weight gradient) each read e and y of the layer -- the largest
// operands of its backward -- and each evaluate g = BN'(e, y).  For the layers where that double read is most of the traffic
// (few channels, many pixels: the depthwise-separable blocks at 1/4 and 1/8 resolution, 300-600 MB per layer) this kernel
// reads e, y, x once and writes e_in once.  Per tile of TM pixels a thread loads one 4-pixel x 8-channel unit of e, y and
// of x (16-byte loads, one tile ahead), evaluates g and a = relu?(BN(x)) in registers and stores them THREE ways:
//   Xs  [pixel][channel of g]           -> e_in^T[ci][p]  = sum_kc  W^T[ci][kc] * g[p][kc]      (matrix cores, as pwfast)
//   Gt  [channel of g][pixel], At [channel of a][pixel]  (transposing 8-byte stores, as wgfast)
//                                       -> dW[kc][ci]    += sum_p   g^T[kc][p]  * a^T[ci][p]     (matrix cores, as wgfast)
// The block keeps its whole [NC][KC] weight-gradient tile in accumulators over all its pixel tiles and writes it once, to
// its own workspace row; the rows are summed by tss_dw_reduce_many (plain column sums: the row is already in the
// [Cout][Cin] order of the parameter).  Envelope: bf16, Cout <= 128, Cin <= 128 (one weight tile per block).
#include <cstdlib>

#include "common.h"

namespace {

typedef bf16_t T;

struct PbArgs {
  long P; int NC, KC;                       // NC = Cout (channels of e / y), KC = Cin (channels of x / e_in)
  const T* e; long lde; const T* y; long ldy; const float* ga; const float* gb; const float* gce; const float* gmu;
  const float* w;                           // [NC][KC] f32
  const T* wT; long ldwT;                   // optional bf16 shadow of the transpose, [KC][NC]
  const T* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu, x_pending;
  T* ein; long ldei; double* stats; float* ws; int gslots;
  float* bias_ws;                           // optional [rows][NC]: per-block partial sums of g over the pixels (bias gradient)
  // DROP instances: nn.Dropout sat between the (pending) activation of x and this layer (tss_pwconv_fwd_drop): one mask byte per
  // (pixel, 8-channel vector of x), bit j = kept; the weight gradient sees a * keep / (1 - p), the input gradient is scaled alike
  const unsigned char* dmask; float dinv;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// NT threads (NW waves); TM pixels per tile, TM / NW per wave; FKM = 16-channel fragments of Cin a wave handles (all of them);
// NFW = 16-channel fragments of Cout whose weight-gradient rows a wave owns (fragments w, w + NW, ...).  Two instances:
// NSPLIT = 2: the waves pair up over the input-channel fragments of the input gradient (half of FKM each, twice the pixels).
//   <256, 128, 4, 1, 1>: Cin, Cout <= 64, two blocks per CU;   <256, 64, 4, 2, 1>: Cin <= 64, Cout <= 128, two blocks per CU;
//   <256, 64, 8, 1, 1>: Cin <= 128, Cout <= 64 (<256, 64, 8, 1, 2> for the DROP form: the waves pair up over the input-channel
//   fragments, half the statistics registers; without dropout the unpaired form is the faster one, 68 vs 76 us) (the 19-class classifier conv: Cout need not be a multiple of 8 -- e / y are read
//   through their pitch, channels >= Cout are zeroed here, the weight-gradient row keeps the parameter's [Cout][Cin] shape);
//   <512, 128, 8, 1, 2>: <= 128 channels, one 8-wave block per CU (opt-in: it loses to the two separate kernels)
// TR: no pixel-major image of g at all -- the input-gradient product reads its g operand out of the channel-major image Gt with
// the transposing LDS read of gfx950 (ds_read_b64_tr_b16: a group of 16 lanes reads a 4-row x 16-column block and lane i receives
// column i), which frees the 17 KB that keep the 128-channel shape from fitting twice into a CU's LDS
template <int NT, int TM, int FKM, int NFW, int NSPLIT, bool TR, bool DROP = false>
__global__ __launch_bounds__(NT, (NT == 256 ? 2 : 1)) void pwbwd_kernel(const PbArgs g) {
  constexpr int NW = NT / 64;
  constexpr int NCM = 16 * NW * NFW, KCM = 16 * FKM;     // channel capacities
  constexpr int RSX = NCM + 8;                           // Xs / Ws row pitch (elements): +16 bytes against bank conflicts
  constexpr int ROWT = TM * 2 + 16;                      // Gt / At row pitch (bytes)
  constexpr int NPG = TM / 4, PXW = TM / (NW / NSPLIT), MFX = PXW / 16, NKP = TM / 32, FKW = FKM / NSPLIT;
  static_assert(PXW % 16 == 0 && TM >= 64, "tile shape (the XOR swizzle of the transposed images needs >= 8 chunks of 16 bytes per row)");
  extern __shared__ __align__(16) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);                    // [TM][RSX]  (TR: absent)
  T* Ws = Xs + (TR ? 0 : TM * RSX);                                 // [KCM][RSX]   W^T: row = input channel, columns = output channels
  unsigned char* Gt = reinterpret_cast<unsigned char*>(Ws + KCM * RSX);   // [NCM][ROWT]
  unsigned char* At = Gt + NCM * ROWT;                   // [KCM][ROWT]
  float* Ec = reinterpret_cast<float*>(At + KCM * ROWT); // [3][KCM]: producer's mean / scale / bias (ReLU mask, statistics)
  float* Cg = Ec + 3 * KCM;                              // [3][NCM]: g = ca*e + cb*y + cc
  float* Ca = Cg + 3 * NCM;                              // [2][KCM]: a = relu?(x*as + ab)
  // DROP: the 16-byte mask rows of a tile's pixels, two tiles deep.  Wave 0 requests them one tile ahead (one row per lane, the
  // FIRST request of an iteration, so that every later wait of the wave covers it) and parks them here at the top of the next
  // iteration; read back byte-wise by the staging threads and by the epilogue.  (LDS-DMA was tried first: a pending
  // global_load_lds makes the compiler put s_waitcnt vmcnt(0) in front of EVERY barrier -- the prefetch of the next tile, issued just
  // before the mid-iteration barrier, was then waited for on the spot: 115 us against 68 without dropout.)
  unsigned char* Ms = reinterpret_cast<unsigned char*>(Ca + 2 * KCM);     // [2][TM][16]
  static_assert(!DROP || TM == 64, "one mask row per lane of wave 0");
  uint4 mreg = make_uint4(0u, 0u, 0u, 0u);
  auto mask_req = [&](long tile) {
    const long p = tile * TM + (threadIdx.x & 63);
    mreg = *reinterpret_cast<const uint4*>(g.dmask + (p < g.P ? p : g.P - 1) * 16);
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wp = wave / NSPLIT, i0 = (wave % NSPLIT) * FKW;     // pixel group and first input-channel fragment of this wave (dX)
  const int NC = g.NC, KC = g.KC;
  const int FN = (NC + 15) >> 4, FK = (KC + 15) >> 4;
  const int nks = (NC + 31) >> 5;                        // k-steps of the input-gradient product

  const int xcd = (int)blockIdx.x & 7, gslot = (int)blockIdx.x >> 3;
  const long ntiles = (g.P + TM - 1) / TM;
  const long per = (ntiles + 7) >> 3;
  const long t_begin = xcd * per + gslot;
  long t_end = xcd * per + per;
  if (t_end > ntiles) t_end = ntiles;

  // ---- staging units of this thread: 4 pixels x 8 channels of (e, y) and of x
  const int nvG = (NC + 7) >> 3, nvA = KC >> 3;
  const int pgG = tid / nvG, cvG = tid - pgG * nvG;
  const int pgA = tid / nvA, cvA = tid - pgA * nvA;
  const bool onG = pgG < NPG, onA = pgA < NPG;
  const T* eg = g.e + (onG ? cvG * 8 : 0);
  const T* yg = (g.y ? g.y : g.e) + (onG ? cvG * 8 : 0);
  const long ldyy = g.y ? g.ldy : g.lde;
  const T* xg = g.x + (onA ? cvA * 8 : 0);
  uint4 re[4], ry[4], rx[4];
  uint2 rxn[FKW][MFX];       // raw producer output under this lane's e_in values (ReLU mask + statistics)
  auto issue = [&](long tile) {
    const long p0 = tile * TM;
    if (onG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + pgG * 4 + i;
        const long pc = p < g.P ? p : p0;          // clamp to the tile's first pixel (always valid) and zero afterwards
        re[i] = *reinterpret_cast<const uint4*>(eg + pc * g.lde);
        if (!DROP) ry[i] = *reinterpret_cast<const uint4*>(yg + pc * ldyy);     // (DROP: no BatchNorm behind the layer, host-checked: g = ca * e)
      }
    }
    if (onA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long p = p0 + pgA * 4 + i;
        const long pc = p < g.P ? p : p0;
        rx[i] = *reinterpret_cast<const uint4*>(xg + pc * g.ldx);
      }
    }
  };
  // the raw producer output under this lane's e_in values travels separately: requested behind the epilogue that used the
  // previous set, so that no second copy of it is alive during the matrix products
  auto issue_x = [&](long tile) {
    const long p0 = tile * TM;
#pragma unroll
    for (int i = 0; i < FKW; ++i) {
      if (i0 + i < FK) {
        const int n = (i0 + i) * 16 + fq * 4;
        const T* px = g.x + (n < KC ? n : 0);
#pragma unroll
        for (int m = 0; m < MFX; ++m) {
          const long p = p0 + wp * PXW + m * 16 + fr;
          rxn[i][m] = *reinterpret_cast<const uint2*>(px + (p < g.P ? p : p0) * g.ldx);
        }
      }
    }
  };
  if (t_begin < t_end) { issue(t_begin); if (g.x_pending) issue_x(t_begin); if (DROP && wave == 0) mask_req(t_begin); }

  // ---- block set-up under the first tile's loads: zero the images once (padding rows / columns stay zero), W^T, constants
  {
    constexpr int total = ((TR ? 0 : TM * RSX) + KCM * RSX) * 2 + (NCM + KCM) * ROWT;
    for (int i = tid; i < total / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (tid < KCM) {
    const bool in = tid < KC, has = g.x_pending != 0;
    Ec[tid] = (in && has && g.xm) ? g.xm[tid] : 0.f;
    Ec[KCM + tid] = (in && has && g.xs) ? g.xs[tid] : 1.f;
    Ec[2 * KCM + tid] = (in && has && g.xb) ? g.xb[tid] : 0.f;
  }
  __syncthreads();
  if (g.wT) {
    const int nvr = NC >> 3;                         // 16-byte vectors per row of the transpose
    for (int i = tid; i < KC * nvr; i += NT) {
      const int ci = i / nvr, v = i - ci * nvr;
      *reinterpret_cast<uint4*>(Ws + ci * RSX + v * 8) = *reinterpret_cast<const uint4*>(g.wT + (long)ci * g.ldwT + v * 8);
    }
  } else {
    // (DROP: e_in = keep / (1 - p) * (g W).  Round 3 folded the factor into these weights; rounding w / (1 - p) to bf16 perturbs every
    // weight by up to 2^-9 -- the SAME perturbation for every pixel, so it does not average out in the producer's d(gamma) / d(beta)
    // sums: 1.6e-3 against a noise level of 2e-4 in the oracle test of round 4.  The factor is applied to the f32 accumulator instead.)
    for (int i = tid; i < NC * KC; i += NT) {        // coalesced reads of w[kc][ci], transposing 2-byte stores
      const int kc = i / KC, ci = i - kc * KC;
      Ws[ci * RSX + kc] = (T)g.w[i];
    }
  }
  // per-channel constants of the two transforms -> LDS (read back as 16-byte vectors by the staging threads: 40 registers less)
  {
    const bool hy = g.y != nullptr, hs = g.xs != nullptr;
    if (tid < NCM) {
      const bool in = tid < NC;
      const float a = (in && g.ga) ? g.ga[tid] : 1.f, b = (in && hy) ? g.gb[tid] : 0.f;
      const float ce = (in && hy) ? g.gce[tid] : 0.f, mu = (in && hy) ? g.gmu[tid] : 0.f;
      Cg[tid] = a; Cg[NCM + tid] = b; Cg[2 * NCM + tid] = hy ? -(a * ce) - b * mu : 0.f;
    }
    if (tid < KCM) {
      const bool in = tid < KC;
      const float sc = (in && hs) ? g.xs[tid] : 1.f;
      const float bb = (in && hs && g.xb) ? g.xb[tid] : 0.f, mm = (in && hs && g.xm) ? g.xm[tid] : 0.f;
      Ca[tid] = sc; Ca[KCM + tid] = __builtin_fmaf(-mm, sc, bb);
    }
  }
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;

  f32x4 dw[NFW][FKM];        // this wave's rows of the weight gradient: output-channel fragments wave, wave + 4, ...
#pragma unroll
  for (int u = 0; u < NFW; ++u)
#pragma unroll
    for (int j = 0; j < FKM; ++j) dw[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // bias gradient: this thread's 8 channels of g over its pixels
  float st1[FKW][4], st2[FKW][4];
#pragma unroll
  for (int i = 0; i < FKW; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  auto unit_ptr = [&](unsigned char* tile, int row, int pg) -> unsigned char* {
    const int boff = pg * 8;
    return tile + row * ROWT + ((((boff >> 4)) ^ ((row >> 3) & 7)) << 4) + (boff & 15);
  };

  int mbuf = 0;
  for (long tile = t_begin; tile < t_end; tile += g.gslots) {
    const long p0 = tile * TM;
    if (DROP && wave == 0) *reinterpret_cast<uint4*>(Ms + mbuf * TM * 16 + (threadIdx.x & 63) * 16) = mreg;   // (last read two tiles ago)
    __syncthreads();          // the previous tile's matrix products have read the images (first pass: set-up complete)
    const unsigned char* mrow = Ms + mbuf * TM * 16;
    if (DROP && wave == 0 && tile + g.gslots < t_end) mask_req(tile + g.gslots);
    mbuf ^= 1;
    if (onG) {
      float v[4][8], ca[8], cb[8], cc[8];
      V8<float>::load(Cg + cvG * 8, ca); V8<float>::load(Cg + NCM + cvG * 8, cb); V8<float>::load(Cg + 2 * NCM + cvG * 8, cc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = p0 + pgG * 4 + i < g.P;
        const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re[i]);
        const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry[DROP ? 0 : i]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (DROP) {
            v[i][2 * h] = ca[2 * h] * blo(ue[h]);
            v[i][2 * h + 1] = ca[2 * h + 1] * bhi(ue[h]);
          } else {
            v[i][2 * h] = ca[2 * h] * blo(ue[h]) + (cb[2 * h] * blo(uy[h]) + cc[2 * h]);
            v[i][2 * h + 1] = ca[2 * h + 1] * bhi(ue[h]) + (cb[2 * h + 1] * bhi(uy[h]) + cc[2 * h + 1]);
          }
        }
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
        if (cvG * 8 + 8 > NC) {                   // ragged Cout: whatever sits in the pitch padding of e / y must not reach the products
#pragma unroll
          for (int j = 0; j < 8; ++j) if (cvG * 8 + j >= NC) v[i][j] = 0.f;
        }
        if (!TR) V8<T>::store(Xs + (pgG * 4 + i) * RSX + cvG * 8, v[i]);
      }
      if (g.bias_ws) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bf16x4 o; o[0] = (T)v[0][j]; o[1] = (T)v[1][j]; o[2] = (T)v[2][j]; o[3] = (T)v[3][j];
        *reinterpret_cast<bf16x4*>(unit_ptr(Gt, cvG * 8 + j, pgG)) = o;
      }
    }
    if (onA) {
      float v[4][8], as[8], ab[8];
      V8<float>::load(Ca + cvA * 8, as); V8<float>::load(Ca + KCM + cvA * 8, ab);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = p0 + pgA * 4 + i < g.P;
        const uint32_t* ux = reinterpret_cast<const uint32_t*>(&rx[i]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[i][2 * h] = fmaxf(blo(ux[h]) * as[2 * h] + ab[2 * h], relu_lo);
          v[i][2 * h + 1] = fmaxf(bhi(ux[h]) * as[2 * h + 1] + ab[2 * h + 1], relu_lo);
        }
        if (DROP) {     // kept or zero; the factor 1 / (1 - p) of the weight gradient is applied once, to the block's finished tile
          const uint32_t mk = mrow[(pgA * 4 + i) * 16 + cvA];
#pragma unroll
          for (int j = 0; j < 8; ++j) if (!((mk >> j) & 1u)) v[i][j] = 0.f;
        }
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bf16x4 o; o[0] = (T)v[0][j]; o[1] = (T)v[1][j]; o[2] = (T)v[2][j]; o[3] = (T)v[3][j];
        *reinterpret_cast<bf16x4*>(unit_ptr(At, cvA * 8 + j, pgA)) = o;
      }
    }
    if (tile + g.gslots < t_end) issue(tile + g.gslots);
    __syncthreads();

    // ---- input gradient: D[ci][p] = sum_kc W^T[ci][kc] * g[p][kc]; this wave's pixels: wave * TM/4 ...
    f32x4 acc[MFX][FKW];
#pragma unroll
    for (int m = 0; m < MFX; ++m)
#pragma unroll
      for (int i = 0; i < FKW; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const T* xrow = Xs + (wp * PXW + fr) * RSX + fq * 8;
      const T* wrow = Ws + fr * RSX + fq * 8;
      for (int ks = 0; ks < nks; ++ks) {
        bf16x8 xf[MFX];
#pragma unroll
        for (int m = 0; m < MFX; ++m) {
          if (!TR) {
            xf[m] = *reinterpret_cast<const bf16x8*>(xrow + m * 16 * RSX + ks * 32);
          } else {
            // lane (fr, fq) needs g[pixel px0 + fr][channels kc0 .. kc0 + 7], kc0 = ks*32 + fq*8: two transposed reads of the blocks
            // (rows kc0 + 0..3 | kc0 + 4..7) x (16 pixels from px0) of Gt; lane 4q + p of the group addresses row q, pixels 4p .. 4p+3
            const int px0 = wp * PXW + m * 16, q = fr >> 2, pq = fr & 3;
            typedef __attribute__((ext_vector_type(4))) short v4s;
            v4s lo, hi;
            {
              const int row = ks * 32 + fq * 8 + q, pg = (px0 >> 2) + pq;
              lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)unit_ptr(Gt, row, pg));
              hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)unit_ptr(Gt, row + 4, pg));
            }
            union { v4s h[2]; bf16x8 v; } u;
            u.h[0] = lo; u.h[1] = hi;
            xf[m] = u.v;
          }
        }
#pragma unroll
        for (int i = 0; i < FKW; ++i) {
          if (i0 + i < FK) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + (i0 + i) * 16 * RSX + ks * 32);
#pragma unroll
            for (int m = 0; m < MFX; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[m], acc[m][i], 0, 0, 0);
          }
        }
      }
    }
    // ---- weight gradient: D[kc][ci] += sum_p g^T[kc][p] * a^T[ci][p]
#pragma unroll
    for (int u = 0; u < NFW; ++u) {
      const int f = wave + NW * u;
      if (f < FN) {
        const int rg = f * 16 + fr;
        const unsigned char* grow = Gt + rg * ROWT;
        for (int ks = 0; ks < NKP; ++ks) {
          const bf16x8 gf = *reinterpret_cast<const bf16x8*>(grow + (((ks * 4 + fq) ^ ((rg >> 3) & 7)) << 4));
#pragma unroll
          for (int j = 0; j < FKM; ++j) {
            if (j < FK) {
              const int rk = j * 16 + fr;
              const bf16x8 af = *reinterpret_cast<const bf16x8*>(At + rk * ROWT + (((ks * 4 + fq) ^ ((rk >> 3) & 7)) << 4));
              dw[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, af, dw[u][j], 0, 0, 0);
            }
          }
        }
      }
    }
    // ---- epilogue of the input gradient: ReLU mask of the producer, statistics, store (lane: pixel fr, 4 channels)
#pragma unroll
    for (int i = 0; i < FKW; ++i) {
      if (i0 + i < FK) {
        const int n = (i0 + i) * 16 + fq * 4;
        if (n < KC) {
          const float4 e0 = *reinterpret_cast<const float4*>(Ec + n);
          const float4 e1 = *reinterpret_cast<const float4*>(Ec + KCM + n);
          const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * KCM + n);
          const float cmm[4] = {e0.x, e0.y, e0.z, e0.w}, cms[4] = {e1.x, e1.y, e1.z, e1.w}, cmb[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
          for (int m = 0; m < MFX; ++m) {
            const long p = p0 + wp * PXW + m * 16 + fr;
            if (p < g.P) {
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = acc[m][i][q];
              if (DROP) {
                const uint32_t mb = (uint32_t)mrow[(wp * PXW + m * 16 + fr) * 16 + (n >> 3)] >> (n & 7);   // bit n % 8 onwards of byte n / 8
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ((mb >> q) & 1u) ? v[q] * g.dinv : 0.f;
              }
              bf16x4 o;
              if (g.x_pending) {
                const uint2 xr = rxn[i][m];
                const float xc[4] = {blo(xr.x) - cmm[0], bhi(xr.x) - cmm[1], blo(xr.y) - cmm[2], bhi(xr.y) - cmm[3]};
                if (g.x_relu) {
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
              }
              *reinterpret_cast<bf16x4*>(g.ein + p * g.ldei + n) = o;
            }
          }
        }
      }
    }
    if (g.x_pending && tile + g.gslots < t_end) issue_x(tile + g.gslots);
  }

  // ---- the block's weight-gradient tile -> its workspace row, [NC][KC] like the parameter (blocks without tiles write zeros)
  const int row = xcd + 8 * gslot;
  {
    float* wr = g.ws + (long)row * NC * KC;
#pragma unroll
    for (int u = 0; u < NFW; ++u) {
      const int f = wave + NW * u;
      if (f < FN) {
#pragma unroll
        for (int j = 0; j < FKM; ++j) {
          if (j < FK) {
            const int ci = j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kc = f * 16 + fq * 4 + r;
              if (kc < NC && ci < KC) wr[(long)kc * KC + ci] = DROP ? dw[u][j][r] * g.dinv : dw[u][j][r];
            }
          }
        }
      }
    }
  }
  // ---- bias-gradient row of this block: the pixel groups of every channel vector meet in LDS, summed in a fixed order
  if (g.bias_ws) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [NPG][NCM]
    if (onG) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[pgG * NCM + cvG * 8 + j] = bsum[j];
    }
    __syncthreads();
    if (tid < NC) {
      float t = 0.f;
      for (int q = 0; q < NPG; ++q) t += red[q * NCM + tid];
      g.bias_ws[(long)row * NC + tid] = t;
    }
  }
  // ---- statistics slab row of this block: sum over the 16 pixel lanes of a fragment row, then over the four waves
  if (g.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [NW waves][2][KCM]
    for (int i = tid; i < NW * 2 * KCM; i += NT) red[i] = 0.f;     // (a wave only covers its own channel fragments)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FKW; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u1 = row16_sum(st1[i][q]), u2 = row16_sum(st2[i][q]);
        if (fr == 0) {
          red[(wave * 2 + 0) * KCM + (i0 + i) * 16 + fq * 4 + q] = u1;
          red[(wave * 2 + 1) * KCM + (i0 + i) * 16 + fq * 4 + q] = u2;
        }
      }
    __syncthreads();
    if (tid < KC) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += (double)red[(w * 2 + 0) * KCM + tid]; b += (double)red[(w * 2 + 1) * KCM + tid]; }
      const int rows_used = 8 * g.gslots;
      g.stats[(long)row * 2 * KC + tid] = a;
      g.stats[(long)row * 2 * KC + KC + tid] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * KC + tid] = 0.0;
        g.stats[(long)rr * 2 * KC + KC + tid] = 0.0;
      }
    }
  }
}

template <int NT, int TM, int FKM, int NFW, bool TR, bool DROP = false>
constexpr size_t smem_bytes() {
  constexpr int NCM = 16 * (NT / 64) * NFW, KCM = 16 * FKM;
  return (size_t)((TR ? 0 : TM) + KCM) * (NCM + 8) * 2 + (size_t)(NCM + KCM) * (TM * 2 + 16) + (5 * KCM + 3 * NCM) * sizeof(float)
         + (DROP ? 2 * TM * 16 : 0);
}

template <int NT, int TM, int FKM, int NFW, int NSPLIT, bool TR, bool DROP = false>
int launch(PbArgs& g, hipStream_t stream, int blocks_per_cu) {
  const long ntiles = (g.P + TM - 1) / TM;
  long gs = (ntiles + 7) / 8;
  const long cap = 32 * blocks_per_cu;            // 8 * gs blocks: at most 256 CUs x blocks per CU, and <= 512 slab rows
  if (gs > cap) gs = cap;
  if (gs < 1) gs = 1;
  g.gslots = (int)gs;
  constexpr size_t smem = smem_bytes<NT, TM, FKM, NFW, TR, DROP>();
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwbwd_kernel<NT, TM, FKM, NFW, NSPLIT, TR, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((pwbwd_kernel<NT, TM, FKM, NFW, NSPLIT, TR, DROP>), dim3(8 * (int)gs), dim3(NT), smem, stream, g);
  return 8 * (int)gs;
}

inline bool small_shape(int Cin, int Cout) { return Cin <= 64 && Cout <= 64; }
inline bool mid_shape(int Cin, int Cout) { return Cin <= 64 && Cout <= 128; }     // 64-pixel tiles, two blocks per CU
inline bool tall_shape(int Cin, int Cout) { return Cin <= 128 && Cout <= 64; }    // 64-pixel tiles, two blocks per CU (classifier conv)
// the two 128-channel instances (neither is used by default: both lose to the two separate kernels -- 139 us for the 8-wave one,
// 209 us for the transposed-read one, which spills 69 registers, against 121 us): TSS_PW_BWD_BIG=2 selects the transposed-read one
inline bool big_tr() { const char* s = getenv("TSS_PW_BWD_BIG"); return s && atoi(s) == 2; }

}  // namespace

extern "C" {

// 1 when tss_pwconv_bwd_fused covers the layer AND is the faster choice (few channels, many pixels: the double read of e, y
// by the two separate kernels dominates); the number of workspace rows it writes comes from tss_pwconv_bwd_fused_rows
int tss_pwconv_bwd_fused_preferred(long P, int Cin, int Cout, int dtype) {
  // TSS_PW_BWD_FUSED: 0 = never; n > 1 = every layer inside the envelope with at least n pixels (tests, A/B runs).
  // Default: the two-blocks-per-CU instances -- both channel counts <= 64 from 50 k pixels: 99 vs 157 us (32 -> 48 channels at
  // 1 M pixels), 43 vs 62 (48 -> 64, 262 k), 23.5 vs 27.9 (64 -> 64, 65 k); Cin <= 64, Cout <= 128 from 200 k pixels: 73 vs 96 us
  // (64 -> 128, 262 k); the 128-channel instance (one 8-wave block per
  // CU, 40 spilled registers) loses (139 vs 119 us for 128 -> 128 at 262 k pixels), so those layers keep the two kernels.
  const char* sw = getenv("TSS_PW_BWD_FUSED");
  if (sw && atoi(sw) == 0) return 0;
  const bool ragged_ok = (Cout % 8) == 0 || (!small_shape(Cin, Cout) && !mid_shape(Cin, Cout) && tall_shape(Cin, Cout));
  const bool inside = dtype == TSS_BF16 && Cin >= 8 && Cout >= 8 && (Cin % 8) == 0 && ragged_ok && Cin <= 128 && Cout <= 128;
  if (sw && atol(sw) > 1) return inside && P >= atol(sw);
  // the classifier conv (128 -> 19 classes at 262 k pixels): 84 us through the general kernels (f32 atomics) -> one sweep
  return inside && ((small_shape(Cin, Cout) && P >= 50000) || (mid_shape(Cin, Cout) && P >= 200000)
                    || (!mid_shape(Cin, Cout) && tall_shape(Cin, Cout) && Cout <= 32));      // at every size: the only path without atomics for a ragged Cout
}

int tss_pwconv_bwd_fused_rows(long P, int Cin, int Cout) {
  const bool small = small_shape(Cin, Cout), mid = !small && mid_shape(Cin, Cout), tall = !small && !mid && tall_shape(Cin, Cout);
  const bool big2 = !small && !mid && !tall && big_tr();
  const long TM = (mid || tall || big2) ? 64 : 128;
  long gs = ((P + TM - 1) / TM + 7) / 8;
  const long cap = (small || mid || tall || big2) ? 64 : 32;
  if (gs > cap) gs = cap;
  if (gs < 1) gs = 1;
  return (int)(8 * gs);
}

// e_in = relu'(BN(x)) * (g W) with g = ga*(e-gce) + gb*(y-gmu), and ws[row][Cout][Cin] = per-block partial sums of
// dW = g^T a, a = relu?(BN(x)) (rows: tss_pwconv_bwd_fused_rows; the caller adds them to dW with tss_dw_reduce_many).
// x_pending = 1: x is the producer's raw output (in_* pending): e_in is masked and bstats written, as tss_pwconv_bwd_data does.
int tss_pwconv_bwd_fused(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                         const float* gmu, const float* w, const void* wT_bf16, const void* x, long ldx, const float* in_mean,
                         const float* in_scale, const float* in_bias, int in_relu, int x_pending, void* e_in, long ldei,
                         double* bstats, float* ws, float* bias_ws, long P, int Cin, int Cout, int dtype, void* stream) {
  const bool ragged = (Cout % 8) != 0;         // only the tall instance takes a ragged Cout (vector loads through the pitch)
  const int CoutV = (Cout + 7) / 8 * 8;
  TSS_REQUIRE(dtype == TSS_BF16 && Cin >= 8 && Cout >= 8 && (Cin % 8) == 0 && Cin <= 128 && Cout <= 128, TSS_ERR_SHAPE);
  TSS_REQUIRE(!ragged || (!small_shape(Cin, Cout) && !mid_shape(Cin, Cout) && tall_shape(Cin, Cout)), TSS_ERR_SHAPE);
  TSS_REQUIRE(P > 0 && e && x && e_in && ws && w, TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= CoutV && (ldx % 8) == 0 && ldx >= Cin && (ldei % 4) == 0 && ldei >= Cin, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= CoutV && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || x_pending, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(x) && (!yraw || tss::aligned16(yraw)) && ((uintptr_t)e_in & 7u) == 0, TSS_ERR_ALIGN);
  PbArgs g = {};
  g.P = P; g.NC = Cout; g.KC = Cin;
  g.e = (const T*)e; g.lde = lde; g.y = (const T*)yraw; g.ldy = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.w = w;
  if (wT_bf16 && !ragged && tss::aligned16(wT_bf16)) { g.wT = (const T*)wT_bf16; g.ldwT = Cout; }
  g.x = (const T*)x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.x_pending = x_pending;
  g.ein = (T*)e_in; g.ldei = ldei; g.stats = bstats; g.ws = ws; g.bias_ws = bias_ws;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream,
                      ((double)P * Cout * (yraw ? 2 : 1) + (double)P * Cin * 2) * 2.0, 4.0 * (double)P * Cin * Cout);
  if (small_shape(Cin, Cout)) launch<256, 128, 4, 1, 1, false>(g, (hipStream_t)stream, 2);
  else if (mid_shape(Cin, Cout)) launch<256, 64, 4, 2, 1, false>(g, (hipStream_t)stream, 2);
  else if (tall_shape(Cin, Cout)) launch<256, 64, 8, 1, 1, false>(g, (hipStream_t)stream, 2);
  else if (big_tr()) launch<256, 64, 8, 2, 2, true>(g, (hipStream_t)stream, 2);
  else launch<512, 128, 8, 1, 2, false>(g, (hipStream_t)stream, 1);
  return tss::check_last("pwconv_bwd_fused");
}

// The same sweep for a layer that applied nn.Dropout to its input on load (tss_pwconv_fwd_drop; mask bytes from tss_dropout_mask):
// the weight gradient is taken against a * keep / (1 - p), e_in = keep / (1 - p) * relu'(BN(x)) * (g W).  One instance (64-pixel
// tiles, Cin <= 128, Cout <= 64, ragged Cout allowed): the classifier convs (TSS/models/fastscnn.py:96-97, contextnet.py:85-86).
int tss_pwconv_bwd_fused_drop_supported(long P, int Cin, int Cout, int dtype) {
  return (dtype == TSS_BF16 && P > 0 && Cin >= 8 && (Cin % 8) == 0 && Cin <= 128 && Cout >= 8 && Cout <= 64) ? 1 : 0;
}

int tss_pwconv_bwd_fused_drop_rows(long P) {
  long gs = ((P + 63) / 64 + 7) / 8;
  if (gs > 64) gs = 64;
  if (gs < 1) gs = 1;
  return (int)(8 * gs);
}

int tss_pwconv_bwd_fused_drop(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                              const float* gmu, const float* w, const void* x, long ldx, const float* in_mean,
                              const float* in_scale, const float* in_bias, int in_relu, int x_pending, const void* mask, float drop_p,
                              void* e_in, long ldei, double* bstats, float* ws, float* bias_ws, long P, int Cin, int Cout, int dtype,
                              void* stream) {
  const int CoutV = (Cout + 7) / 8 * 8;
  TSS_REQUIRE(tss_pwconv_bwd_fused_drop_supported(P, Cin, Cout, dtype), TSS_ERR_SHAPE);
  TSS_REQUIRE(e && x && e_in && ws && w && mask && drop_p > 0.f && drop_p < 1.f, TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= CoutV && (ldx % 8) == 0 && ldx >= Cin && (ldei % 4) == 0 && ldei >= Cin, TSS_ERR_SHAPE);
  TSS_REQUIRE(!yraw, TSS_ERR_SHAPE);           // a convolution behind nn.Dropout with a training-mode BatchNorm behind IT: not this entry
  TSS_REQUIRE(!bstats || x_pending, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(x) && ((uintptr_t)e_in & 7u) == 0, TSS_ERR_ALIGN);
  PbArgs g = {};
  g.P = P; g.NC = Cout; g.KC = Cin;
  g.e = (const T*)e; g.lde = lde; g.y = nullptr; g.ldy = 0; g.ga = ga; g.gb = nullptr; g.gce = nullptr; g.gmu = nullptr;
  g.w = w;
  g.x = (const T*)x; g.ldx = ldx; g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; g.x_relu = in_relu; g.x_pending = x_pending;
  g.ein = (T*)e_in; g.ldei = ldei; g.stats = bstats; g.ws = ws; g.bias_ws = bias_ws;
  g.dmask = (const unsigned char*)mask; g.dinv = 1.f / (1.f - drop_p);
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream,
                      ((double)P * Cout * (yraw ? 2 : 1) + (double)P * Cin * 2) * 2.0 + (double)P * (Cin / 8), 4.0 * (double)P * Cin * Cout);
  launch<256, 64, 8, 1, 2, false, true>(g, (hipStream_t)stream, 2);
  return tss::check_last("pwconv_bwd_fused_drop");
}

}  // extern "C"

// The arms of a pyramid pooling module between the pools and the concat (TSS/models/fastscnn.py:106-112,
// Conv2dBlock(in_channels, in_channels / 4, 1) on the 1 x 1, 2 x 2, 3 x 3, 6 x 6 pooled maps): per arm a 1x1 convolution on
// B * bins^2 <= 512 pixels, its training-mode BatchNorm statistics, their finalize and the running-statistics update.
// Through the general operators that is 8 launches forward (conv + finalize per arm) and 12 backward (finalize + weight gradient +
// input gradient per arm) of 5-10 us each for 1.2 MFLOP -- every dependent launch costs >= 4.7 us in the replayed step.  Here
// ONE block per arm does all of it (an arm's pixels all sit in one block, so the batch statistics need no second kernel):
//   ppm_arms_fwd : raw = x W^T (matrix cores), sums of raw and raw^2 -> (mean, invstd, gamma * invstd), running statistics
//   ppm_arms_bwd : sums of e and e (raw - mean) -> d(gamma), d(beta), g = BN'(e, raw); dW += g^T a; e_in = g W
// bf16 activations, f32 parameters; C (input channels) <= 128 and a multiple of 32, Ca (arm channels) 16 or 32.
#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 256, MAXA = 4;

struct ArmsArgs {
  const T* x[MAXA]; long ldx[MAXA];          // pooled maps [P][C]
  const float* w[MAXA];                      // [Ca][C]
  const float* gamma[MAXA];
  float* rmean[MAXA]; float* rvar[MAXA]; long long* nbt[MAXA];
  T* y[MAXA]; long ldy[MAXA];                // raw conv outputs [P][Ca]
  float* vec[MAXA];                          // BatchNorm link vectors [6][Ca]: mean, invstd, scale, ga, gb, gce
  int P[MAXA];
  int C, Ca, training;
  float eps, momentum;
  // backward
  const T* e[MAXA]; long lde[MAXA];
  float* dw[MAXA]; float* dgamma[MAXA]; float* dbeta[MAXA]; int accumulate;
  T* ein[MAXA]; long ldei[MAXA];
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ---------------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(NT) void ppm_arms_fwd_kernel(const ArmsArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int arm = blockIdx.x, C = g.C, Ca = g.Ca, RS = C + 8, FA = Ca >> 4;
  T* Ws = reinterpret_cast<T*>(smem);                 // [Ca][RS]
  T* Xs = Ws + Ca * RS;                               // [64][RS]
  float* red = reinterpret_cast<float*>(Xs + 64 * RS);   // [4 waves][2][Ca]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int P = g.P[arm];
  const T* x = g.x[arm];
  const long ldx = g.ldx[arm];
  const int nv = C >> 3;
  // every global load of a phase is requested before the first one is used: a block is a handful of memory round trips long
  uint4 pre[4];                              // this thread's vectors of the next 64-pixel tile (64 * C / 8 <= 1024 vectors)
  auto issue = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * NT;
      const int r = i / nv, v = i - r * nv;
      const int p = t0 + r;
      pre[u] = *reinterpret_cast<const uint4*>(x + (long)((i < 64 * nv && p < P) ? p : 0) * ldx + (i < 64 * nv ? v : 0) * 8);
    }
  };
  issue(0);
  {
    const float* w = g.w[arm];
    const int nq = (Ca * C) >> 2;            // float4 units of the [Ca][C] weight (<= 1024)
    float4 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = tid + u * NT; wv[u] = *reinterpret_cast<const float4*>(w + (long)(i < nq ? i : 0) * 4); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * NT;
      if (i < nq) {
        const int co = (i * 4) / C, c = i * 4 - co * C;
        bf16x4 o; o[0] = (T)wv[u].x; o[1] = (T)wv[u].y; o[2] = (T)wv[u].z; o[3] = (T)wv[u].w;
        *reinterpret_cast<bf16x4*>(Ws + co * RS + c) = o;
      }
    }
  }
  float st1[2][4], st2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }
  for (int t0 = 0; t0 < P; t0 += 64) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * NT;
      if (i < 64 * nv) {
        const int r = i / nv, v = i - r * nv;
        *reinterpret_cast<uint4*>(Xs + r * RS + v * 8) = (t0 + r < P) ? pre[u] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if (t0 + 64 < P) issue(t0 + 64);
    __syncthreads();
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const T* xrow = Xs + (wave * 16 + fr) * RS + fq * 8;
    const T* wrow = Ws + fr * RS + fq * 8;
    for (int ks = 0; ks < (C >> 5); ++ks) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + ks * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < FA) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + i * 16 * RS + ks * 32);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[i], 0, 0, 0);   // D[channel fq*4 + q][pixel fr]
        }
      }
    }
    const int p = t0 + wave * 16 + fr;
    if (p < P) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < FA) {
          bf16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (T)acc[i][q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[i][q] += rq; st2[i][q] += rq * rq; }
          *reinterpret_cast<bf16x4*>(g.y[arm] + (long)p * g.ldy[arm] + i * 16 + fq * 4) = o;
        }
      }
    }
  }
  if (!g.training) return;                 // eval mode: the affine comes from the running statistics (caller)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float u = row16_sum(st1[i][q]), w2 = row16_sum(st2[i][q]);
      if (fr == 0 && i < FA) {
        red[(wave * 2 + 0) * Ca + i * 16 + fq * 4 + q] = u;
        red[(wave * 2 + 1) * Ca + i * 16 + fq * 4 + q] = w2;
      }
    }
  __syncthreads();
  if (tid == 0 && g.nbt[arm]) *g.nbt[arm] += 1;
  if (tid < Ca) {
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { s += (double)red[(w * 2 + 0) * Ca + tid]; ss += (double)red[(w * 2 + 1) * Ca + tid]; }
    const double count = (double)P;
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)g.eps));
    const float gm = g.gamma[arm] ? g.gamma[arm][tid] : 1.f;
    const float m = (float)mean;
    float* vec = g.vec[arm];
    vec[tid] = m;
    vec[Ca + tid] = invstd;
    vec[2 * Ca + tid] = gm * invstd;
    if (g.rmean[arm]) g.rmean[arm][tid] = (1.f - g.momentum) * g.rmean[arm][tid] + g.momentum * m;
    if (g.rvar[arm]) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      g.rvar[arm][tid] = (1.f - g.momentum) * g.rvar[arm][tid] + g.momentum * (float)unbiased;
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
// pixel chunks of 128: g pixel-major (Gs) for e_in^T = W^T g^T, g and a channel-major (Gt, Xt) for dW = g^T a
__global__ __launch_bounds__(NT) void ppm_arms_bwd_kernel(const ArmsArgs g) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int CH = 128, ROWT = CH * 2 + 16;      // channel-major image row: 128 pixels + 16 bytes
  const int arm = blockIdx.x, C = g.C, Ca = g.Ca, RG = Ca + 8, FA = Ca >> 4, FC = C >> 4;
  T* Wt = reinterpret_cast<T*>(smem);                 // [C][RG]   W^T: row = input channel, columns = arm channels
  T* Gs = Wt + C * RG;                                // [CH][RG]  g, pixel-major
  unsigned char* Gt = reinterpret_cast<unsigned char*>(Gs + CH * RG);   // [Ca][ROWT]  g, channel-major (swizzled 16-byte chunks)
  unsigned char* Xt = Gt + Ca * ROWT;                 // [C][ROWT]   a, channel-major
  float* cf = reinterpret_cast<float*>(Xt + C * ROWT);  // [4][Ca]: ga, gb, gce, mean
  auto unit_ptr = [&](unsigned char* tile, int row, int pg) -> unsigned char* {
    const int boff = pg * 8;
    return tile + row * ROWT + ((((boff >> 4)) ^ ((row >> 3) & 7)) << 4) + (boff & 15);
  };
  double* dred = reinterpret_cast<double*>(cf + 4 * Ca);   // [8 pixel groups][2][Ca]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int P = g.P[arm];
  const T* e = g.e[arm]; const T* y = g.y[arm]; const T* x = g.x[arm];
  const long lde = g.lde[arm], ldy = g.ldy[arm], ldx = g.ldx[arm];
  const float* vec = g.vec[arm];
  // ---- (1) sums of e and e * (raw - mean).  Thread = one 8-channel vector (the same for all its items, NT % nvg == 0) of pixels
  // tid / nvg, tid / nvg + NT / nvg, ...: all loads requested first, f64 partials, then lanes -> waves -> block in a fixed order
  const int nvg = Ca >> 3, nvx = C >> 3;
  {
    const int v = tid % nvg, pg = tid / nvg, pstep = NT / nvg;
    float mu[8];
    V4<float>::load(vec + v * 8, mu); V4<float>::load(vec + v * 8 + 4, mu + 4);
    constexpr int MAXU = 8;                           // P <= 512, nvg <= 4: at most 512 * 4 / 256 items per thread
    uint4 re[MAXU], ry[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int p = pg + u * pstep;
      const long pc = p < P ? p : 0;
      re[u] = *reinterpret_cast<const uint4*>(e + pc * lde + v * 8);
      ry[u] = *reinterpret_cast<const uint4*>(y + pc * ldy + v * 8);
    }
    // W^T under those loads
    {
      const float* w = g.w[arm];
      const int nq = (Ca * C) >> 2;
      float4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = tid + u * NT; wv[u] = *reinterpret_cast<const float4*>(w + (long)(i < nq ? i : 0) * 4); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * NT;
        if (i < nq) {
          const int cc = (i * 4) / C, c = i * 4 - cc * C;
          Wt[(c + 0) * RG + cc] = (T)wv[u].x; Wt[(c + 1) * RG + cc] = (T)wv[u].y;
          Wt[(c + 2) * RG + cc] = (T)wv[u].z; Wt[(c + 3) * RG + cc] = (T)wv[u].w;
        }
      }
    }
    double se[8], sey[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { se[j] = 0.0; sey[j] = 0.0; }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      if (pg + u * pstep < P) {
        const uint32_t* ue = reinterpret_cast<const uint32_t*>(&re[u]);
        const uint32_t* uy = reinterpret_cast<const uint32_t*>(&ry[u]);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float e0 = blo(ue[h]), e1 = bhi(ue[h]);
          se[2 * h] += (double)e0; sey[2 * h] += (double)e0 * (double)(blo(uy[h]) - mu[2 * h]);
          se[2 * h + 1] += (double)e1; sey[2 * h + 1] += (double)e1 * (double)(bhi(uy[h]) - mu[2 * h + 1]);
        }
      }
    }
    // lanes with equal lane % nvg hold the same channels: butterfly over the other lane bits (nvg is 2 or 4)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      for (int m = 32; m >= nvg; m >>= 1) { se[j] += __shfl_xor(se[j], m, 64); sey[j] += __shfl_xor(sey[j], m, 64); }
    }
    if (lane < nvg) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { dred[(wave * 2 + 0) * Ca + lane * 8 + j] = se[j]; dred[(wave * 2 + 1) * Ca + lane * 8 + j] = sey[j]; }
    }
  }
  __syncthreads();
  if (tid < Ca) {
    double se = 0.0, sey = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { se += dred[(q * 2 + 0) * Ca + tid]; sey += dred[(q * 2 + 1) * Ca + tid]; }
    const double r = (double)vec[Ca + tid];
    const double dg = r * sey, db = se;
    if (g.dgamma[arm]) g.dgamma[arm][tid] = (g.accumulate ? g.dgamma[arm][tid] : 0.f) + (float)dg;
    if (g.dbeta[arm]) g.dbeta[arm][tid] = (g.accumulate ? g.dbeta[arm][tid] : 0.f) + (float)db;
    const double k = (g.gamma[arm] ? (double)g.gamma[arm][tid] : 1.0) * r;
    float ga, gb, gce;
    if (g.training) { const double count = (double)P; ga = (float)k; gb = (float)(-k * (dg / count) * r); gce = (float)(db / count); }
    else { ga = (float)k; gb = 0.f; gce = 0.f; }
    cf[tid] = ga; cf[Ca + tid] = gb; cf[2 * Ca + tid] = gce; cf[3 * Ca + tid] = vec[tid];
    float* vo = g.vec[arm];
    vo[3 * Ca + tid] = ga; vo[4 * Ca + tid] = gb; vo[5 * Ca + tid] = gce;
  }
  f32x4 dwa[2][2];       // this wave's tiles of the weight gradient: arm-channel fragments 0..1 x input-channel fragments 2*wave, 2*wave + 1
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) dwa[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int p0 = 0; p0 < P; p0 += CH) {
    __syncthreads();          // coefficients ready (first pass) / the previous chunk's products have read the images
    // ---- (2) g = ga * (e - gce) + gb * (raw - mean), rounded to bf16, both layouts; a = x (the pooled map, materialised), transposed
    {
      // staging units of 4 pixels x 8 channels (as wgrad.hip / pwbwd.hip): per channel ONE 8-byte store of 4 consecutive pixels into
      // the channel-major images, 16-byte chunks XOR-swizzled by the row's vector index (a plain [channel][pixel] transpose puts the
      // 64 lanes of a store on 4 LDS banks).  All global loads of the chunk are requested before the first one is used.
      uint4 ge[4], gy[4], rx[2][4];
      const bool onG = tid < 32 * nvg;
      const int pgG = tid / nvg, cvG = tid - pgG * nvg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = p0 + pgG * 4 + i;
        const long pc = (onG && p < P) ? p : 0;
        ge[i] = *reinterpret_cast<const uint4*>(e + pc * lde + (onG ? cvG : 0) * 8);
        gy[i] = *reinterpret_cast<const uint4*>(y + pc * ldy + (onG ? cvG : 0) * 8);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int id = tid + u * NT;
        const bool on = id < 32 * nvx;
        const int pgA = id / nvx, cvA = id - pgA * nvx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = p0 + pgA * 4 + i;
          rx[u][i] = *reinterpret_cast<const uint4*>(x + (long)((on && p < P) ? p : 0) * ldx + (on ? cvA : 0) * 8);
        }
      }
      if (onG) {
        float gv[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t* ue = reinterpret_cast<const uint32_t*>(&ge[i]);
          const uint32_t* uy = reinterpret_cast<const uint32_t*>(&gy[i]);
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int c0 = cvG * 8 + 2 * h;
            gv[i][2 * h] = cf[c0] * (blo(ue[h]) - cf[2 * Ca + c0]) + cf[Ca + c0] * (blo(uy[h]) - cf[3 * Ca + c0]);
            gv[i][2 * h + 1] = cf[c0 + 1] * (bhi(ue[h]) - cf[2 * Ca + c0 + 1]) + cf[Ca + c0 + 1] * (bhi(uy[h]) - cf[3 * Ca + c0 + 1]);
          }
          if (p0 + pgG * 4 + i >= P) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[i][j] = 0.f;
          }
          V8<T>::store(Gs + (pgG * 4 + i) * RG + cvG * 8, gv[i]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x4 o; o[0] = (T)gv[0][j]; o[1] = (T)gv[1][j]; o[2] = (T)gv[2][j]; o[3] = (T)gv[3][j];
          *reinterpret_cast<bf16x4*>(unit_ptr(Gt, cvG * 8 + j, pgG)) = o;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int id = tid + u * NT;
        if (id < 32 * nvx) {
          const int pgA = id / nvx, cvA = id - pgA * nvx;
          const T* t0 = reinterpret_cast<const T*>(&rx[u][0]); const T* t1 = reinterpret_cast<const T*>(&rx[u][1]);
          const T* t2 = reinterpret_cast<const T*>(&rx[u][2]); const T* t3 = reinterpret_cast<const T*>(&rx[u][3]);
          const T z = (T)0.f;
          const int pb = p0 + pgA * 4;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            bf16x4 o; o[0] = pb < P ? t0[j] : z; o[1] = pb + 1 < P ? t1[j] : z; o[2] = pb + 2 < P ? t2[j] : z; o[3] = pb + 3 < P ? t3[j] : z;
            *reinterpret_cast<bf16x4*>(unit_ptr(Xt, cvA * 8 + j, pgA)) = o;
          }
        }
      }
    }
    __syncthreads();
    // ---- (3) e_in^T[c][p] = sum_co W^T[c][co] g[p][co]: this wave's 32 pixels, all input-channel fragments, one or two k-steps of 16... (Ca <= 32: one step of 32, zero padded)
    {
      f32x4 acc[2][8];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 gf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        gf[m] = *reinterpret_cast<const bf16x8*>(Gs + (wave * 32 + m * 16 + fr) * RG + fq * 8);
        if (fq * 8 >= Ca) gf[m] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};        // Ca == 16: columns 16..31 of the k-step do not exist
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < FC) {
          bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wt + (j * 16 + fr) * RG + fq * 8);
          if (fq * 8 >= Ca) wf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf[m], acc[m][j], 0, 0, 0);   // D[c fq*4+q][pixel fr]
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int p = p0 + wave * 32 + m * 16 + fr;
        if (p < P) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < FC) {
              bf16x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (T)acc[m][j][q];
              *reinterpret_cast<bf16x4*>(g.ein[arm] + (long)p * g.ldei[arm] + j * 16 + fq * 4) = o;
            }
          }
        }
      }
    }
    // ---- (4) dW[co][c] += sum_p g^T[co][p] a^T[c][p]: k-steps of 32 pixels over the chunk
#pragma unroll
    for (int ks = 0; ks < CH / 32; ++ks) {
      bf16x8 gf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rg = i * 16 + fr;
        gf[i] = (i < FA) ? *reinterpret_cast<const bf16x8*>(Gt + rg * ROWT + (((ks * 4 + fq) ^ ((rg >> 3) & 7)) << 4)) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int fc = 2 * wave + j;
        if (fc < FC) {
          const int rk = fc * 16 + fr;
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(Xt + rk * ROWT + (((ks * 4 + fq) ^ ((rk >> 3) & 7)) << 4));
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (i < FA) dwa[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], af, dwa[i][j], 0, 0, 0);   // D[co fq*4+q][c fr]
        }
      }
    }
  }
  float* dw = g.dw[arm];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int fc = 2 * wave + j;
      if (i < FA && fc < FC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long idx = (long)(i * 16 + fq * 4 + q) * C + fc * 16 + fr;
          dw[idx] = (g.accumulate ? dw[idx] : 0.f) + dwa[i][j][q];
        }
      }
    }
}

size_t fwd_smem(int C, int Ca) { return (size_t)(Ca + 64) * (C + 8) * 2 + (size_t)8 * Ca * sizeof(float); }
size_t bwd_smem(int C, int Ca) {
  return (size_t)(C + 128) * (Ca + 8) * 2 + (size_t)(Ca + C) * 136 * 2 + (size_t)4 * Ca * sizeof(float) + (size_t)2 * NT * sizeof(double) + 16;
}

bool arms_ok(int narms, int C, int Ca, const int* P) {
  if (narms < 1 || narms > MAXA || C < 32 || C > 128 || (C % 32) != 0 || (Ca != 16 && Ca != 32)) return false;
  for (int a = 0; a < narms; ++a) if (P[a] < 1 || P[a] > 512) return false;
  return true;
}

}  // namespace

extern "C" {

int tss_ppm_arms_supported(int narms, int C, int Ca, const int* P, int dtype) {
  return dtype == TSS_BF16 && arms_ok(narms, C, Ca, P) ? 1 : 0;
}

int tss_ppm_arms_fwd(const void* const* x, const long* ldx, const float* const* w, const float* const* gamma, float* const* running_mean,
                     float* const* running_var, long long* const* num_batches_tracked, void* const* y, const long* ldy, float* const* vec,
                     const int* P, int narms, int C, int Ca, int training, float eps, float momentum, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(arms_ok(narms, C, Ca, P), TSS_ERR_SHAPE);
  ArmsArgs g = {};
  for (int a = 0; a < narms; ++a) {
    TSS_REQUIRE(x[a] && w[a] && y[a] && (ldx[a] % 8) == 0 && ldx[a] >= C && (ldy[a] % 4) == 0 && ldy[a] >= Ca, TSS_ERR_SHAPE);
    TSS_REQUIRE(!training || vec[a], TSS_ERR_SHAPE);
    TSS_REQUIRE(tss::aligned16(x[a]) && ((uintptr_t)y[a] & 7u) == 0, TSS_ERR_ALIGN);
    g.x[a] = (const T*)x[a]; g.ldx[a] = ldx[a]; g.w[a] = w[a]; g.gamma[a] = gamma ? gamma[a] : nullptr;
    g.rmean[a] = running_mean ? running_mean[a] : nullptr; g.rvar[a] = running_var ? running_var[a] : nullptr;
    g.nbt[a] = num_batches_tracked ? num_batches_tracked[a] : nullptr;
    g.y[a] = (T*)y[a]; g.ldy[a] = ldy[a]; g.vec[a] = vec ? vec[a] : nullptr; g.P[a] = P[a];
  }
  g.C = C; g.Ca = Ca; g.training = training; g.eps = eps; g.momentum = momentum;
  const size_t smem = fwd_smem(C, Ca);
  static tss::DevOnce attr;
  if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ppm_arms_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_smem(128, 32));
  double px = 0; for (int a = 0; a < narms; ++a) px += P[a];
  tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, px * (C + Ca) * 2.0, 2.0 * px * C * Ca);
  hipLaunchKernelGGL(ppm_arms_fwd_kernel, dim3(narms), dim3(NT), smem, (hipStream_t)stream, g);
  return tss::check_last("ppm_arms_fwd");
}

int tss_ppm_arms_bwd(const void* const* e, const long* lde, const void* const* y, const long* ldy, const void* const* x, const long* ldx,
                     const float* const* w, const float* const* gamma, float* const* vec, float* const* dw, float* const* dgamma,
                     float* const* dbeta, int accumulate, void* const* e_in, const long* ldei, const int* P, int narms, int C, int Ca,
                     int training, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(arms_ok(narms, C, Ca, P), TSS_ERR_SHAPE);
  ArmsArgs g = {};
  for (int a = 0; a < narms; ++a) {
    TSS_REQUIRE(e[a] && y[a] && x[a] && w[a] && vec[a] && dw[a] && e_in[a], TSS_ERR_SHAPE);
    TSS_REQUIRE((lde[a] % 8) == 0 && lde[a] >= Ca && (ldy[a] % 8) == 0 && ldy[a] >= Ca && (ldx[a] % 8) == 0 && ldx[a] >= C
                && (ldei[a] % 4) == 0 && ldei[a] >= C, TSS_ERR_SHAPE);
    TSS_REQUIRE(tss::aligned16(e[a]) && tss::aligned16(y[a]) && tss::aligned16(x[a]) && ((uintptr_t)e_in[a] & 7u) == 0, TSS_ERR_ALIGN);
    g.e[a] = (const T*)e[a]; g.lde[a] = lde[a]; g.y[a] = (T*)const_cast<void*>(y[a]); g.ldy[a] = ldy[a];
    g.x[a] = (const T*)x[a]; g.ldx[a] = ldx[a]; g.w[a] = w[a]; g.gamma[a] = gamma ? gamma[a] : nullptr; g.vec[a] = vec[a];
    g.dw[a] = dw[a]; g.dgamma[a] = dgamma ? dgamma[a] : nullptr; g.dbeta[a] = dbeta ? dbeta[a] : nullptr;
    g.ein[a] = (T*)e_in[a]; g.ldei[a] = ldei[a]; g.P[a] = P[a];
  }
  g.C = C; g.Ca = Ca; g.training = training; g.accumulate = accumulate;
  const size_t smem = bwd_smem(C, Ca);
  static tss::DevOnce attr;
  if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ppm_arms_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_smem(128, 32));
  double px = 0; for (int a = 0; a < narms; ++a) px += P[a];
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream, px * (2.0 * Ca + 2.0 * C) * 2.0, 4.0 * px * C * Ca);
  hipLaunchKernelGGL(ppm_arms_bwd_kernel, dim3(narms), dim3(NT), smem, (hipStream_t)stream, g);
  return tss::check_last("ppm_arms_bwd");
}

}  // extern "C"

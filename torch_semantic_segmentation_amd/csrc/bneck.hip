// Eval-mode inverted residual as ONE kernel (round 4): 1x1 expand -> BatchNorm -> ReLU -> depthwise 3x3
// (stride 1 / 2) -> BatchNorm -> ReLU -> 1x1 project -> BatchNorm (+ skip) -> ReLU of TSS/models/fastscnn.py:138-161 and
// TSS/models/contextnet.py:129-147 with FROZEN statistics.  With batch statistics every convolution needs a grid-wide
// reduction before its consumer may start; with running statistics nothing forces a kernel boundary inside the
// block, and the two 6x-expanded tensors -- ~40 % of the traffic of an eval forward -- need never exist in HBM:
//
//   block = one TH x TW tile of OUTPUT pixels of one image.  The (S (TH - 1) + 3) x (S (TW - 1) + 3) input pixels under it (1-pixel
//   halo) are copied to LDS once.  The expanded channels are walked in chunks of 64:
//     1. E = relu(bn1(Xh W1_c^T)) for EVERY halo pixel (matrix cores; pixels outside the image are zeroed: the depthwise layer
//        pads its INPUT with zeros, not with relu(bn1(0)))                                            -> LDS, bf16 [halo pixel][64]
//     2. D = relu(bn2(dw3x3(E)))  for the tile's output pixels (vector unit, 8 channels per lane)     -> LDS, bf16 [pixel][64]
//     3. acc += D W3_c^T           (matrix cores, accumulators stay in registers over the chunks)
//   epilogue: y = relu(bn3(acc) + x)  (the skip is read from the input tile in LDS).
//   The weights of chunk c + 1 (W1 rows, W3 columns: 32 KB) are requested while chunk c computes.
//
// HBM traffic of a block: input tile once (halo overlap: L2), output once.  Cost: the expand product is recomputed on the halo
// (1.4 x for 8 x 16 tiles; the matrix cores idle otherwise), and every block streams the block's whole weight set from L2.
// bf16 activations only; Cmid a multiple of 64, Cout a multiple of 16 <= 128, Cin a multiple of 8 <= 128.
#include <cstdlib>

#include "common.h"

namespace {

typedef bf16_t T;
constexpr int NT = 512, CM = 64, PE = CM * 2 + 16;      // row pitch (bytes) of the E / D / W3 images: + 16 against bank conflicts

struct BkArgs {
  const T* x; long ldx; T* y; long ldy;
  const T* w1b; const float* w1f;     // [Cmid][Cin]: bf16 shadow, or the f32 parameter
  const float* wdw;                   // [Cmid][9]
  const T* w3b; const float* w3f;     // [Cout][Cmid]
  const float* m1; const float* s1; const float* b1;      // frozen BatchNorm behind each layer: (x - mean) * scale + beta
  const float* m2; const float* s2; const float* b2;
  const float* m3; const float* s3; const float* b3;
  int B, H, W, Ho, Wo, Cin, Cmid, Cout, residual, tiles_y, tiles_x;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ uint4 load_w8(const T* wb, const float* wf, long idx, bool ok) {
  // 8 consecutive weights as bf16 (from the shadow, or converted from f32); zeros when !ok
  uint4 r = make_uint4(0u, 0u, 0u, 0u);
  if (wb) {
    r = *reinterpret_cast<const uint4*>(wb + (ok ? idx : 0));
  } else {
    const float4 a = *reinterpret_cast<const float4*>(wf + (ok ? idx : 0));
    const float4 b = *reinterpret_cast<const float4*>(wf + (ok ? idx : 0) + 4);
    bf16x8 o;
    o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w; o[4] = (T)b.x; o[5] = (T)b.y; o[6] = (T)b.z; o[7] = (T)b.w;
    r = *reinterpret_cast<uint4*>(&o);
  }
  if (!ok) r = make_uint4(0u, 0u, 0u, 0u);
  return r;
}

template <int S, int TH, int TW>
struct Geo {
  static constexpr int TP = TH * TW, HH = S * (TH - 1) + 3, HW = S * (TW - 1) + 3, HP = HH * HW, HPp = (HP + 15) & ~15, NPF = HPp / 16;
  static constexpr int NPX = TP / 16, WPP = 8 / NPX, MAXF3 = 8 / WPP;      // step 3: pixel fragments, waves per pixel fragment
  static_assert(TP % 64 == 0 && (NPX == 8 || NPX == 4), "tile");
  // LDS bytes for (Cin, Cout): Xh + W1c | E | D | W3c | constants
  static size_t lds(int Cin, int Cout) {
    const int kwp = (Cin + 31) & ~31, px = kwp * 2 + 16, cop = (Cout + 15) & ~15;
    return (size_t)HPp * px + (size_t)CM * px + (size_t)HPp * PE + (size_t)TP * PE + (size_t)cop * PE
           + (4 * CM + 9 * CM + 2 * 128) * sizeof(float) + HPp * sizeof(float);
  }
};

template <int S, int TH, int TW>
__global__ __launch_bounds__(NT, 2) void bneck_eval_kernel(const BkArgs g) {
  typedef Geo<S, TH, TW> G;
  constexpr int TP = G::TP, HW = G::HW, HP = G::HP, HPp = G::HPp, NPF = G::NPF;
  extern __shared__ __align__(16) unsigned char smem[];
  const int Cin = g.Cin, Cmid = g.Cmid, Cout = g.Cout;
  const int kwp = (Cin + 31) & ~31, nks1 = kwp >> 5, nvec = Cin >> 3, nvecp = kwp >> 3, PX = kwp * 2 + 16;
  const int cop = (Cout + 15) & ~15, NCF = cop >> 4;
  unsigned char* Xh = smem;
  unsigned char* W1c = Xh + HPp * PX;
  unsigned char* E = W1c + CM * PX;
  unsigned char* D = E + HPp * PE;
  unsigned char* W3c = D + TP * PE;
  float* K1 = reinterpret_cast<float*>(W3c + cop * PE);     // [2][CM]: scale, shift of bn1 for this chunk
  float* K2 = K1 + 2 * CM;                                  // [2][CM]: bn2
  float* Wd = K2 + 2 * CM;                                  // [9][CM]: depthwise taps, tap-major
  float* K3 = Wd + 9 * CM;                                  // [2][128]: bn3
  float* Vf = K3 + 2 * 128;                                 // [HPp]: 1 inside the image, 0 outside

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  int bid = blockIdx.x;
  const int tx = bid % g.tiles_x; bid /= g.tiles_x;
  const int ty = bid % g.tiles_y; const int b = bid / g.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // ---- weights of chunk 0 (requested first), input tile, constants
  // W1c: CM rows x nvecp vectors (<= 1024); W3c: cop rows x 8 vectors (<= 1024): two vectors of each per thread
  uint4 pw1[2], pw3[2];
  auto request_weights = [&](int c) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i1 = tid + u * NT;
      const int r1 = i1 / nvecp, v1 = i1 - r1 * nvecp;
      pw1[u] = load_w8(g.w1b, g.w1f, (long)(c * CM + r1) * Cin + v1 * 8, r1 < CM && v1 < nvec);
      const int r3 = i1 >> 3, v3 = i1 & 7;
      pw3[u] = load_w8(g.w3b, g.w3f, (long)r3 * Cmid + c * CM + v3 * 8, r3 < Cout);
    }
  };
  request_weights(0);
  for (int i = tid; i < HPp * nvecp; i += NT) {
    const int hp = i / nvecp, v = i - hp * nvecp;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = hp < HP && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W && v < nvec;
    uint4 val = *reinterpret_cast<const uint4*>(g.x + (ok ? (((long)b * g.H + iy) * g.W + ix) * g.ldx + v * 8 : 0));
    if (!ok) val = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(Xh + hp * PX + v * 16) = val;
  }
  for (int hp = tid; hp < HPp; hp += NT) {
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    Vf[hp] = (hp < HP && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? 1.f : 0.f;
  }
  for (int c = tid; c < 128; c += NT) {
    const bool in = c < Cout;
    const float sc = in ? g.s3[c] : 0.f;
    K3[c] = sc; K3[128 + c] = in ? __builtin_fmaf(-g.m3[c], sc, g.b3[c]) : 0.f;
  }

  // step 3 accumulators: this wave's pixel fragment x its share of the output-channel fragments
  const int pxf = wave % G::NPX, part = wave / G::NPX;
  f32x4 acc3[G::MAXF3];
#pragma unroll
  for (int i = 0; i < G::MAXF3; ++i) acc3[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = Cmid / CM;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();          // the previous chunk's products have read W1c / W3c / D (first pass: nothing to wait for)
    // ---- (a) this chunk's weights and constants -> LDS; request the next chunk's
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i1 = tid + u * NT;
      const int r1 = i1 / nvecp, v1 = i1 - r1 * nvecp;
      if (r1 < CM) *reinterpret_cast<uint4*>(W1c + r1 * PX + v1 * 16) = pw1[u];
      const int r3 = i1 >> 3, v3 = i1 & 7;
      if (r3 < cop) *reinterpret_cast<uint4*>(W3c + r3 * PE + v3 * 16) = pw3[u];
    }
    if (tid < CM) {
      const int ch = c * CM + tid;
      const float a1 = g.s1[ch], a2 = g.s2[ch];
      K1[tid] = a1; K1[CM + tid] = __builtin_fmaf(-g.m1[ch], a1, g.b1[ch]);
      K2[tid] = a2; K2[CM + tid] = __builtin_fmaf(-g.m2[ch], a2, g.b2[ch]);
    }
    for (int i = tid; i < 9 * CM; i += NT) {
      const int tap = i / CM, ch = i - tap * CM;
      Wd[i] = g.wdw[(long)(c * CM + ch) * 9 + tap];
    }
    if (c + 1 < nchunk) request_weights(c + 1);
    __syncthreads();

    // ---- (b) E = relu(bn1(Xh W1c^T)) on every halo pixel: wave = one 16-channel fragment of the chunk x every second pixel fragment
    {
      const int cf = wave & 3, half = wave >> 2;
      const unsigned char* wrow = W1c + (cf * 16 + fr) * PX + fq * 16;
      const float4 k1s = *reinterpret_cast<const float4*>(K1 + cf * 16 + fq * 4);
      const float4 k1b = *reinterpret_cast<const float4*>(K1 + CM + cf * 16 + fq * 4);
      const float s1v[4] = {k1s.x, k1s.y, k1s.z, k1s.w}, b1v[4] = {k1b.x, k1b.y, k1b.z, k1b.w};
#pragma unroll
      for (int j0 = 0; j0 < (NPF + 1) / 2; j0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nks1; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + ks * 64);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pf = half + 2 * (j0 + u);
            if (pf < NPF) {
              const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xh + (pf * 16 + fr) * PX + fq * 16 + ks * 64);
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[u], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pf = half + 2 * (j0 + u);
          if (pf < NPF) {
            const int px = pf * 16 + fr;
            const float vin = Vf[px];
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (T)(fmaxf(__builtin_fmaf(acc[u][q], s1v[q], b1v[q]), 0.f) * vin);
            *reinterpret_cast<bf16x4*>(E + px * PE + (cf * 16 + fq * 4) * 2) = o;
          }
        }
      }
    }
    __syncthreads();

    // ---- (c) D = relu(bn2(dw3x3(E))): lane = 8 channels of one output pixel
    {
      const int v = tid & 7;
      float wt[9][8];
#pragma unroll
      for (int t = 0; t < 9; ++t) V8<float>::load(Wd + t * CM + v * 8, wt[t]);
      float s2v[8], b2v[8];
      V8<float>::load(K2 + v * 8, s2v); V8<float>::load(K2 + CM + v * 8, b2v);
#pragma unroll
      for (int ps = 0; ps < TP / 64; ++ps) {
        const int q = (tid >> 3) + ps * 64;
        const int qy = q / TW, qx = q - qy * TW;
        const unsigned char* e0 = E + ((qy * S) * HW + qx * S) * PE + v * 16;
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const uint4 r = *reinterpret_cast<const uint4*>(e0 + (dy * HW + dx) * PE);
            const float* w = wt[dy * 3 + dx];
            sum[0] = __builtin_fmaf(blo(r.x), w[0], sum[0]); sum[1] = __builtin_fmaf(bhi(r.x), w[1], sum[1]);
            sum[2] = __builtin_fmaf(blo(r.y), w[2], sum[2]); sum[3] = __builtin_fmaf(bhi(r.y), w[3], sum[3]);
            sum[4] = __builtin_fmaf(blo(r.z), w[4], sum[4]); sum[5] = __builtin_fmaf(bhi(r.z), w[5], sum[5]);
            sum[6] = __builtin_fmaf(blo(r.w), w[6], sum[6]); sum[7] = __builtin_fmaf(bhi(r.w), w[7], sum[7]);
          }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)fmaxf(__builtin_fmaf(sum[j], s2v[j], b2v[j]), 0.f);
        *reinterpret_cast<bf16x8*>(D + q * PE + v * 16) = o;
      }
    }
    __syncthreads();

    // ---- (d) acc3 += D W3c^T
    {
      const unsigned char* drow = D + (pxf * 16 + fr) * PE + fq * 16;
#pragma unroll
      for (int ks = 0; ks < CM / 32; ++ks) {
        const bf16x8 df = *reinterpret_cast<const bf16x8*>(drow + ks * 64);
#pragma unroll
        for (int ii = 0; ii < G::MAXF3; ++ii) {
          const int i = part + ii * G::WPP;
          if (i < NCF) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(W3c + (i * 16 + fr) * PE + fq * 16 + ks * 64);
            acc3[ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df, acc3[ii], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: y = relu(bn3(acc) + x); lane = pixel fr of its fragment, 4 consecutive channels
  {
    const int q = pxf * 16 + fr;
    const int qy = q / TW, qx = q - qy * TW;
    const int oy = oy0 + qy, ox = ox0 + qx;
    if (oy < g.Ho && ox < g.Wo) {
      T* yp = g.y + (((long)b * g.Ho + oy) * g.Wo + ox) * g.ldy;
      const unsigned char* xc = Xh + ((qy * S + 1) * HW + qx * S + 1) * PX;     // the pixel under this output (stride 1: the skip)
#pragma unroll
      for (int ii = 0; ii < G::MAXF3; ++ii) {
        const int i = part + ii * G::WPP;
        const int ch = i * 16 + fq * 4;
        if (i < NCF && ch < Cout) {
          const float4 sc = *reinterpret_cast<const float4*>(K3 + ch);
          const float4 sh = *reinterpret_cast<const float4*>(K3 + 128 + ch);
          float v[4] = {__builtin_fmaf(acc3[ii][0], sc.x, sh.x), __builtin_fmaf(acc3[ii][1], sc.y, sh.y),
                        __builtin_fmaf(acc3[ii][2], sc.z, sh.z), __builtin_fmaf(acc3[ii][3], sc.w, sh.w)};
          if (g.residual) {
            const uint2 xr = *reinterpret_cast<const uint2*>(xc + ch * 2);
            v[0] += blo(xr.x); v[1] += bhi(xr.x); v[2] += blo(xr.y); v[3] += bhi(xr.y);
          }
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (T)fmaxf(v[k], 0.f);
          *reinterpret_cast<bf16x4*>(yp + ch) = o;
        }
      }
    }
  }
}

template <int S, int TH, int TW>
int launch(BkArgs& g, hipStream_t stream) {
  typedef Geo<S, TH, TW> G;
  g.tiles_y = (g.Ho + TH - 1) / TH; g.tiles_x = (g.Wo + TW - 1) / TW;
  const size_t smem = G::lds(g.Cin, g.Cout);
  if (smem > 160 * 1024) return TSS_ERR_SHAPE;
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bneck_eval_kernel<S, TH, TW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((bneck_eval_kernel<S, TH, TW>), dim3(g.B * g.tiles_y * g.tiles_x), dim3(NT), smem, stream, g);
  return TSS_OK;
}

}  // namespace

extern int g_tss_disable_fast;

extern "C" {

int tss_bneck_eval_supported(int Cin, int Cmid, int Cout, int stride, int residual, int dtype) {
  static const int sw = getenv("TSS_BNECK_EVAL") ? atoi(getenv("TSS_BNECK_EVAL")) : 1;       // A/B switch
  if (!sw || g_tss_disable_fast || dtype != TSS_BF16) return 0;
  if (stride != 1 && stride != 2) return 0;
  if (Cin < 8 || Cin > 128 || (Cin % 8) != 0 || Cmid < CM || (Cmid % CM) != 0 || Cout < 16 || Cout > 128 || (Cout % 4) != 0) return 0;
  if (residual && (stride != 1 || Cin != Cout)) return 0;
  return 1;
}

// x: bf16 NHWC [B][H][W][Cin] (pitch ldx), materialised.  y: bf16 NHWC [B][Ho][Wo][Cout], Ho = (H - 1) / stride + 1.
// w1 / w3: the f32 parameters [Cmid][Cin] / [Cout][Cmid]; w1_bf16 / w3_bf16: optional current bf16 shadows of them (same layout).
// (mean_i, scale_i, beta_i): the frozen BatchNorm behind layer i as tss_bn_eval_affine writes it.  residual: y = relu(bn3(.) + x).
int tss_bneck_eval_fwd(const void* x, long ldx, const float* w1, const void* w1_bf16, const float* mean1, const float* scale1,
                       const float* beta1, const float* wdw, const float* mean2, const float* scale2, const float* beta2,
                       const float* w3, const void* w3_bf16, const float* mean3, const float* scale3, const float* beta3,
                       int residual, void* y, long ldy, int B, int H, int W, int Cin, int Cmid, int Cout, int stride, int dtype,
                       void* stream) {
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(tss_bneck_eval_supported(Cin, Cmid, Cout, stride, residual, dtype), TSS_ERR_SHAPE);
  TSS_REQUIRE(B > 0 && H > 0 && W > 0 && x && y && w1 && wdw && w3 && mean1 && scale1 && beta1 && mean2 && scale2 && beta2
              && mean3 && scale3 && beta3, TSS_ERR_SHAPE);
  TSS_REQUIRE((ldx % 8) == 0 && ldx >= Cin && (ldy % 4) == 0 && ldy >= Cout, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(x) && ((uintptr_t)y & 7u) == 0 && tss::aligned16(w1) && tss::aligned16(w3), TSS_ERR_ALIGN);
  BkArgs g = {};
  g.x = (const T*)x; g.ldx = ldx; g.y = (T*)y; g.ldy = ldy;
  if (w1_bf16 && tss::aligned16(w1_bf16)) g.w1b = (const T*)w1_bf16; else g.w1f = w1;
  if (w3_bf16 && tss::aligned16(w3_bf16)) g.w3b = (const T*)w3_bf16; else g.w3f = w3;
  g.wdw = wdw;
  g.m1 = mean1; g.s1 = scale1; g.b1 = beta1; g.m2 = mean2; g.s2 = scale2; g.b2 = beta2; g.m3 = mean3; g.s3 = scale3; g.b3 = beta3;
  g.B = B; g.H = H; g.W = W; g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
  g.Cin = Cin; g.Cmid = Cmid; g.Cout = Cout; g.residual = residual;
  const long P = (long)B * g.Ho * g.Wo;
  tss::ProfScope prof(TSS_K_PWCONV_FWD, (hipStream_t)stream, ((double)B * H * W * Cin + (double)P * Cout) * 2.0,
                      2.0 * ((double)B * H * W * Cin * Cmid + (double)P * Cmid * (9 + Cout)));
  int rc;
  // 8 x 16 tiles when they fill the chip, else 8 x 8 (twice the blocks); stride 2: 4 x 16 outputs over a 9 x 33 input tile
  if (stride == 2) rc = launch<2, 4, 16>(g, (hipStream_t)stream);
  else if ((long)B * ((g.Ho + 7) / 8) * ((g.Wo + 15) / 16) >= 256) rc = launch<1, 8, 16>(g, (hipStream_t)stream);
  else rc = launch<1, 8, 8>(g, (hipStream_t)stream);
  if (rc) return rc;
  return tss::check_last("bneck_eval_fwd");
}

}  // extern "C"

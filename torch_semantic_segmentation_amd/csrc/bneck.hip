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
//   Two barriers per chunk: phase 1 = stage chunk c + 1's weights and constants into the other LDS set (requested a phase earlier, in
//   registers meanwhile) + step 2 of chunk c; phase 2 = step 3 of chunk c + step 1 of chunk c + 1 (E of chunk c has been consumed).
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

#ifdef TSS_TIMING
__device__ unsigned long long g_bk_timing[8];     // debug builds: cycles of wave 0 in [set-up, staging, depthwise, project, expand, epilogue, -, blocks]
#define BK_T(var) unsigned long long var; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define BK_T(var)
#endif

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
  static constexpr int NPX = TP / 16, WPP = 8 / NPX, MAXF3 = 8 / WPP;      // project: pixel fragments, waves per pixel fragment
  static constexpr int NG1 = (NPF + 7) / 8;                                 // expand: pixel fragments per wave (a wave = one channel fragment x every second pixel fragment x two halves)
  static_assert(TP % 32 == 0 && (NPX == 8 || NPX == 4 || NPX == 2) && (S == 1 || TP % 64 == 0), "tile");
  static constexpr int KSET = (4 * CM + 9 * CM) * (int)sizeof(float);      // per-chunk constants: bn1 (scale, shift), bn2, 9 taps
  // stride 1: E is kept in f32 -- the depthwise stage is bound by instruction issue, and unpacking bf16 was half of its instructions
  // (9 taps x (8 unpack + 8 fma) per 8 channels); with f32 rows it is 2 reads + 4 packed fmas per tap.  Stride 2 needs four times the
  // halo pixels per output and keeps bf16 (an f32 E tile of 304 pixels would not fit).
  static constexpr bool EF32 = (S == 1);
  static constexpr int PEE = EF32 ? CM * 4 + 16 : PE;                       // row pitch of E (bytes)
  // LDS bytes for (Cin, Cout): Xh | E | D | 2 x (W1c, W3c, chunk constants) | bn3 | validity
  static size_t lds(int Cin, int Cout) {
    const int kwp = (Cin + 31) & ~31, px = kwp * 2 + 16, cop = (Cout + 15) & ~15;
    return (size_t)HPp * px + (size_t)HPp * PEE + (size_t)TP * PE + 2 * ((size_t)CM * px + (size_t)cop * PE + KSET)
           + 2 * 128 * sizeof(float) + HPp * sizeof(float);
  }
};

// what a thread holds of the NEXT chunk's weights and constants while the current chunk computes
struct Pre {
  uint4 w1[2], w3[2];
  float k[6];         // tid < CM: scale / mean / beta of bn1 and bn2 for channel tid of the chunk
  float wd[2];        // depthwise taps: elements tid and tid + NT of the chunk's [9][CM] table
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
  unsigned char* E = Xh + HPp * PX;
  constexpr int PEE = G::PEE;
  unsigned char* D = E + HPp * PEE;
  unsigned char* Sets = D + TP * PE;                          // two sets of (W1c [CM][PX], W3c [cop][PE], constants)
  const int SETB = CM * PX + cop * PE + G::KSET;
  float* K3 = reinterpret_cast<float*>(Sets + 2 * SETB);     // [2][128]: bn3 scale, shift
  float* Vf = K3 + 2 * 128;                                  // [HPp]: 1 inside the image, 0 outside
  auto W1c = [&](int set) { return Sets + set * SETB; };
  auto W3c = [&](int set) { return Sets + set * SETB + CM * PX; };
  auto Kc = [&](int set) { return reinterpret_cast<float*>(Sets + set * SETB + CM * PX + cop * PE); };   // [2][CM] bn1 | [2][CM] bn2 | [9][CM] taps

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  BK_T(t0);
#ifdef TSS_TIMING
  unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull};
#endif
  int bid = blockIdx.x;
  const int tx = bid % g.tiles_x; bid /= g.tiles_x;
  const int ty = bid % g.tiles_y; const int b = bid / g.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int nchunk = Cmid / CM;

  // W1c: CM rows x nvecp vectors (<= 1024); W3c: cop rows x 8 vectors (<= 1024): two vectors of each per thread
  const int r1a = tid / nvecp, v1a = tid - r1a * nvecp, r1b = (tid + NT) / nvecp, v1b = (tid + NT) - r1b * nvecp;
  auto request = [&](int c, Pre& p) {
    p.w1[0] = load_w8(g.w1b, g.w1f, (long)(c * CM + r1a) * Cin + v1a * 8, r1a < CM && v1a < nvec);
    p.w1[1] = load_w8(g.w1b, g.w1f, (long)(c * CM + r1b) * Cin + v1b * 8, r1b < CM && v1b < nvec);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i3 = tid + u * NT, r3 = i3 >> 3, v3 = i3 & 7;
      p.w3[u] = load_w8(g.w3b, g.w3f, (long)r3 * Cmid + c * CM + v3 * 8, r3 < Cout);
    }
    const int ch = c * CM + (tid & (CM - 1));
    p.k[0] = g.s1[ch]; p.k[1] = g.m1[ch]; p.k[2] = g.b1[ch]; p.k[3] = g.s2[ch]; p.k[4] = g.m2[ch]; p.k[5] = g.b2[ch];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * NT;
      const int tap = i / CM, cc = i - tap * CM;
      p.wd[u] = g.wdw[(long)(c * CM + cc) * 9 + (i < 9 * CM ? tap : 0)];
    }
  };
  auto stage = [&](int set, const Pre& p) {
    if (r1a < CM) *reinterpret_cast<uint4*>(W1c(set) + r1a * PX + v1a * 16) = p.w1[0];
    if (r1b < CM) *reinterpret_cast<uint4*>(W1c(set) + r1b * PX + v1b * 16) = p.w1[1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i3 = tid + u * NT, r3 = i3 >> 3, v3 = i3 & 7;
      if (r3 < cop) *reinterpret_cast<uint4*>(W3c(set) + r3 * PE + v3 * 16) = p.w3[u];
    }
    float* K = Kc(set);
    if (tid < CM) {
      K[tid] = p.k[0]; K[CM + tid] = __builtin_fmaf(-p.k[1], p.k[0], p.k[2]);
      K[2 * CM + tid] = p.k[3]; K[3 * CM + tid] = __builtin_fmaf(-p.k[4], p.k[3], p.k[5]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * NT;
      if (i < 9 * CM) K[4 * CM + i] = p.wd[u];
    }
  };

  // E = relu(bn1(Xh W1c^T)) on every halo pixel: wave = one 16-channel fragment of the chunk x every second pixel fragment
  auto expand = [&](int set) {
    const int cf = wave & 3, half = wave >> 2;
    const unsigned char* wrow = W1c(set) + (cf * 16 + fr) * PX + fq * 16;
    const float* K = Kc(set);
    const float4 k1s = *reinterpret_cast<const float4*>(K + cf * 16 + fq * 4);
    const float4 k1b = *reinterpret_cast<const float4*>(K + CM + cf * 16 + fq * 4);
    const float s1v[4] = {k1s.x, k1s.y, k1s.z, k1s.w}, b1v[4] = {k1b.x, k1b.y, k1b.z, k1b.w};
    bf16x8 wf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + (ks < nks1 ? ks : 0) * 64);
    // every product of the wave first, every epilogue (the LDS stores) after: with the stores of one group in front of the reads of the
    // next, the compiler must keep them in order (same LDS array) and the wave pays one LDS round trip per group
    constexpr int NG = (NPF + 1) / 2;
    f32x4 acc[NG];
    float vin[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int pf = half + 2 * u;
      vin[u] = Vf[(pf < NPF ? pf : 0) * 16 + fr];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < nks1) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          const int pf = half + 2 * u;
          if (pf < NPF) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xh + (pf * 16 + fr) * PX + fq * 16 + ks * 64);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc[u], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int pf = half + 2 * u;
      if (pf < NPF) {
        const int px = pf * 16 + fr;
        if (G::EF32) {
          float4 o;
          o.x = fmaxf(__builtin_fmaf(acc[u][0], s1v[0], b1v[0]), 0.f) * vin[u]; o.y = fmaxf(__builtin_fmaf(acc[u][1], s1v[1], b1v[1]), 0.f) * vin[u];
          o.z = fmaxf(__builtin_fmaf(acc[u][2], s1v[2], b1v[2]), 0.f) * vin[u]; o.w = fmaxf(__builtin_fmaf(acc[u][3], s1v[3], b1v[3]), 0.f) * vin[u];
          *reinterpret_cast<float4*>(E + px * PEE + (cf * 16 + fq * 4) * 4) = o;
        } else {
          bf16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (T)(fmaxf(__builtin_fmaf(acc[u][q], s1v[q], b1v[q]), 0.f) * vin[u]);
          *reinterpret_cast<bf16x4*>(E + px * PEE + (cf * 16 + fq * 4) * 2) = o;
        }
      }
    }
  };

  // ---- set-up: chunk 0's weights and constants (requested first), input tile, bn3, validity
  Pre pre;
  request(0, pre);
  {
    // the input tile: every load of a lane in flight at once (a rolled loop pays one memory round trip per trip)
    constexpr int NI = (HPp * 16 + NT - 1) / NT;          // trips at the widest input (16 vectors per pixel)
    uint4 xv[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = tid + t * NT;
      const int hp = i / nvecp, v = i - hp * nvecp;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = i < HPp * nvecp && hp < HP && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W && v < nvec;
      xv[t] = *reinterpret_cast<const uint4*>(g.x + (ok ? (((long)b * g.H + iy) * g.W + ix) * g.ldx + v * 8 : 0));
      if (!ok) xv[t] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = tid + t * NT;
      const int hp = i / nvecp, v = i - hp * nvecp;
      if (i < HPp * nvecp) *reinterpret_cast<uint4*>(Xh + hp * PX + v * 16) = xv[t];
    }
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
  stage(0, pre);
  if (nchunk > 1) request(1, pre);
  __syncthreads();
  expand(0);
  __syncthreads();

  // project accumulators: this wave's pixel fragment x its share of the output-channel fragments
  const int pxf = wave % G::NPX, part = wave / G::NPX;
  f32x4 acc3[G::MAXF3];
#pragma unroll
  for (int i = 0; i < G::MAXF3; ++i) acc3[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  BK_T(t1);
  for (int c = 0; c < nchunk; ++c) {
    const int set = c & 1;
    BK_T(ta);
    // ---- phase 1: the NEXT chunk's weights / constants (requested one phase ago) -> the other set; D = relu(bn2(dw3x3(E)))
    if (c + 1 < nchunk) {
      stage(set ^ 1, pre);
      if (c + 2 < nchunk) request(c + 2, pre);
    }
    BK_T(tb);
    if (G::EF32) {
      // lane = 4 channels of a 2 x 2 block of output pixels: its 4 x 4 window is read once (16 16-byte reads: 4 per output instead of 9),
      // the 9 taps once per four outputs, packed f32 fmas.  (The stage is bound by LDS reads and instruction issue at two waves per SIMD.)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float* K = Kc(set);
      const int v = tid & 15, item = tid >> 4;                 // 16 channel quads x (TP / 4) blocks
      if (item < TP / 4) {
        f32x2 wt[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float4 a = *reinterpret_cast<const float4*>(K + 4 * CM + t * CM + v * 4);
          wt[t][0] = (f32x2){a.x, a.y}; wt[t][1] = (f32x2){a.z, a.w};
        }
        const int by = item / (TW / 2), bx = item - by * (TW / 2);
        const int qy = by * 2, qx = bx * 2;
        const unsigned char* e0 = E + (qy * HW + qx) * PEE + v * 16;
        f32x2 sum[2][2][2];
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
          for (int ox = 0; ox < 2; ++ox) { sum[oy][ox][0] = (f32x2){0.f, 0.f}; sum[oy][ox][1] = (f32x2){0.f, 0.f}; }
#pragma unroll
        for (int wy = 0; wy < 4; ++wy) {
          f32x2 px[4][2];
#pragma unroll
          for (int wx = 0; wx < 4; ++wx) {
            const float4 a = *reinterpret_cast<const float4*>(e0 + (wy * HW + wx) * PEE);
            px[wx][0] = (f32x2){a.x, a.y}; px[wx][1] = (f32x2){a.z, a.w};
          }
#pragma unroll
          for (int oy = 0; oy < 2; ++oy) {
            const int dy = wy - oy;
            if (dy >= 0 && dy < 3) {
#pragma unroll
              for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                  sum[oy][ox][0] = __builtin_elementwise_fma(px[ox + dx][0], wt[dy * 3 + dx][0], sum[oy][ox][0]);
                  sum[oy][ox][1] = __builtin_elementwise_fma(px[ox + dx][1], wt[dy * 3 + dx][1], sum[oy][ox][1]);
                }
            }
          }
        }
        const float4 sc = *reinterpret_cast<const float4*>(K + 2 * CM + v * 4);
        const float4 sh = *reinterpret_cast<const float4*>(K + 3 * CM + v * 4);
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
          for (int ox = 0; ox < 2; ++ox) {
            bf16x4 o;
            o[0] = (T)fmaxf(__builtin_fmaf(sum[oy][ox][0][0], sc.x, sh.x), 0.f); o[1] = (T)fmaxf(__builtin_fmaf(sum[oy][ox][0][1], sc.y, sh.y), 0.f);
            o[2] = (T)fmaxf(__builtin_fmaf(sum[oy][ox][1][0], sc.z, sh.z), 0.f); o[3] = (T)fmaxf(__builtin_fmaf(sum[oy][ox][1][1], sc.w, sh.w), 0.f);
            *reinterpret_cast<bf16x4*>(D + ((qy + oy) * TW + qx + ox) * PE + v * 8) = o;
          }
      }
    } else {
      const float* K = Kc(set);
      const int v = tid & 7;
      float wt[9][8];
#pragma unroll
      for (int t = 0; t < 9; ++t) V8<float>::load(K + 4 * CM + t * CM + v * 8, wt[t]);
      float s2v[8], b2v[8];
      V8<float>::load(K + 2 * CM + v * 8, s2v); V8<float>::load(K + 3 * CM + v * 8, b2v);
#pragma unroll
      for (int ps = 0; ps < TP / 64; ++ps) {
        const int q = (tid >> 3) + ps * 64;
        const int qy = q / TW, qx = q - qy * TW;
        const unsigned char* e0 = E + ((qy * S) * HW + qx * S) * PEE + v * 16;
        uint4 r[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) r[t] = *reinterpret_cast<const uint4*>(e0 + ((t / 3) * HW + (t % 3)) * PEE);
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float* w = wt[t];
          sum[0] = __builtin_fmaf(blo(r[t].x), w[0], sum[0]); sum[1] = __builtin_fmaf(bhi(r[t].x), w[1], sum[1]);
          sum[2] = __builtin_fmaf(blo(r[t].y), w[2], sum[2]); sum[3] = __builtin_fmaf(bhi(r[t].y), w[3], sum[3]);
          sum[4] = __builtin_fmaf(blo(r[t].z), w[4], sum[4]); sum[5] = __builtin_fmaf(bhi(r[t].z), w[5], sum[5]);
          sum[6] = __builtin_fmaf(blo(r[t].w), w[6], sum[6]); sum[7] = __builtin_fmaf(bhi(r[t].w), w[7], sum[7]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)fmaxf(__builtin_fmaf(sum[j], s2v[j], b2v[j]), 0.f);
        *reinterpret_cast<bf16x8*>(D + q * PE + v * 16) = o;
      }
    }
    __syncthreads();
    BK_T(tc);
    // ---- phase 2: acc3 += D W3c^T, and the NEXT chunk's expand product into E (this chunk's E has been consumed)
    {
      const unsigned char* drow = D + (pxf * 16 + fr) * PE + fq * 16;
      const unsigned char* w3 = W3c(set) + fr * PE + fq * 16;
      bf16x8 df[CM / 32];
#pragma unroll
      for (int ks = 0; ks < CM / 32; ++ks) df[ks] = *reinterpret_cast<const bf16x8*>(drow + ks * 64);
#pragma unroll
      for (int ii = 0; ii < G::MAXF3; ++ii) {
        const int i = part + ii * G::WPP;
        if (i < NCF) {
#pragma unroll
          for (int ks = 0; ks < CM / 32; ++ks) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w3 + i * 16 * PE + ks * 64);
            acc3[ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df[ks], acc3[ii], 0, 0, 0);
          }
        }
      }
    }
    BK_T(td);
    if (c + 1 < nchunk) expand(set ^ 1);
    __syncthreads();
#ifdef TSS_TIMING
    BK_T(te);
    ph[0] += tb - ta; ph[1] += tc - tb; ph[2] += td - tc; ph[3] += te - td;
#endif
  }
  BK_T(t2);

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
#ifdef TSS_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BK_T(t3);
  if (threadIdx.x == 0) {
    atomicAdd(&g_bk_timing[0], t1 - t0); atomicAdd(&g_bk_timing[1], ph[0]); atomicAdd(&g_bk_timing[2], ph[1]); atomicAdd(&g_bk_timing[3], ph[2]);
    atomicAdd(&g_bk_timing[4], ph[3]); atomicAdd(&g_bk_timing[5], t3 - t2); atomicAdd(&g_bk_timing[7], 1ull);
  }
#endif
}

template <int S, int TH, int TW>
bool fits(const BkArgs& g) { return Geo<S, TH, TW>::lds(g.Cin, g.Cout) <= 160 * 1024; }

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

#ifdef TSS_TIMING
extern "C" int tss_debug_bk_timing(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bk_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bk_timing), z, sizeof(z)); }
  return 0;
}
#endif

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
  // 8 x 16 tiles when they fill the chip (and fit the LDS), else 8 x 8, else 4 x 8 (a block per CU matters more than the halo: the 1/32
  // maps of a 2048 x 4096 image are 128 tiles of 8 x 8); stride 2: 4 x 16 outputs over a 9 x 33 input tile
  const long t88 = (long)B * ((g.Ho + 7) / 8) * ((g.Wo + 7) / 8);
  if (stride == 2) rc = launch<2, 4, 16>(g, (hipStream_t)stream);
  else if ((long)B * ((g.Ho + 7) / 8) * ((g.Wo + 15) / 16) >= 256 && fits<1, 8, 16>(g)) rc = launch<1, 8, 16>(g, (hipStream_t)stream);
  else if (t88 >= 200) rc = launch<1, 8, 8>(g, (hipStream_t)stream);
  else rc = launch<1, 4, 8>(g, (hipStream_t)stream);
  if (rc) return rc;
  return tss::check_last("bneck_eval_fwd");
}

}  // extern "C"

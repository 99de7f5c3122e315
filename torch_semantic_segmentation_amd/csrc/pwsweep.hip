// Backward of the LARGE 1x1 layers in ONE sweep (round 4): input gradient AND weight gradient from a single pass over (e, y, x).
//
// pwbwd.hip does this for the few-channel layers; its 128-channel instances lost to the two separate kernels (pwfast_kernel<true> +
// wgfast_kernel) because they ran out of registers: spills put scratch waits into the tile loop, and every such wait drains the
// prefetched loads of the next tile.  The separate kernels read e and y -- the big operands of a 1x1 layer's backward -- twice:
// P (2 Cout + 2 Cin) + P (2 Cout + Cin) elements against P (2 Cout + 2 Cin) here.  This file is the form that fits:
//
//   * ONE 512-thread block per CU (8 waves, 2 per SIMD, 256 registers each); the block keeps its whole [Cout][Cin] weight-gradient
//     tile in accumulators over all its pixel tiles (32 - 48 registers per lane) and writes it once, to a workspace row in the
//     parameter's own order (summed with the depthwise rows by tss_dw_reduce_many at the end of the backward pass);
//   * the tile loads (16 bytes per lane) are inline-assembly requests with hand-placed `s_waitcnt vmcnt(N)` (N = the epilogue's
//     stores, which are younger): tile t + 1 is requested right after tile t's registers went to LDS and flies under the matrix
//     products and the epilogue of tile t -- the compiler's wait-count pass never sees them, so it can neither drain them at a
//     barrier nor at the loop's back edge (what the depthwise row pipeline taught);
//   * the waves are SPECIALISED for staging -- NGW of them normalise g = BN'(e, y), the others a = relu?(BN(x)) -- so a lane's
//     channel vector, and with it the folded BatchNorm constants in its registers, is fixed;
//   * g and a are written to LDS ONCE, pixel-major (16-byte stores), in the dual-use XOR image of MI355X's transposing LDS read:
//     the input-gradient product reads g row-wise (ds_read_b128), the weight-gradient product reads g^T and a^T out of the SAME
//     images with ds_read_b64_tr_b16 (pwbwd.hip stores three images through transposing 8-byte stores); the raw x needed by the
//     epilogue (ReLU mask, BatchNorm-backward sums of the producer) is a third image, copied as it arrived;
//   * the images are double-buffered: one barrier per tile;
//   * W^T (bf16 shadow, [Cin][Cout]) lives in REGISTERS as MFMA fragments for the whole launch (24 - 48 registers): no weight
//     image in LDS and no weight reads in the loop.
// Shapes (Cin -> Cout): 128 -> 128 (TM = 64), 64 -> 384 and 384 -> 64 (TM = 32), 32 -> 192 (64) and 192 -> 32 (32): FastSCNN's classifier / fusion layers at 1/8
// resolution and its 6x bottleneck expand / project layers at 1/8 and 1/16 resolution (TSS/models/fastscnn.py:138-161, 188-199);
// P must be a multiple of TM (else the caller keeps the two-kernel path).
#include <cstdlib>

#include "common.h"

namespace {

typedef bf16_t T;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) short v4s;

struct SwArgs {
  long P;
  const T* e; long lde; const T* y; long ldy;                 // y NULL: no BatchNorm statistics behind the layer (g = ga * e)
  const float* ga; const float* gb; const float* gce; const float* gmu;
  const T* wT; long ldwT;                                      // bf16 [Cin][Cout]
  const T* x; long ldx; const float* xm; const float* xs; const float* xb; int x_relu, x_pending;
  const T* radd; long ldr;                                     // materialised input only: e_in = g W + radd
  T* ein; long ldei; double* stats; float* ws; int gslots;
};

__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// requests the compiler does not count (see above); base: wave-uniform, off: this lane's byte offset
__device__ __forceinline__ void req16(u32x4& dst, const void* base, int off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}
__device__ __forceinline__ void req8(u32x2& dst, const void* base, int off) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}

// byte offset of 16-byte chunk ch (0..15) of row `row` in a [rows][256 B] image that serves ds_read_b128 row reads and
// ds_read_b64_tr_b16 transposed reads alike (cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, const unsigned char* hi) {
  union { v4s h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lo);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)hi);
  return u.v;
}

// FN / FK: 16-channel fragments of Cout / Cin.  TM: pixels per tile.  NGW: waves that stage g (the other 8 - NGW stage a).
// DXI x DXM: input-channel fragments x pixel fragments of the input-gradient tile a wave owns; DWN x DWK: output-channel x
// input-channel fragments of the weight-gradient tile a wave owns.
template <int FN, int FK, int TM_, int NGW_, int DXI_, int DXM_, int DWN_, int DWK_, bool KLDS_ = false>
struct Cfg {
  static constexpr bool KLDS = KLDS_;      // the folded BatchNorm constants of a lane's channel vector are re-read from LDS per tile (24 registers less)
  static constexpr int TM = TM_, NGW = NGW_, DXI = DXI_, DXM = DXM_, DWN = DWN_, DWK = DWK_;
  static constexpr int NC = FN * 16, KC = FK * 16, NVG = NC / 8, NVA = KC / 8, NAW = 8 - NGW;
  static constexpr int RPG = NGW * 64 / NVG, NPG = TM / RPG, RPA = NAW * 64 / NVA, NPA = TM / RPA;
  static constexpr int CBG = (NC + 127) / 128, CBA = (KC + 127) / 128;
  static constexpr int NKX = NC / 32, NKW = TM / 32, PXF = TM / 16, DXW = DXI * DXM;
  static constexpr int NPXG = PXF / DXM, NKCG = FN / DWN;
  static constexpr int GI = CBG * TM * 256, AI = CBA * TM * 256, BUF = GI + 2 * AI;      // g, a, raw x (XP)
  static constexpr int LDS = 2 * BUF + (3 * NC + 5 * KC) * 4;
  static_assert((NGW * 64) % NVG == 0 && TM % RPG == 0 && (NAW * 64) % NVA == 0 && TM % RPA == 0, "staging maps");
  static_assert(NPG <= 4 && NPA <= 4, "staging registers");
  static constexpr int NR = (2 * NPG > 4 || NPA > 4) ? 8 : 4;      // 16-byte request registers of a lane (g: e and y per pass, a: x per pass)
  static_assert(FK % DXI == 0 && PXF % DXM == 0 && (FK / DXI) * (PXF / DXM) == 8, "input-gradient tile split over 8 waves");
  static_assert(FN % DWN == 0 && FK % DWK == 0 && (FN / DWN) * (FK / DWK) == 8, "weight-gradient tile split over 8 waves");
  static_assert(NC % 32 == 0 && TM % 32 == 0, "k-steps");
};

template <int N> __device__ __forceinline__ void arrived(u32x4 (&R)[8], u32x2 (&Q)[4]) {
  asm volatile("s_waitcnt vmcnt(%12)"
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]),
                 "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3])
               : "n"(N));
}
template <int N> __device__ __forceinline__ void arrived(u32x4 (&R)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7])
               : "n"(N));
}
template <int N> __device__ __forceinline__ void arrived(u32x4 (&R)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void arrived(u32x4 (&R)[4], u32x2 (&Q)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3])
               : "n"(N));
}

// XP: x is a producer's raw output with its BatchNorm (+ReLU) pending -- e_in is masked, the producer's BatchNorm-backward sums leave as a
// slab row, the raw x travels to the epilogue through a third LDS image.  !XP: x is a materialised tensor -- optional skip gradient (radd)
// prefetched in the fragment layout, no statistics.
template <class C, bool XP>
__global__ __launch_bounds__(512, 2) void pwsweep_kernel(const SwArgs g) {
  static_assert(XP || C::DXW <= 4, "skip-gradient registers");
  constexpr int NC = C::NC, KC = C::KC, TM = C::TM;
  extern __shared__ __align__(16) unsigned char smem[];
  float* Cst = reinterpret_cast<float*>(smem + 2 * C::BUF);
  float* Cg = Cst;                 // [3][NC]: g = ca * e + cb * y + cc
  float* Ca = Cg + 3 * NC;         // [2][KC]: a = relu?(x * as + ab)
  float* Ec = Ca + 2 * KC;         // [3][KC]: producer's mean / scale / bias (ReLU mask + statistics in the epilogue)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const bool isG = wave < C::NGW;

  const int xcd = (int)blockIdx.x & 7, gslot = (int)blockIdx.x >> 3;
  const long ntiles = g.P / TM;
  const long per = (ntiles + 7) >> 3;
  const long t_begin = xcd * per + gslot;
  long t_end = xcd * per + per;
  if (t_end > ntiles) t_end = ntiles;

  // ---- this wave's share of the two products
  const int pxg = wave % C::NPXG, cig = wave / C::NPXG;          // input gradient: pixel fragments pxg * DXM + m, channels cig * DXI + i
  const int kcg = wave % C::NKCG, cig2 = wave / C::NKCG;         // weight gradient: rows kcg * DWN + u, columns cig2 * DWK + j

  // ---- staging map of this lane (fixed channel vector: its BatchNorm constants stay in registers)
  int voff0 = 0, voff1 = 0, ldsoff[4] = {0, 0, 0, 0};
  if (isG) {
    const int cv = tid % C::NVG, r0 = tid / C::NVG;
    voff0 = (int)((r0 * g.lde + cv * 8) * 2);
    voff1 = (int)((r0 * (g.y ? g.ldy : g.lde) + cv * 8) * 2);
#pragma unroll
    for (int p = 0; p < C::NPG; ++p) ldsoff[p] = (cv >> 4) * TM * 256 + img_off(r0 + p * C::RPG, cv & 15);
  } else {
    const int ta = tid - C::NGW * 64;
    const int cv = ta % C::NVA, r0 = ta / C::NVA;
    voff0 = (int)((r0 * g.ldx + cv * 8) * 2);
#pragma unroll
    for (int p = 0; p < C::NPA; ++p) ldsoff[p] = C::GI + (cv >> 4) * TM * 256 + img_off(r0 + p * C::RPA, cv & 15);
  }
  const int voffr = (int)((fr * g.ldr + fq * 4) * 2);             // the skip gradient under this lane's outputs (fragment layout)
  const T* yb = g.y ? g.y : g.e;
  const long ldyb = g.y ? g.ldy : g.lde;

  u32x4 R[C::NR];      // (deliberately not initialised: a zero-fill would be compiler-made moves into request registers)
  u32x2 Q[4];
  auto issue = [&](long tile) {
    const long p0 = tile * TM;
    if (isG) {
#pragma unroll
      for (int p = 0; p < C::NPG; ++p) {
        req16(R[2 * p], g.e + (p0 + p * C::RPG) * g.lde, voff0);
        req16(R[2 * p + 1], yb + (p0 + p * C::RPG) * ldyb, voff1);
      }
    } else {
#pragma unroll
      for (int p = 0; p < C::NPA; ++p) req16(R[p], g.x + (p0 + p * C::RPA) * g.ldx, voff0);
    }
    if (!XP && g.radd) {
#pragma unroll
      for (int i = 0; i < C::DXI; ++i)
#pragma unroll
        for (int m = 0; m < C::DXM; ++m)
          req8(Q[(i * C::DXM + m) & 3], g.radd + (p0 + (pxg * C::DXM + m) * 16) * g.ldr + (cig * C::DXI + i) * 16, voffr);
    }
  };
  if (t_begin < t_end) issue(t_begin);

  // ---- block set-up under the first tile's loads: constants -> LDS, W^T fragments -> registers
  for (int c = tid; c < NC; c += 512) {
    const bool hy = g.y != nullptr;
    const float a = g.ga ? g.ga[c] : 1.f, b = hy ? g.gb[c] : 0.f;
    const float ce = hy ? g.gce[c] : 0.f, mu = hy ? g.gmu[c] : 0.f;
    Cg[c] = a; Cg[NC + c] = b; Cg[2 * NC + c] = hy ? -(a * ce) - b * mu : 0.f;
  }
  for (int c = tid; c < KC; c += 512) {
    const bool hs = g.xs != nullptr;
    const float sc = hs ? g.xs[c] : 1.f, bb = (hs && g.xb) ? g.xb[c] : 0.f, mm = (hs && g.xm) ? g.xm[c] : 0.f;
    Ca[c] = sc; Ca[KC + c] = __builtin_fmaf(-mm, sc, bb);
    if (XP) { Ec[c] = mm; Ec[KC + c] = sc; Ec[2 * KC + c] = bb; }
  }
  bf16x8 wreg[C::DXI][C::NKX];
#pragma unroll
  for (int i = 0; i < C::DXI; ++i)
#pragma unroll
    for (int ks = 0; ks < C::NKX; ++ks)
      wreg[i][ks] = *reinterpret_cast<const bf16x8*>(g.wT + (long)((cig * C::DXI + i) * 16 + fr) * g.ldwT + ks * 32 + fq * 8);
  // pinned: left to itself the compiler re-loads these (invariant) fragments from memory inside the tile loop to save registers, and its
  // waits for them drain the prefetched tile
#pragma unroll
  for (int i = 0; i < C::DXI; ++i)
#pragma unroll
    for (int ks = 0; ks < C::NKX; ++ks) asm volatile("" : "+v"(wreg[i][ks]));
  __syncthreads();
  // this lane's folded constants: g-stagers (ca, cb, cc), a-stagers (as, ab, -: the third vector reads the start of Ec and is not used)
  const float* kbase = isG ? Cg + (tid % C::NVG) * 8 : Ca + ((tid - C::NGW * 64) % C::NVA) * 8;
  const int kstride = isG ? NC : KC;
  float K[24];
#define TSS_SW_LOAD_K() do { V8<float>::load(kbase, K); V8<float>::load(kbase + kstride, K + 8); V8<float>::load(kbase + 2 * kstride, K + 16); } while (0)
  TSS_SW_LOAD_K();
  const float relu_lo = g.x_relu ? 0.f : -TSS_INF;

  // ---- LDS read offsets of this lane (inside one image buffer)
  const int swz = ((fr & 3) << 2) | ((fr >> 2) & 3);             // row reads: row = 16 m + fr
  int rowoff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rowoff[c] = 256 * fr + 16 * (((c << 2) | fq) ^ swz);
  // transposed reads: the block of rows (pixels) fq * 8 + 4 h .. + 3, columns (channels) 16 f .. 16 f + 15; this lane supplies the
  // address of row fr >> 2, channels 4 (fr & 3) .. + 3
  int troffG[C::DWN][2], troffA[C::DWK][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = fq * 8 + 4 * h + (fr >> 2);
#pragma unroll
    for (int u = 0; u < C::DWN; ++u) {
      const int f = kcg * C::DWN + u;
      troffG[u][h] = (f >> 3) * TM * 256 + img_off(row, (f & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
    }
#pragma unroll
    for (int j = 0; j < C::DWK; ++j) {
      const int f = cig2 * C::DWK + j;
      troffA[j][h] = C::GI + (f >> 3) * TM * 256 + img_off(row, (f & 7) * 2 + ((fr & 3) >> 1)) + 8 * (fr & 1);
    }
  }

  f32x4 dwacc[C::DWN][C::DWK];
#pragma unroll
  for (int u = 0; u < C::DWN; ++u)
#pragma unroll
    for (int j = 0; j < C::DWK; ++j) dwacc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NST = XP ? C::DXI : 1;
  float st1[NST][4], st2[NST][4];
#pragma unroll
  for (int i = 0; i < NST; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[i][q] = 0.f; st2[i][q] = 0.f; }

  // the first tile has to be there before the loop; inside the loop ONE wait, at the end of an iteration, covers the next tile (a wait
  // at the top with a first-iteration special case made the compiler copy the request registers in FRONT of the tied wait statement)
  if (XP) arrived<0>(R); else arrived<0>(R, Q);
  int buf = 0;
  for (long tile = t_begin; tile < t_end; tile += g.gslots) {
    const long p0 = tile * TM;
    unsigned char* img = smem + buf * C::BUF;
    buf ^= 1;
    // ---- the tile that has arrived: normalise, write the images
    if (C::KLDS) TSS_SW_LOAD_K();
    if (isG) {
#pragma unroll
      for (int p = 0; p < C::NPG; ++p) {
        const u32x4 re = R[2 * p], ry = R[2 * p + 1];
        float v[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[2 * h] = K[2 * h] * blo(re[h]) + (K[8 + 2 * h] * blo(ry[h]) + K[16 + 2 * h]);
          v[2 * h + 1] = K[2 * h + 1] * bhi(re[h]) + (K[8 + 2 * h + 1] * bhi(ry[h]) + K[16 + 2 * h + 1]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
        *reinterpret_cast<bf16x8*>(img + ldsoff[p]) = o;
      }
    } else {
#pragma unroll
      for (int p = 0; p < C::NPA; ++p) {
        const u32x4 rx = R[p];
        if (XP) *reinterpret_cast<u32x4*>(img + C::AI + ldsoff[p]) = rx;      // raw copy for the epilogue
        float v[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[2 * h] = fmaxf(blo(rx[h]) * K[2 * h] + K[8 + 2 * h], relu_lo);
          v[2 * h + 1] = fmaxf(bhi(rx[h]) * K[2 * h + 1] + K[8 + 2 * h + 1], relu_lo);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
        *reinterpret_cast<bf16x8*>(img + ldsoff[p]) = o;
      }
    }
    // the skip gradient of THIS tile has to outlive the re-request of its registers
    constexpr int NQ = XP ? 1 : C::DXW;
    u32x2 qr[NQ];
    if (!XP) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(qr[i][0]), "=&v"(qr[i][1]) : "v"(Q[i & 3][0]), "v"(Q[i & 3][1]));
    }
    asm volatile("" ::: "memory");          // every use of the raw vectors (VALU reads, the LDS copy) is issued before their re-request
    if (tile + g.gslots < t_end) issue(tile + g.gslots);
    __syncthreads();

    // ---- input gradient: D[ci][p] = sum_kc W^T[ci][kc] g[p][kc]
    f32x4 acc[C::DXI][C::DXM];
#pragma unroll
    for (int i = 0; i < C::DXI; ++i)
#pragma unroll
      for (int m = 0; m < C::DXM; ++m) acc[i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::NKX; ++ks) {
      bf16x8 gf[C::DXM];
#pragma unroll
      for (int m = 0; m < C::DXM; ++m)
        gf[m] = *reinterpret_cast<const bf16x8*>(img + (ks >> 2) * TM * 256 + (pxg * C::DXM + m) * 16 * 256 + rowoff[ks & 3]);
#pragma unroll
      for (int i = 0; i < C::DXI; ++i)
#pragma unroll
        for (int m = 0; m < C::DXM; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[i][ks], gf[m], acc[i][m], 0, 0, 0);
    }
    // ---- weight gradient: D[kc][ci] += sum_p g[p][kc] a[p][ci], both operands through the transposing read
#pragma unroll
    for (int ks = 0; ks < C::NKW; ++ks) {
      bf16x8 gA[C::DWN], aB[C::DWK];
#pragma unroll
      for (int u = 0; u < C::DWN; ++u) gA[u] = tr_pair(img + ks * 32 * 256 + troffG[u][0], img + ks * 32 * 256 + troffG[u][1]);
#pragma unroll
      for (int j = 0; j < C::DWK; ++j) aB[j] = tr_pair(img + ks * 32 * 256 + troffA[j][0], img + ks * 32 * 256 + troffA[j][1]);
#pragma unroll
      for (int u = 0; u < C::DWN; ++u)
#pragma unroll
        for (int j = 0; j < C::DWK; ++j) dwacc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gA[u], aB[j], dwacc[u][j], 0, 0, 0);
    }
    // ---- epilogue of the input gradient: lane = pixel fr of a fragment, 4 consecutive channels
#pragma unroll
    for (int i = 0; i < C::DXI; ++i) {
      const int cf = cig * C::DXI + i;                 // input-channel fragment
      const int n = cf * 16 + fq * 4;
      float cmm[4] = {0.f, 0.f, 0.f, 0.f}, cms[4] = {1.f, 1.f, 1.f, 1.f}, cmb[4] = {0.f, 0.f, 0.f, 0.f};
      if (XP) {
        const float4 e0 = *reinterpret_cast<const float4*>(Ec + n);
        const float4 e1 = *reinterpret_cast<const float4*>(Ec + KC + n);
        const float4 e2 = *reinterpret_cast<const float4*>(Ec + 2 * KC + n);
        cmm[0] = e0.x; cmm[1] = e0.y; cmm[2] = e0.z; cmm[3] = e0.w;
        cms[0] = e1.x; cms[1] = e1.y; cms[2] = e1.z; cms[3] = e1.w;
        cmb[0] = e2.x; cmb[1] = e2.y; cmb[2] = e2.z; cmb[3] = e2.w;
      }
#pragma unroll
      for (int m = 0; m < C::DXM; ++m) {
        const int pm = pxg * C::DXM + m;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][m][q];
        bf16x4 o;
        if (XP) {
          const uint2 xr = *reinterpret_cast<const uint2*>(img + C::GI + C::AI + (cf >> 3) * TM * 256 + pm * 16 * 256 + 256 * fr
                                                           + 16 * ((((cf & 7) << 1) | (fq >> 1)) ^ swz) + 8 * (fq & 1));
          const float xc[4] = {blo(xr.x) - cmm[0], bhi(xr.x) - cmm[1], blo(xr.y) - cmm[2], bhi(xr.y) - cmm[3]};
          if (g.x_relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!(xc[q] * cms[q] + cmb[q] > 0.f)) v[q] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float rq = (float)o[q]; st1[XP ? i : 0][q] += rq; st2[XP ? i : 0][q] += rq * xc[q]; }
        } else {
          if (g.radd) {
            const u32x2 rr = qr[XP ? 0 : i * C::DXM + m];
            v[0] += blo(rr[0]); v[1] += bhi(rr[0]); v[2] += blo(rr[1]); v[3] += bhi(rr[1]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (T)v[q];
        }
        *reinterpret_cast<bf16x4*>(g.ein + (p0 + pm * 16 + fr) * g.ldei + n) = o;
      }
    }
    // ---- the next tile (requested above): younger than its requests are exactly this epilogue's DXW stores
    if (XP) arrived<C::DXW>(R); else arrived<C::DXW>(R, Q);
  }

  // ---- the block's weight-gradient tile -> its workspace row, [Cout][Cin] like the parameter (blocks without tiles write zeros)
  const int row = xcd + 8 * gslot;
  {
    float* wr = g.ws + (long)row * NC * KC;
#pragma unroll
    for (int u = 0; u < C::DWN; ++u)
#pragma unroll
      for (int j = 0; j < C::DWK; ++j) {
        const int kc0 = (kcg * C::DWN + u) * 16 + fq * 4, ci = (cig2 * C::DWK + j) * 16 + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[(long)(kc0 + q) * KC + ci] = dwacc[u][j][q];
      }
  }
  // ---- statistics slab row of this block: sum over the 16 pixel lanes of a fragment row, then over the pixel groups
  if (XP && g.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [NPXG][2][KC]
#pragma unroll
    for (int i = 0; i < C::DXI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u1 = row16_sum(st1[XP ? i : 0][q]), u2 = row16_sum(st2[XP ? i : 0][q]);
        if (fr == 0) {
          const int n = (cig * C::DXI + i) * 16 + fq * 4 + q;
          red[(pxg * 2 + 0) * KC + n] = u1;
          red[(pxg * 2 + 1) * KC + n] = u2;
        }
      }
    __syncthreads();
    for (int c = tid; c < KC; c += 512) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < C::NPXG; ++w) { a += (double)red[(w * 2 + 0) * KC + c]; b += (double)red[(w * 2 + 1) * KC + c]; }
      const int rows_used = 8 * g.gslots;
      g.stats[(long)row * 2 * KC + c] = a;
      g.stats[(long)row * 2 * KC + KC + c] = b;
      for (int rr = row + rows_used; rr < TSS_STAT_SLABS; rr += rows_used) {
        g.stats[(long)rr * 2 * KC + c] = 0.0;
        g.stats[(long)rr * 2 * KC + KC + c] = 0.0;
      }
    }
  }
}

//                FN  FK  TM NGW DXI DXM DWN DWK
typedef Cfg<8, 8, 64, 4, 2, 2, 2, 4> CfgSq;          // 128 -> 128
typedef Cfg<24, 4, 32, 6, 1, 1, 3, 4> CfgExpand;     // 64 -> 384
typedef Cfg<4, 24, 32, 2, 3, 2, 4, 3, true> CfgProject;    // 384 -> 64
typedef Cfg<12, 2, 64, 6, 1, 1, 3, 1> CfgExpand32;         // 32 -> 192  (ContextNet's context branch at 1/8 resolution of the 1/4 image)
typedef Cfg<2, 12, 32, 2, 3, 1, 1, 3> CfgProject32;        // 192 -> 32

inline int shape_kind(int Cin, int Cout) {
  if (Cin == 128 && Cout == 128) return 1;
  if (Cin == 64 && Cout == 384) return 2;
  if (Cin == 384 && Cout == 64) return 3;
  if (Cin == 32 && Cout == 192) return 4;
  if (Cin == 192 && Cout == 32) return 5;
  return 0;
}
inline bool is_expand(int kind) { return kind == 2 || kind == 4; }      // takes a materialised input (a block input)
inline bool is_project(int kind) { return kind == 3 || kind == 5; }     // takes a producer's raw output (pending BatchNorm + ReLU)
inline int tile_px(int kind) { return (kind == 1 || kind == 4) ? 64 : 32; }

inline int slots_for(long P, int kind) {
  long gs = (P / tile_px(kind) + 7) / 8;
  if (gs > 32) gs = 32;            // 8 * gs blocks: one per CU
  if (gs < 1) gs = 1;
  return (int)gs;
}

template <class C, bool XP>
void launch(SwArgs& g, int gs, hipStream_t stream) {
  g.gslots = gs;
  static tss::DevOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwsweep_kernel<C, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  hipLaunchKernelGGL((pwsweep_kernel<C, XP>), dim3(8 * gs), dim3(512), C::LDS, stream, g);
}

}  // namespace

extern int g_tss_disable_fast;

extern "C" {

// 1 when tss_pwconv_bwd_sweep covers the layer (shape, pitch-independent part) AND is the faster choice
int tss_pwconv_bwd_sweep_preferred(long P, int Cin, int Cout, int x_pending, int dtype) {
  static const int sw = getenv("TSS_PW_SWEEP") ? atoi(getenv("TSS_PW_SWEEP")) : 1;       // A/B switch: 0 = never
  static const long minp = getenv("TSS_PW_SWEEP_MINP") ? atol(getenv("TSS_PW_SWEEP_MINP")) : 32768;
  const int kind = shape_kind(Cin, Cout);
  if (is_expand(kind) && x_pending) return 0;       // the expand instances take a materialised input (a block input), the project instances a pending one
  if (is_project(kind) && !x_pending) return 0;
  return (sw && !g_tss_disable_fast && dtype == TSS_BF16 && kind && P >= minp && (P % tile_px(kind)) == 0) ? 1 : 0;
}

int tss_pwconv_bwd_sweep_rows(long P, int Cin, int Cout) {
  const int kind = shape_kind(Cin, Cout);
  return kind ? 8 * slots_for(P, kind) : 0;
}

int tss_pwconv_bwd_sweep(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                         const float* gmu, const void* wT_bf16, const void* x, long ldx, const float* in_mean,
                         const float* in_scale, const float* in_bias, int in_relu, int x_pending, const void* radd, long ldr,
                         void* e_in, long ldei, double* bstats, float* ws, long P, int Cin, int Cout, int dtype, void* stream) {
  const int kind = shape_kind(Cin, Cout);
  TSS_REQUIRE(dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(kind && P > 0 && (P % tile_px(kind)) == 0 && e && x && e_in && ws && wT_bf16, TSS_ERR_SHAPE);
  TSS_REQUIRE(!(is_expand(kind) && x_pending) && !(is_project(kind) && !x_pending), TSS_ERR_SHAPE);
  TSS_REQUIRE((lde % 8) == 0 && lde >= Cout && (ldx % 8) == 0 && ldx >= Cin && (ldei % 4) == 0 && ldei >= Cin, TSS_ERR_SHAPE);
  TSS_REQUIRE(lde < (1L << 20) && ldx < (1L << 20) && ldyr < (1L << 20) && ldr < (1L << 20), TSS_ERR_SHAPE);      // 32-bit lane offsets
  TSS_REQUIRE(!yraw || ((ldyr % 8) == 0 && ldyr >= Cout && ga && gb && gce && gmu), TSS_ERR_SHAPE);
  TSS_REQUIRE(!bstats || x_pending, TSS_ERR_SHAPE);
  TSS_REQUIRE(!radd || (!x_pending && (ldr % 4) == 0 && ldr >= Cin), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(e) && tss::aligned16(x) && (!yraw || tss::aligned16(yraw)) && tss::aligned16(wT_bf16)
              && ((uintptr_t)e_in & 7u) == 0 && (!radd || ((uintptr_t)radd & 7u) == 0), TSS_ERR_ALIGN);
  SwArgs g = {};
  g.P = P;
  g.e = (const T*)e; g.lde = lde; g.y = (const T*)yraw; g.ldy = ldyr; g.ga = ga; g.gb = gb; g.gce = gce; g.gmu = gmu;
  g.wT = (const T*)wT_bf16; g.ldwT = Cout;
  g.x = (const T*)x; g.ldx = ldx; g.x_relu = in_relu; g.x_pending = x_pending;
  if (x_pending) { g.xm = in_mean; g.xs = in_scale; g.xb = in_bias; }
  g.radd = (const T*)radd; g.ldr = ldr;
  g.ein = (T*)e_in; g.ldei = ldei; g.stats = bstats; g.ws = ws;
  tss::ProfScope prof(TSS_K_PWCONV_BWD_DATA, (hipStream_t)stream,
                      ((double)P * Cout * (yraw ? 2 : 1) + (double)P * Cin * (radd ? 3 : 2)) * 2.0, 4.0 * (double)P * Cin * Cout);
  const int gs = slots_for(P, kind);
  if (kind == 1) { if (x_pending) launch<CfgSq, true>(g, gs, (hipStream_t)stream); else launch<CfgSq, false>(g, gs, (hipStream_t)stream); }
  else if (kind == 2) launch<CfgExpand, false>(g, gs, (hipStream_t)stream);
  else if (kind == 3) launch<CfgProject, true>(g, gs, (hipStream_t)stream);
  else if (kind == 4) launch<CfgExpand32, false>(g, gs, (hipStream_t)stream);
  else launch<CfgProject32, true>(g, gs, (hipStream_t)stream);
  return tss::check_last("pwconv_bwd_sweep");
}

}  // extern "C"

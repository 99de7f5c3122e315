// Shared device/host helpers for the gfx950 kernels of the FastSCNN / ContextNet hot path.
// Everything here is CDNA4-only: 64-lane waves, MFMA 16x16 tiles, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tss_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TSS_WAVE 64
// Per-channel statistics leave a kernel as one PARTIAL ROW per block ("slab"), written with plain stores into a
// [TSS_STAT_SLABS][2C] f64 buffer and summed by the finalize kernels: no atomics (512 blocks adding 2C values to
// the same addresses cost 50-100 us per launch on MI355X), and the sums are run-to-run deterministic.
#define TSS_STAT_SLABS 512
#define TSS_MAX_PERSISTENT_BLOCKS 1024  // 256 CUs x 4: enough waves in flight for HBM-bound loops

// ---------------------------------------------------------------------------------------------
// 8-wide channel vectors: the unit of every NHWC access (16 B of bf16, 32 B of f32 per lane).
template <typename T> struct V8;

template <> struct V8<float> {
  static __device__ __forceinline__ void load(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ float round(float x) { return x; }
  struct Raw { float4 a, b; };
  static __device__ __forceinline__ Raw load_raw(const float* p) {
    Raw r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r;
  }
  static __device__ __forceinline__ void unpack(const Raw& r, float v[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
  }
};

template <> struct V8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float v[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(p) = o;
  }
  static __device__ __forceinline__ float round(float x) { return (float)(bf16_t)x; }
  typedef uint4 Raw;  // issue all loads of a tap window first, convert afterwards
  static __device__ __forceinline__ Raw load_raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void unpack(const Raw& r, float v[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
  }
};

// 4-wide access used by the MFMA epilogues (one lane owns 4 consecutive channels of one pixel).
template <typename T> struct V4;
template <> struct V4<float> {
  static __device__ __forceinline__ void load(const float* p, float v[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  static __device__ __forceinline__ void store(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct V4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float v[4]) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[4]) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x4*>(p) = o;
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

// Branch-free "ReLU and/or zero": clamp(a, lo, hi) as ONE v_med3_f32.
//   relu, valid  : (0, +inf)     identity, valid : (-inf, +inf)     invalid tap / padding : (0, 0)
// A runtime relu flag written as `flag ? max(a,0) : a` costs a compare + select per element instead.
__device__ __forceinline__ float clamp3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
#define TSS_INF __builtin_inff()

// ---------------------------------------------------------------------------------------------
// Reductions.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// XCD-aware tile order (MI355X: 8 XCDs, blocks are dealt round-robin, so b and b+8 share an L2).
// Every XCD owns one contiguous band of the tile range and its resident blocks sweep that band
// together (slot-interleaved), so neighbouring tiles, which share halo rows or operand panels, are
// in flight on the same 4 MiB L2 at about the same time.  Placement changes speed only, never
// results.  The grid must be a multiple of 8 blocks (tss::persistent_blocks guarantees it).
struct TileRange { int begin, end, step; };
// `lead` blocks (a multiple of 8) in front of the grid do something else (carried reductions): the sweep is over the
// remaining gridDim.x - lead blocks, and blockIdx.x - lead keeps its XCD.
__device__ __forceinline__ TileRange xcd_tiles(int ntiles, int lead = 0) {
  const int bid = (int)blockIdx.x - lead;
  const int xcd = bid & 7;
  const int slot = bid >> 3;
  const int slots = ((int)gridDim.x - lead) >> 3;
  const int per = (ntiles + 7) >> 3;
  const int b0 = xcd * per;
  int b1 = b0 + per;
  if (b1 > ntiles) b1 = ntiles;
  TileRange r;
  r.begin = b0 + slot; r.end = b1; r.step = slots;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Bilinear (align_corners=True) index arithmetic, identical to torch's upsample_bilinear2d: scale = (in-1)/(out-1)
// in f32 (0 when out == 1), src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap ac_tap(float scale, int dst, int in) {
  const float src = scale * (float)dst;
  Tap t;
  t.i0 = (int)src;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}
// outputs whose taps can include source index i: a conservative window [lo, hi]
__device__ __forceinline__ void ac_window(float scale, int i, int out, int* lo, int* hi) {
  if (scale <= 0.f) { *lo = 0; *hi = out - 1; return; }
  int l = (int)floorf((float)(i - 1) / scale) - 1;
  int h = (int)ceilf((float)(i + 1) / scale) + 1;
  *lo = l < 0 ? 0 : l;
  *hi = h > out - 1 ? out - 1 : h;
}
// weight with which output `dst` reads source index i (0 if it does not)
__device__ __forceinline__ float ac_weight(float scale, int dst, int in, int i) {
  const Tap t = ac_tap(scale, dst, in);
  float w = 0.f;
  if (t.i0 == i) w += t.l0;
  if (t.i1 == i) w += t.l1;
  return w;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: 4 VALU adds with a row
// rotate modifier instead of 4 ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// ---------------------------------------------------------------------------------------------
// Host side: error mapping + optional per-kernel profiling (HIP events on the launch stream).
namespace tss {

int check_last(const char* what);

struct ProfScope {
  ProfScope(int kernel_id, hipStream_t stream, double alg_bytes, double flops);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

// Grid for a persistent, tile-looping kernel: a multiple of 8 (one slot per XCD), at most `cap`.
inline int persistent_blocks(long tiles, int cap = TSS_MAX_PERSISTENT_BLOCKS) {
  long b = tiles < 1 ? 1 : tiles;
  if (b > cap) b = cap;
  return (int)((b + 7) / 8 * 8);
}

// hipFuncSetAttribute is per device: one flag per device ordinal (the call is idempotent, so a race between the forward and
// the autograd thread is harmless)
struct DevOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    d = (d >= 0 && d < 64) ? d : 0;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace tss

#define TSS_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)

// Tail of LEDNet's split-shuffle-non-bottleneck unit (SSnbtBlock.forward, TSS/models/lednet.py:112-124) in ONE pass each way:
//     y   = cat([bn_l(left), bn_r(right)], 1)          the two branches' pending BatchNorms
//     y   = dropout2d(y)                               a [B, C] multiplier (0 or 1 / (1 - p)), drawn by the caller
//     out = channel_shuffle(relu(x + y), 2)            out[2 j + g] = (.)[g C/2 + j]
// As four operators (concat_joined, scale_rows, join, channel_shuffle) this is nine tensor passes forward and ten backward on maps of
// 134 - 268 MB (7.6 ms of LEDNet's 43 ms step, profiles/r04_step_kernels_lednet.txt); fused it reads the two raw branch outputs and x and
// writes the unit's output (3 passes), and backward reads d(out), out and the raw branch outputs and writes the branches' gradient and the
// skip gradient (5 passes; 4 without dropout, where the two are the same tensor) while taking the BatchNorm-backward sums of both branches.
// A lane owns 8 channels of each branch = 16 consecutive OUTPUT channels of one pixel: every access is a whole 16-byte (bf16) vector.
#include "common.h"

namespace {

constexpr int NT = 256;

struct TailArgs {
  const void* l; long ldl; const float* ml; const float* sl; const float* bl;      // left raw, its BatchNorm as (mean, scale, shift)
  const void* r; long ldr; const float* mr; const float* sr; const float* br;
  const void* x; long ldx;
  const float* m;                                      // [B][C] dropout multipliers or NULL
  void* out; long ldo;
  const void* dout; long lddo;                         // backward
  void* e; long lde; void* gs; long ldgs;              // e: [P][C] (left half | right half) gradient of the BatchNorm outputs; gs: skip gradient
  double* stats_l; double* stats_r;                    // BatchNorm-backward slab rows [TSS_STAT_SLABS][2 * C/2] or NULL
  long HW; int B, C;
};

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<float> { typedef double type; };

__device__ __forceinline__ void coef8(const float* p, const float* safe, int c0, float dflt, float out[8]) {
  const float* q = p ? p + c0 : safe;
  float v[8];
  V4<float>::load(q, v); V4<float>::load(q + 4, v + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = p ? v[j] : dflt;
}

template <typename T>
__global__ __launch_bounds__(NT) void ssnbt_tail_fwd_kernel(const TailArgs g) {
  const int H = g.C >> 1, HV = H >> 3;                 // channels / vectors of one branch
  const int NPL = NT / HV;
  const int tid = threadIdx.x, cg = tid % HV, pl = tid / HV, j0 = cg * 8;
  if (pl >= NPL) return;
  const T* L = reinterpret_cast<const T*>(g.l); const T* R = reinterpret_cast<const T*>(g.r);
  const T* X = reinterpret_cast<const T*>(g.x); T* O = reinterpret_cast<T*>(g.out);
  const float* safe = reinterpret_cast<const float*>(g.l);
  float ml[8], sl[8], bl[8], mr[8], sr[8], br[8];
  coef8(g.sl, safe, j0, 1.f, sl); coef8(g.sl ? g.ml : nullptr, safe, j0, 0.f, ml); coef8(g.sl ? g.bl : nullptr, safe, j0, 0.f, bl);
  coef8(g.sr, safe, j0, 1.f, sr); coef8(g.sr ? g.mr : nullptr, safe, j0, 0.f, mr); coef8(g.sr ? g.br : nullptr, safe, j0, 0.f, br);
#pragma unroll
  for (int j = 0; j < 8; ++j) { bl[j] -= ml[j] * sl[j]; br[j] -= mr[j] * sr[j]; }      // y = raw * s + (b - mean * s)
  constexpr int JU = 2;
  const long stride = (long)gridDim.x * NPL;
  for (int b = 0; b < g.B; ++b) {
    float dl[8], dr[8];
    coef8(g.m ? g.m + (long)b * g.C : nullptr, safe, j0, 1.f, dl);
    coef8(g.m ? g.m + (long)b * g.C + H : nullptr, safe, j0, 1.f, dr);
    const long base = (long)b * g.HW;
    for (long q0 = (long)blockIdx.x * NPL + pl; q0 < g.HW; q0 += stride * JU) {
      typename V8<T>::Raw rl[JU], rr[JU], x0[JU], x1[JU];
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const long q = q0 + u * stride;
        const long p = base + (q < g.HW ? q : q0);
        rl[u] = V8<T>::load_raw(L + p * g.ldl + j0);
        rr[u] = V8<T>::load_raw(R + p * g.ldr + j0);
        x0[u] = V8<T>::load_raw(X + p * g.ldx + j0);
        x1[u] = V8<T>::load_raw(X + p * g.ldx + H + j0);
      }
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const long q = q0 + u * stride;
        if (q >= g.HW) break;
        const long p = base + q;
        float a[8], c[8], xa[8], xc[8], o0[8], o1[8];
        V8<T>::unpack(rl[u], a); V8<T>::unpack(rr[u], c); V8<T>::unpack(x0[u], xa); V8<T>::unpack(x1[u], xc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yl = fmaxf(xa[j] + dl[j] * (a[j] * sl[j] + bl[j]), 0.f);
          const float yr = fmaxf(xc[j] + dr[j] * (c[j] * sr[j] + br[j]), 0.f);
          if (j < 4) { o0[2 * j] = yl; o0[2 * j + 1] = yr; } else { o1[2 * (j - 4)] = yl; o1[2 * (j - 4) + 1] = yr; }
        }
        V8<T>::store(O + p * g.ldo + 2 * j0, o0);
        V8<T>::store(O + p * g.ldo + 2 * j0 + 8, o1);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void ssnbt_tail_bwd_kernel(const TailArgs g) {
  typedef typename Acc<T>::type A;
  __shared__ __align__(16) unsigned char smem[NT * 8 * 4 * sizeof(A)];
  const int H = g.C >> 1, HV = H >> 3;
  const int NPL = NT / HV;
  const int tid = threadIdx.x, cg = tid % HV, pl = tid / HV, j0 = cg * 8;
  const bool active = pl < NPL;
  const T* L = reinterpret_cast<const T*>(g.l); const T* R = reinterpret_cast<const T*>(g.r);
  const T* D = reinterpret_cast<const T*>(g.dout); const T* O = reinterpret_cast<const T*>(g.out);
  T* E = reinterpret_cast<T*>(g.e); T* GS = reinterpret_cast<T*>(g.gs);
  const float* safe = reinterpret_cast<const float*>(g.dout);
  const bool stats = g.stats_l != nullptr;
  float ml[8], mr[8];
  coef8(stats ? g.ml : nullptr, safe, active ? j0 : 0, 0.f, ml);
  coef8(stats ? g.mr : nullptr, safe, active ? j0 : 0, 0.f, mr);
  A s0l[8], s1l[8], s0r[8], s1r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0l[j] = 0; s1l[j] = 0; s0r[j] = 0; s1r[j] = 0; }
  if (active) {
    constexpr int JU = 2;
    const long stride = (long)gridDim.x * NPL;
    const T* pL = stats ? L : D; const long ldl = stats ? g.ldl : g.lddo;     // absent streams re-read d(out)
    const T* pR = stats ? R : D; const long ldr = stats ? g.ldr : g.lddo;
    for (int b = 0; b < g.B; ++b) {
      float dl[8], dr[8];
      coef8(g.m ? g.m + (long)b * g.C : nullptr, safe, j0, 1.f, dl);
      coef8(g.m ? g.m + (long)b * g.C + H : nullptr, safe, j0, 1.f, dr);
      const long base = (long)b * g.HW;
      for (long q0 = (long)blockIdx.x * NPL + pl; q0 < g.HW; q0 += stride * JU) {
        typename V8<T>::Raw d0[JU], d1[JU], o0[JU], o1[JU], rl[JU], rr[JU];
#pragma unroll
        for (int u = 0; u < JU; ++u) {
          const long q = q0 + u * stride;
          const long p = base + (q < g.HW ? q : q0);
          d0[u] = V8<T>::load_raw(D + p * g.lddo + 2 * j0); d1[u] = V8<T>::load_raw(D + p * g.lddo + 2 * j0 + 8);
          o0[u] = V8<T>::load_raw(O + p * g.ldo + 2 * j0); o1[u] = V8<T>::load_raw(O + p * g.ldo + 2 * j0 + 8);
          rl[u] = V8<T>::load_raw(pL + p * ldl + (stats ? j0 : 0)); rr[u] = V8<T>::load_raw(pR + p * ldr + (stats ? j0 : 0));
        }
#pragma unroll
        for (int u = 0; u < JU; ++u) {
          const long q = q0 + u * stride;
          if (q >= g.HW) break;
          const long p = base + q;
          float da[8], db[8], oa[8], ob[8], a[8], c[8], gl[8], gr[8], el[8], er[8];
          V8<T>::unpack(d0[u], da); V8<T>::unpack(d1[u], db); V8<T>::unpack(o0[u], oa); V8<T>::unpack(o1[u], ob);
          V8<T>::unpack(rl[u], a); V8<T>::unpack(rr[u], c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float dv0 = j < 4 ? da[2 * j] : db[2 * (j - 4)], dv1 = j < 4 ? da[2 * j + 1] : db[2 * (j - 4) + 1];
            const float ov0 = j < 4 ? oa[2 * j] : ob[2 * (j - 4)], ov1 = j < 4 ? oa[2 * j + 1] : ob[2 * (j - 4) + 1];
            gl[j] = ov0 > 0.f ? dv0 : 0.f;
            gr[j] = ov1 > 0.f ? dv1 : 0.f;
            el[j] = V8<T>::round(gl[j] * dl[j]);
            er[j] = V8<T>::round(gr[j] * dr[j]);
          }
          V8<T>::store(GS + p * g.ldgs + j0, gl);
          V8<T>::store(GS + p * g.ldgs + H + j0, gr);
          if (E) { V8<T>::store(E + p * g.lde + j0, el); V8<T>::store(E + p * g.lde + H + j0, er); }
          if (stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              s0l[j] += (A)el[j]; s1l[j] += (A)el[j] * (A)(a[j] - ml[j]);
              s0r[j] += (A)er[j]; s1r[j] += (A)er[j] * (A)(c[j] - mr[j]);
            }
          }
        }
      }
    }
  }
  if (!stats) return;
  A* red = reinterpret_cast<A*>(smem);                 // [NPL][4][H]
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(pl * 4 + 0) * H + j0 + j] = s0l[j]; red[(pl * 4 + 1) * H + j0 + j] = s1l[j];
      red[(pl * 4 + 2) * H + j0 + j] = s0r[j]; red[(pl * 4 + 3) * H + j0 + j] = s1r[j];
    }
  }
  __syncthreads();
  for (int i = tid; i < 4 * H; i += NT) {
    const int which = i / H, c = i - which * H;
    double s = 0.0;
    for (int q = 0; q < NPL; ++q) s += (double)red[(q * 4 + which) * H + c];
    double* dst = (which < 2) ? g.stats_l : g.stats_r;
    const int col = (which & 1) * H + c;
    dst[(long)blockIdx.x * 2 * H + col] = s;
    for (int r = blockIdx.x + gridDim.x; r < TSS_STAT_SLABS; r += gridDim.x) dst[(long)r * 2 * H + col] = 0.0;      // rows nobody owns
  }
}

}  // namespace

extern "C" {

/* out = channel_shuffle(relu(x + m * cat([bn_l(left), bn_r(right)])), 2): the tail of SSnbtBlock, TSS/models/lednet.py:112-124.
 * left / right: the branches' raw convolution outputs [P][C/2]; (mean, scale, shift): their BatchNorms as applied on load (scale NULL: none);
 * m: [B][C] dropout multipliers or NULL. */
int tss_ssnbt_tail_fwd(const void* left, long ldl, const float* mean_l, const float* scale_l, const float* shift_l,
                       const void* right, long ldr, const float* mean_r, const float* scale_r, const float* shift_r,
                       const void* x, long ldx, const float* m, void* out, long ldo, int B, long HW, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 16) == 0 && C <= 512 && (ldl % 8) == 0 && ldl >= C / 2 && (ldr % 8) == 0 && ldr >= C / 2 && (ldx % 8) == 0 && ldx >= C &&
              (ldo % 8) == 0 && ldo >= C && left && right && x && out && B > 0 && HW > 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(left) && tss::aligned16(right) && tss::aligned16(x) && tss::aligned16(out), TSS_ERR_ALIGN);
  TailArgs g = {};
  g.l = left; g.ldl = ldl; g.ml = mean_l; g.sl = scale_l; g.bl = shift_l;
  g.r = right; g.ldr = ldr; g.mr = mean_r; g.sr = scale_r; g.br = shift_r;
  g.x = x; g.ldx = ldx; g.m = m; g.out = out; g.ldo = ldo; g.HW = HW; g.B = B; g.C = C;
  const int npl = NT / (C / 16);
  long grid = (HW + npl - 1) / npl;
  if (grid > 2048) grid = 2048;
  tss::ProfScope prof(TSS_K_JOIN_FWD, (hipStream_t)stream, 3.0 * B * HW * C * (dtype == TSS_BF16 ? 2 : 4), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ssnbt_tail_fwd_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(ssnbt_tail_fwd_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, g);
  return tss::check_last("ssnbt_tail_fwd");
}

/* gs = unshuffle(d(out) * [out > 0]) (gradient of the skip), e = gs * m (gradient of the BatchNorm outputs, [P][C] = left | right; NULL when
 * m is NULL: e == gs), stats_l / stats_r: BatchNorm-backward slab rows of the two branches (sum e, sum e (raw - mean)) or NULL (frozen). */
int tss_ssnbt_tail_bwd(const void* dout, long lddo, const void* out, long ldo,
                       const void* left, long ldl, const float* mean_l, const void* right, long ldr, const float* mean_r,
                       const float* m, void* e, long lde, void* gs, long ldgs, double* stats_l, double* stats_r,
                       int B, long HW, int C, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (C % 16) == 0 && C <= 512 && (lddo % 8) == 0 && lddo >= C && (ldo % 8) == 0 && ldo >= C && (ldgs % 8) == 0 && ldgs >= C &&
              dout && out && gs && B > 0 && HW > 0 && (!e || ((lde % 8) == 0 && lde >= C)) && ((stats_l != nullptr) == (stats_r != nullptr)), TSS_ERR_SHAPE);
  TSS_REQUIRE(!stats_l || (left && right && mean_l && mean_r && (ldl % 8) == 0 && ldl >= C / 2 && (ldr % 8) == 0 && ldr >= C / 2), TSS_ERR_SHAPE);
  TSS_REQUIRE((m != nullptr) == (e != nullptr), TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(dout) && tss::aligned16(out) && tss::aligned16(gs) && (!e || tss::aligned16(e)), TSS_ERR_ALIGN);
  TailArgs g = {};
  g.dout = dout; g.lddo = lddo; g.out = const_cast<void*>(out); g.ldo = ldo;
  g.l = left; g.ldl = ldl; g.ml = mean_l; g.r = right; g.ldr = ldr; g.mr = mean_r;
  g.m = m; g.e = e; g.lde = lde; g.gs = gs; g.ldgs = ldgs; g.stats_l = stats_l; g.stats_r = stats_r; g.HW = HW; g.B = B; g.C = C;
  const int npl = NT / (C / 16);
  long grid = (HW + npl - 1) / npl;
  if (grid > TSS_STAT_SLABS) grid = TSS_STAT_SLABS;    // one statistics slab row per block
  const int nt = 2 + 1 + (e ? 1 : 0) + (stats_l ? 1 : 0);
  tss::ProfScope prof(TSS_K_JOIN_BWD, (hipStream_t)stream, (double)nt * B * HW * C * (dtype == TSS_BF16 ? 2 : 4), 0);
  if (dtype == TSS_BF16) hipLaunchKernelGGL(ssnbt_tail_bwd_kernel<bf16_t>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(ssnbt_tail_bwd_kernel<float>, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, g);
  return tss::check_last("ssnbt_tail_bwd");
}

}  // extern "C"

// Online hard example mining cross-entropy (TSS/losses/ohem_loss.py:10-21), the loss of the reference's training recipe:
//   l = per-pixel CE (0 for ignored pixels), n = int(numel * frac), v = the (n+1)-th largest l;
//   v > thresh ?  mean(l[l > thresh])  :  mean(the n largest l).
// The reference sorts all 16.8 M losses (and reads one element back on the host); here v is found by a 3-pass radix
// select on the float bits (losses are >= 0, so the bit pattern is monotonic): 11 + 11 + 10 bit histograms in LDS, a
// one-block scan between the passes, everything on the device (graph-capturable, no sort, no host round trip).
// Ties at v share the remaining weight evenly (the reference takes an arbitrary subset of them; the loss value is the
// same, and exact ties of positive losses do not occur in practice).
#include "common.h"

namespace {

constexpr int NT = 256;

struct OhemState {            // device workspace, zero-initialised by the caller
  unsigned int hist[2048];
  unsigned int prefix;        // known upper bits of v
  unsigned int pad;
  long long remaining;        // rank of v among the elements that share the prefix (0-based, descending)
  double acc[5];              // sum(l > thresh), count(l > thresh), sum(l > v), count(l > v), count(l == v)
};

template <typename T>
__global__ __launch_bounds__(NT) void ohem_pixel_kernel(const T* logits, const long long* target, float* lse_out,
                                                        float* pix, long B, int C, long HW, int ignore_index) {
  const long groups = B * (HW / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    const T* base = logits + b * C * HW + off;
    float m[8], s[8], lt[8];
    long long t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; s[j] = 0.f; lt[j] = 0.f; t[j] = target[b * HW + off + j]; }
    for (int c = 0; c < C; ++c) {
      float v[8];
      V8<T>::load(base + (long)c * HW, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mn = fmaxf(m[j], v[j]);
        s[j] = s[j] * __expf(m[j] - mn) + __expf(v[j] - mn);
        m[j] = mn;
        if (t[j] == c) lt[j] = v[j];
      }
    }
    float l[8], p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = m[j] + __logf(s[j]);
      const float d = l[j] - lt[j];
      p[j] = (t[j] != ignore_index && t[j] >= 0 && t[j] < C) ? fmaxf(d, 0.f) : 0.f;     // a CE value is >= 0 (rounding may give -1e-7)
    }
    V8<float>::store(lse_out + b * HW + off, l);
    V8<float>::store(pix + b * HW + off, p);
  }
}

__device__ __forceinline__ unsigned int himask(int pass) { return pass == 0 ? 0u : (pass == 1 ? 0xFFE00000u : 0xFFFFFC00u); }
__device__ __forceinline__ int shift_of(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __forceinline__ int bins_of(int pass) { return pass == 2 ? 1024 : 2048; }

__global__ __launch_bounds__(NT) void ohem_hist_kernel(const float* pix, long n, OhemState* st, int pass) {
  __shared__ unsigned int h[2048];
  for (int i = threadIdx.x; i < 2048; i += NT) h[i] = 0u;
  __syncthreads();
  const unsigned int hm = himask(pass), prefix = st->prefix & hm;
  const int sh = shift_of(pass);
  const unsigned int bm = (unsigned int)bins_of(pass) - 1u;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint4 k = reinterpret_cast<const uint4*>(pix)[i];
    if ((k.x & hm) == prefix) atomicAdd(&h[(k.x >> sh) & bm], 1u);
    if ((k.y & hm) == prefix) atomicAdd(&h[(k.y >> sh) & bm], 1u);
    if ((k.z & hm) == prefix) atomicAdd(&h[(k.z >> sh) & bm], 1u);
    if ((k.w & hm) == prefix) atomicAdd(&h[(k.w >> sh) & bm], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += NT)
    if (h[i]) atomicAdd(&st->hist[i], h[i]);
}

// one block: finds the bin (from the top) that holds the element of rank `remaining`, extends the prefix, clears hist
__global__ __launch_bounds__(NT) void ohem_scan_kernel(OhemState* st, int pass, long long first_rank) {
  __shared__ unsigned int part[NT];
  __shared__ int sel_group;
  __shared__ long long before_group;
  const int tid = threadIdx.x;
  const int nb = bins_of(pass), per = nb / NT;        // 8 or 4 bins per thread, thread 0 owns the HIGHEST bins
  if (pass == 0 && tid == 0) st->remaining = first_rank;
  unsigned int mine = 0u;
  for (int q = 0; q < per; ++q) mine += st->hist[nb - 1 - (tid * per + q)];
  part[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    const long long r = st->remaining;
    long long cum = 0;
    int gsel = NT - 1;
    long long before = 0;
    for (int gidx = 0; gidx < NT; ++gidx) {
      if (cum + part[gidx] > r) { gsel = gidx; before = cum; break; }
      cum += part[gidx];
      before = cum;
    }
    sel_group = gsel; before_group = before;
  }
  __syncthreads();
  if (tid == 0) {
    const long long r = st->remaining;
    long long cum = before_group;
    int bsel = nb - 1 - (sel_group * per + per - 1);
    for (int q = 0; q < per; ++q) {
      const int bin = nb - 1 - (sel_group * per + q);
      const unsigned int hcount = st->hist[bin];
      if (cum + hcount > r) { bsel = bin; break; }
      cum += hcount;
    }
    st->prefix = (st->prefix & himask(pass)) | ((unsigned int)bsel << shift_of(pass));
    st->remaining = r - cum;
  }
  __syncthreads();
  for (int i = tid; i < 2048; i += NT) st->hist[i] = 0u;
}

__global__ __launch_bounds__(NT) void ohem_sum_kernel(const float* pix, long n, OhemState* st, float thresh) {
  __shared__ double red[5][NT / 64];
  const float v = __uint_as_float(st->prefix);
  double a[5] = {0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float l = pix[i];
    if (l > thresh) { a[0] += l; a[1] += 1.0; }
    if (l > v) { a[2] += l; a[3] += 1.0; }
    else if (l == v) a[4] += 1.0;
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const double s = wave_sum(a[q]);
    if ((threadIdx.x & 63) == 0) red[q][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double s = 0.0;
    for (int wv = 0; wv < NT / 64; ++wv) s += red[threadIdx.x][wv];
    atomicAdd(&st->acc[threadIdx.x], s);
  }
}

// params: [mode (1: threshold, 0: top-n), cut value, weight of l > cut, weight of l == cut]
__global__ void ohem_finalize_kernel(OhemState* st, float thresh, long long n_top, float* loss, float* params) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float v = __uint_as_float(st->prefix);
  if (v > thresh) {
    *loss = (float)(st->acc[0] / st->acc[1]);
    params[0] = 1.f; params[1] = thresh; params[2] = (float)(1.0 / st->acc[1]); params[3] = 0.f;
  } else {
    const double n = (double)n_top, ties = n - st->acc[3];
    *loss = (float)((st->acc[2] + ties * (double)v) / n);           // n == 0 -> nan, like torch's empty mean
    params[0] = 0.f; params[1] = v; params[2] = (float)(1.0 / n);
    params[3] = st->acc[4] > 0.0 ? (float)(ties / (st->acc[4] * n)) : 0.f;
  }
  for (int q = 0; q < 5; ++q) st->acc[q] = 0.0;                     // the workspace is reusable without clearing
  st->prefix = 0u; st->remaining = 0;
}

template <typename T>
__global__ __launch_bounds__(NT) void ohem_bwd_kernel(const T* logits, const long long* target, const float* lse,
                                                      const float* pix, const float* params, const float* grad_out,
                                                      T* dlogits, long B, int C, long HW, int ignore_index) {
  const long groups = B * (HW / 8);
  const float cut = params[1], wgt = params[2], weq = params[0] != 0.f ? 0.f : params[3];
  const float gs = grad_out ? *grad_out : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (HW / 8);
    const long off = (i - b * (HW / 8)) * 8;
    float l[8], p[8], w[8];
    long long t[8];
    V8<float>::load(lse + b * HW + off, l);
    V8<float>::load(pix + b * HW + off, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      t[j] = target[b * HW + off + j];
      const float sel = p[j] > cut ? wgt : (p[j] == cut ? weq : 0.f);
      w[j] = (t[j] != ignore_index && t[j] >= 0 && t[j] < C) ? sel * gs : 0.f;
    }
    for (int c = 0; c < C; ++c) {
      float v[8], d[8];
      V8<T>::load(logits + (b * C + c) * HW + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = (__expf(v[j] - l[j]) - (t[j] == c ? 1.f : 0.f)) * w[j];
      V8<T>::store(dlogits + (b * C + c) * HW + off, d);
    }
  }
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

long tss_ohem_workspace_bytes(void) { return (long)sizeof(OhemState); }

int tss_ohem_fwd(const void* logits, const long long* target, float* lse, float* pixel_loss, void* workspace,
                 float* loss, float* params /*[4]*/, long B, int C, long HW, int ignore_index, float thresh_loss,
                 long n_top, int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0 && n_top >= 0 && n_top < B * HW, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(lse) && tss::aligned16(pixel_loss) && tss::aligned16(workspace), TSS_ERR_ALIGN);
  const long n = B * HW;
  if (n == 0) return TSS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(ohem_pixel_kernel<bf16_t>, dim3(grid_for(n / 8)), dim3(NT), 0, st, (const bf16_t*)logits, target, lse, pixel_loss, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(ohem_pixel_kernel<float>, dim3(grid_for(n / 8)), dim3(NT), 0, st, (const float*)logits, target, lse, pixel_loss, B, C, HW, ignore_index);
  return tss_ohem_select(pixel_loss, workspace, loss, params, n, thresh_loss, n_top, stream);
}

// The selection of TSS/losses/ohem_loss.py:13-21 on an array of n per-pixel losses (>= 0; n % 4 == 0): loss value + the
// parameters [mode, cut, weight of l > cut, weight of l == cut] the backward kernels turn into per-pixel weights.
int tss_ohem_select(const float* pixel_loss, void* workspace, float* loss, float* params, long n, float thresh_loss, long n_top,
                    void* stream) {
  TSS_REQUIRE(n > 0 && (n % 4) == 0 && n_top >= 0 && n_top < n && pixel_loss && workspace && loss && params, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(pixel_loss) && tss::aligned16(workspace), TSS_ERR_ALIGN);
  hipStream_t st = (hipStream_t)stream;
  OhemState* ws = reinterpret_cast<OhemState*>(workspace);
  const int hgrid = grid_for(n / 4) > 1024 ? 1024 : grid_for(n / 4);
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(hgrid), dim3(NT), 0, st, pixel_loss, n, ws, pass);
    hipLaunchKernelGGL(ohem_scan_kernel, dim3(1), dim3(NT), 0, st, ws, pass, (long long)n_top);
  }
  hipLaunchKernelGGL(ohem_sum_kernel, dim3(hgrid), dim3(NT), 0, st, pixel_loss, n, ws, thresh_loss);
  hipLaunchKernelGGL(ohem_finalize_kernel, dim3(1), dim3(64), 0, st, ws, thresh_loss, (long long)n_top, loss, params);
  return tss::check_last("ohem_select");
}

int tss_ohem_bwd(const void* logits, const long long* target, const float* lse, const float* pixel_loss,
                 const float* params, const float* grad_out, void* dlogits, long B, int C, long HW, int ignore_index,
                 int dtype, void* stream) {
  TSS_REQUIRE(dtype == TSS_F32 || dtype == TSS_BF16, TSS_ERR_DTYPE);
  TSS_REQUIRE(C > 0 && (HW % 8) == 0, TSS_ERR_SHAPE);
  TSS_REQUIRE(tss::aligned16(logits) && tss::aligned16(dlogits) && tss::aligned16(lse) && tss::aligned16(pixel_loss), TSS_ERR_ALIGN);
  const long groups = B * (HW / 8);
  if (groups == 0) return TSS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSS_BF16)
    hipLaunchKernelGGL(ohem_bwd_kernel<bf16_t>, dim3(grid_for(groups)), dim3(NT), 0, st, (const bf16_t*)logits, target, lse, pixel_loss, params, grad_out, (bf16_t*)dlogits, B, C, HW, ignore_index);
  else
    hipLaunchKernelGGL(ohem_bwd_kernel<float>, dim3(grid_for(groups)), dim3(NT), 0, st, (const float*)logits, target, lse, pixel_loss, params, grad_out, (float*)dlogits, B, C, HW, ignore_index);
  return tss::check_last("ohem_bwd");
}

}  // extern "C"

// Host-side support: HIP error mapping and the optional per-launch profiler.
// The profiler brackets every kernel launch with two HIP events recorded on the launch stream, so the
// durations it reports are device times of exactly the kernels bench.py attributes bytes to.
#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct KernelInfo { const char* name; const char* symbol; };
const KernelInfo kInfo[TSS_K_COUNT] = {
    // symbol = the kernels rocprofv3 lists for the entry point on the bf16 path (rocprofv3 prints the bool template
    // argument: <false> forward, <true> backward-data), then the general kernel the f32 / ragged shapes fall back to
    {"pwconv_fwd", "pwfast_kernel<false>|pwfast_mc_kernel<false>|convgemm_kernel"},
    // (pwconv_bwd_data also carries the one-sweep backward kernels, pwbwd_kernel / pwsweep_kernel: input gradient + weight gradient)
    {"pwconv_bwd_data", "pwfast_kernel<true>|pwfast_mc_kernel<true>|pwbwd_kernel|pwsweep_kernel|convgemm_kernel"},
    {"pwconv_bwd_weight", "wgfast_kernel|wgrad_kernel"},
    {"conv3x3_fwd", "conv3x3_wstat_kernel|conv3x3_stream_kernel|conv3x3_lean_kernel<false>|fc1d_kernel|convgemm_kernel<fwd>"}, {"conv3x3_bwd_data", "conv3x3_lean_kernel<true>|fc1d_kernel|convgemm_kernel<bwd>"}, {"conv3x3_bwd_weight", "fc1d_wgrad_kernel|wgrad_kernel"},
    {"stem3x3_fwd", "stem_fwd_mfma_kernel|convgemm_kernel"}, {"stem3x3_bwd_weight", "stem_wgrad_mfma_kernel|stem_wgrad_kernel"},
    // (dwconv3x3_bwd_data also carries the one-sweep backward, dw_bwd_roll_s{1,2}_kernel: input gradient + weight gradient)
    {"dwconv3x3_fwd", "dw_fwd_roll_kernel|dw_fwd_strip_kernel"}, {"dwconv3x3_bwd_data", "dw_bwd_roll_s1_kernel|dw_bwd_roll_s2_kernel|dw_bwd_data_strip_kernel"},
    {"dwconv3x3_bwd_weight", "dw_bwd_weight_strip_kernel"},
    {"bn_finalize", "bn_finalize_kernel"}, {"bn_bwd_finalize", "bn_bwd_finalize_kernel"},
    {"join_fwd", "join_fwd_kernel"}, {"join_bwd", "join_bwd_kernel"},
    {"dropout", "dropout_kernel"}, {"bias_grad", "colsum_kernel"}, {"adamw", "adamw_kernel"},
    {"bilinear_nhwc_fwd", "bilinear_nhwc_fwd_kernel|ppm_concat_fwd_kernel"}, {"bilinear_nhwc_bwd_rows", "bilinear_nhwc_bwd_rows_kernel|ppm_concat_bwd_kernel"},
    {"bilinear_nhwc_bwd_cols", "bilinear_nhwc_bwd_cols_kernel"},
    {"bilinear_planar_fwd", "bilinear_planar_fwd_kernel"},
    {"upsample_head_fwd", "upsample_head_fwd_kernel"}, {"upsample_head_bwd_rows", "upsample_head_bwd_rows_kernel"},
    {"upsample_head_bwd_cols", "upsample_head_bwd_cols_kernel"},
    {"adaptive_pool_fwd", "adaptive_pool_fwd_kernel|ppm_pool_fwd_kernel"}, {"adaptive_pool_bwd", "adaptive_pool_bwd_kernel|ppm_pool_bwd_kernel"},
    {"copy_nhwc", "copy_nhwc_kernel"},
    {"cross_entropy_fwd", "ce_fwd_kernel"}, {"cross_entropy_bwd", "ce_bwd_kernel"}, {"argmax_confusion", "argmax_confusion_kernel"},
    {"upsample_ce_fwd", "upsample_ce_onepass_kernel"}, {"upsample_ce_bwd", "upsample_ce_gather_kernel"},
};

struct Rec { int kid; hipEvent_t a, b; double bytes, flops; };
struct Total { long launches; double ms, bytes, flops; };
struct Done { int kid; double ms, bytes; };
std::vector<Done> g_done;

std::mutex g_mu;
bool g_enabled = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
Total g_tot[TSS_K_COUNT];
thread_local char g_err[256] = "";

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

}  // namespace

namespace tss {

int check_last(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err == hipSuccess) return TSS_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
  return TSS_ERR_HIP;
}

ProfScope::ProfScope(int kernel_id, hipStream_t s, double alg_bytes, double flops) : slot(-1), stream(s) {
  if (!g_enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.kid = kernel_id; r.bytes = alg_bytes; r.flops = flops;
  r.a = get_event(); r.b = get_event();
  (void)hipEventRecord(r.a, stream);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[slot].b, stream);
}

}  // namespace tss

extern "C" {

int tss_version(void) { return 1; }

// hipMemsetAsync on the launch stream (a MEMSET node when the stream is being captured).  Only used by tools/graph_memset_probe.py
// and the engine's TSS_MEMSET_NODES=1 diagnostic: the product clears its buffers from kernels (see tss_cast_weights).
int tss_memset_zero(void* p, long bytes, void* stream) {
  if (!p || bytes <= 0) return TSS_OK;
  if (hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) return TSS_ERR_HIP;
  return tss::check_last("memset_zero");
}
const char* tss_last_error(void) { return g_err; }
const char* tss_arch(void) { return "gfx950"; }

int tss_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on != 0;
  return TSS_OK;
}

int tss_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (Rec& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  for (int i = 0; i < TSS_K_COUNT; ++i) g_tot[i] = Total{0, 0.0, 0.0, 0.0};
  g_done.clear();
  return TSS_OK;
}

int tss_prof_collect(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (Rec& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return TSS_ERR_HIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return TSS_ERR_HIP;
    Total& t = g_tot[r.kid];
    t.launches += 1; t.ms += ms; t.bytes += r.bytes; t.flops += r.flops;
    g_done.push_back(Done{r.kid, (double)ms, r.bytes});
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  return TSS_OK;
}

int tss_prof_get(int kernel_id, long* launches, double* total_ms, double* alg_bytes, double* flops) {
  if (kernel_id < 0 || kernel_id >= TSS_K_COUNT) return TSS_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(g_mu);
  const Total& t = g_tot[kernel_id];
  if (launches) *launches = t.launches;
  if (total_ms) *total_ms = t.ms;
  if (alg_bytes) *alg_bytes = t.bytes;
  if (flops) *flops = t.flops;
  return TSS_OK;
}

long tss_prof_records(int* kernel_ids, double* ms, double* alg_bytes, long max_records) {
  std::lock_guard<std::mutex> lk(g_mu);
  long n = (long)g_done.size() < max_records ? (long)g_done.size() : max_records;
  for (long i = 0; i < n; ++i) { kernel_ids[i] = g_done[i].kid; ms[i] = g_done[i].ms; alg_bytes[i] = g_done[i].bytes; }
  return (long)g_done.size();
}

const char* tss_prof_name(int kernel_id) {
  return (kernel_id >= 0 && kernel_id < TSS_K_COUNT) ? kInfo[kernel_id].name : "";
}
const char* tss_prof_symbol(int kernel_id) {
  return (kernel_id >= 0 && kernel_id < TSS_K_COUNT) ? kInfo[kernel_id].symbol : "";
}

}  // extern "C"

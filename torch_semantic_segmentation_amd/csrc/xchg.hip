// Cross-replica (Sync) BatchNorm without a collective library in the loop (round 4).
//
// apex.parallel.convert_syncbn_model (TSS scripts/train_fastscnn.py:144-145) makes every BatchNorm all-reduce a [2C + 1] vector in
// forward and another in backward: 88 latency-bound collectives per FastSCNN step.  Through RCCL that was slab_reduce + all_reduce +
// finalize per layer and direction (+0.61 ms per step with ONE rank).  Here the exchange is part of the finalize kernel itself:
//
//   * every rank owns a MAILBOX in fine-grained device memory, mapped into every peer of the node through HIP IPC
//     (hipIpcGetMemHandle / hipIpcOpenMemHandle; tss_ipc_*);
//   * block b of the finalize grid reduces the slab rows of ITS 8 channels, writes the 16 sums (+ the element count) into its cell of
//     EVERY rank's mailbox -- system-scope stores over xGMI, then a system-scope release fence, then the cell's flag = its call counter --
//     polls the flags of the peers' cells in its OWN mailbox (bounded), and adds the cells up in rank order: every rank computes the
//     same global sums bit for bit, with no reduction tree and no second launch;
//   * the cells are 4 slots deep (slot = call counter mod 4): a rank can run at most one exchange ahead of a peer that has not read
//     the previous one yet, because completing an exchange needs the peer's flag of that exchange;
//   * the call counter lives on the device, one word per block index, so a replayed HIP graph advances it by itself.
// One rank: the kernel degenerates to bn_finalize_kernel plus a few stores to itself.  More than one GPU has NOT been run on hardware
// available to this repository: the two-rank test shares one GPU between two processes (the IPC mapping, the protocol and the
// arithmetic are the real ones; xGMI latency and cross-device visibility are not exercised).
#include <cstring>

#include "bnfin.h"
#include "common.h"

namespace {

using namespace tss_fin;      // Xchg, exchange(), cell_off() live in bnfin.h: the riding finalize blocks of wgrad.hip use them too

__global__ __launch_bounds__(FIN_NT) void bn_finalize_xchg_kernel(const double* sums, double count, const Xchg x, const float* gamma,
                                                                  float eps, float momentum, float* running_mean, float* running_var,
                                                                  long long* num_batches, float* mean_out, float* invstd_out,
                                                                  float* scale, int C) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  const int cp = min(blockIdx.x * FIN_CH + (int)(threadIdx.x & (FIN_CH - 1)), C - 1);
  const float g_in = (gamma ? gamma : mean_out)[cp];
  const float rm_in = (running_mean ? running_mean : mean_out)[cp];
  const float rv_in = (running_var ? running_var : mean_out)[cp];
  double ssum, ssq, cnt = count;
  int c;
  slab_sum(sums, C, blockIdx.x, &ssum, &ssq, &c);
  exchange(x, blockIdx.x, ssum, ssq, cnt);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double mean = ssum / cnt;
  double var = ssq / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? g_in : 1.f;
  const float m = (float)mean;
  mean_out[c] = m;
  invstd_out[c] = invstd;
  scale[c] = g * invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * rm_in + momentum * m;
  if (running_var) {
    const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_var[c] = (1.f - momentum) * rv_in + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_xchg_kernel(const tss_bn_bwd_job j) { bn_bwd_finalize_block(j, blockIdx.x); }

bool fill(Xchg& x, const void* const* peers, int rank, int world, unsigned long long* ctr) {
  if (!peers || !ctr || world < 1 || world > XMAXW || rank < 0 || rank >= world) return false;
  for (int p = 0; p < world; ++p) {
    if (!peers[p]) return false;
    x.peers[p] = (unsigned long long*)peers[p];
  }
  x.rank = rank; x.world = world; x.ctr = ctr;
  return true;
}

}  // namespace

extern "C" {

long tss_bn_xchg_bytes(void) { return (XHEAD + (long)XSLOTS * XMAXW * XMAXB * XCELL) * 8; }
int tss_bn_xchg_counters(void) { return XMAXB; }

// A zero-filled mailbox in fine-grained device memory + its IPC handle (64 bytes).  The pointer is NOT a torch allocation: free
// it with tss_ipc_free.  tss_ipc_open maps a peer's handle into this process (tss_ipc_close unmaps).
int tss_ipc_alloc(long bytes, void** ptr_out, void* handle_out) {
  TSS_REQUIRE(bytes > 0 && ptr_out && handle_out, TSS_ERR_SHAPE);
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc(&p, (size_t)bytes) != hipSuccess) return tss::check_last("ipc_alloc");
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return tss::check_last("ipc_alloc"); }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return tss::check_last("ipc_get_handle"); }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle_out, &h, 64);
  *ptr_out = p;
  return TSS_OK;
}

int tss_ipc_open(const void* handle, void** ptr_out) {
  TSS_REQUIRE(handle && ptr_out, TSS_ERR_SHAPE);
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return tss::check_last("ipc_open");
  *ptr_out = p;
  return TSS_OK;
}

int tss_ipc_close(void* ptr) { return (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) ? tss::check_last("ipc_close") : TSS_OK; }
int tss_ipc_free(void* ptr) { return (ptr && hipFree(ptr) != hipSuccess) ? tss::check_last("ipc_free") : TSS_OK; }

// error word of this rank's mailbox (host copy): 0, or 1 when an exchange gave up waiting for a peer
int tss_bn_xchg_error(const void* mailbox, long* out) {
  TSS_REQUIRE(mailbox && out, TSS_ERR_SHAPE);
  unsigned long long v = 0;
  if (hipMemcpy(&v, mailbox, 8, hipMemcpyDeviceToHost) != hipSuccess) return tss::check_last("bn_xchg_error");
  *out = (long)v;
  return TSS_OK;
}

// tss_bn_finalize with the statistics summed over the ranks of one node inside the kernel (see the top of this file).
// peers: HOST array of `world` mailbox pointers as mapped into this process (peers[rank] = this rank's own mailbox);
// counters: tss_bn_xchg_counters() zero-initialised 64-bit words of ordinary device memory, private to this rank.
int tss_bn_finalize_xchg(const double* sums, double count, const void* const* peers, int rank, int world, void* counters,
                         const float* gamma, float eps, float momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* mean_out, float* invstd_out, float* scale, int C, void* stream) {
  TSS_REQUIRE(C > 0 && C <= XMAXB * FIN_CH && count >= 1.0 && sums && mean_out && invstd_out && scale, TSS_ERR_SHAPE);
  Xchg x = {};
  TSS_REQUIRE(fill(x, peers, rank, world, (unsigned long long*)counters), TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_FINALIZE, (hipStream_t)stream, 40.0 * C, 0);
  hipLaunchKernelGGL(bn_finalize_xchg_kernel, dim3(fin_blocks(C)), dim3(FIN_NT), 0, (hipStream_t)stream, sums, count, x, gamma, eps,
                     momentum, running_mean, running_var, num_batches_tracked, mean_out, invstd_out, scale, C);
  return tss::check_last("bn_finalize_xchg");
}

int tss_bn_bwd_finalize_xchg(const double* bstats, double count, const void* const* peers, int rank, int world, void* counters,
                             const float* invstd, const float* gamma, int accumulate, float* dgamma, float* dbeta,
                             float* ga, float* gb, float* gce, int C, void* stream) {
  TSS_REQUIRE(C > 0 && C <= XMAXB * FIN_CH && count >= 1.0 && bstats && invstd && ga && gb && gce, TSS_ERR_SHAPE);
  Xchg x = {};
  TSS_REQUIRE(fill(x, peers, rank, world, (unsigned long long*)counters), TSS_ERR_SHAPE);
  tss::ProfScope prof(TSS_K_BN_BWD_FINALIZE, (hipStream_t)stream, 48.0 * C, 0);
  tss_bn_bwd_job j = {bstats, count, invstd, gamma, 1, accumulate, dgamma, dbeta, ga, gb, gce, C};
  j.xchg_world = world; j.xchg_rank = rank; j.xchg_counters = counters;
  for (int p = 0; p < world; ++p) j.xchg_peers[p] = (void*)peers[p];
  hipLaunchKernelGGL(bn_bwd_finalize_xchg_kernel, dim3(fin_blocks(C)), dim3(FIN_NT), 0, (hipStream_t)stream, j);
  return tss::check_last("bn_bwd_finalize_xchg");
}

}  // extern "C"

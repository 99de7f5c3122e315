/* libtss_hip.so -- C ABI of the MI355X (gfx950) hot path of torch_semantic_segmentation.
 *
 * The reference (bernardomig/torch_semantic_segmentation) has no native code: its FastSCNN / ContextNet
 * convolution stacks run through torch.nn leaf modules -> ATen -> MIOpen/cuDNN.  Each entry point below
 * replaces the ATen dispatch made by one of those leaf-module call sites; the citation after "replaces:"
 * is the reference interface (file:line under torch_semantic_segmentation/, "TSS/").
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (the library never
 *     allocates, frees or retains tensor memory), `stream` is a hipStream_t passed as void*.
 *   - activations are NHWC ("channels_last"): a tensor is P = B*H*W pixel rows of C channels, row pitch
 *     `ld*` in ELEMENTS (>= C, multiple of 8) so channel slices of a wider buffer can be addressed.
 *   - dtype: TSS_F32 (parity path, exact-f32 MFMA) or TSS_BF16 (performance path, f32 accumulate).
 *     Parameters, statistics and gradients of parameters are always f32 (statistics: f64 sums).
 *   - deferred BatchNorm: a conv output is stored RAW; the BatchNorm(+ReLU) that follows it is applied by
 *     the CONSUMER on load: a = relu?((x - in_mean) * in_scale + in_bias) with in_scale = gamma/sqrt(var+eps),
 *     in_bias = beta (in_scale NULL: identity; in_mean / in_bias NULL: 0).  In backward, the tensor `e` is
 *     d(loss)/d(BN output) (already ReLU-masked) and the BN backward g = ga*(e - gce) + gb*(y_raw - gmu) is
 *     applied on load from per-channel coefficients produced by tss_bn_bwd_finalize (gmu = batch mean;
 *     yraw NULL: g = ga*e, frozen statistics; ga NULL too: g = e).  Both forms subtract the mean BEFORE
 *     scaling, as torch does, so channels with |mean| >> std lose no precision.
 *   - every function returns TSS_OK or a negative TSS_ERR_* code; launches are asynchronous on `stream`,
 *     there is no internal synchronisation and no global mutable state besides the optional profiler.
 */
#ifndef TSS_HIP_H
#define TSS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TSS_OK 0
#define TSS_ERR_DTYPE (-1)  /* unknown dtype code */
#define TSS_ERR_SHAPE (-2)  /* unsupported shape / pitch (e.g. channels not a multiple of 8) */
#define TSS_ERR_ALIGN (-3)  /* pointer not 16-byte aligned */
#define TSS_ERR_HIP (-4)    /* hipGetLastError() != hipSuccess after the launch; see tss_last_error() */

#define TSS_F32 0
#define TSS_BF16 1

/* The arguments of tss_bn_bwd_finalize as a plain struct: a finalize that rides in front of another launch
 * (tss_pwconv_bwd_weight's `fin`) is described by one of these, filled in by the caller on the host. */
typedef struct tss_bn_bwd_job {
  const double* bstats; double count; const float* invstd; const float* gamma; int training; int accumulate;
  float* dgamma; float* dbeta; float* ga; float* gb; float* gce; int C;
  /* cross-replica statistics (round 4, csrc/xchg.hip): xchg_world > 0 -- the sums cross the ranks inside the finalize blocks, as
   * tss_bn_bwd_finalize_xchg does (peers: the world's mailboxes as mapped into this process, rank-indexed; counters: this rank's
   * call counters); xchg_world == 0: this replica's own statistics */
  int xchg_world; int xchg_rank; void* xchg_peers[8]; void* xchg_counters;
} tss_bn_bwd_job;

/* kernel ids for the profiler (tss_prof_*) */
enum {
  TSS_K_PWCONV_FWD = 0, TSS_K_PWCONV_BWD_DATA, TSS_K_PWCONV_BWD_WEIGHT,
  TSS_K_CONV3X3_FWD, TSS_K_CONV3X3_BWD_DATA, TSS_K_CONV3X3_BWD_WEIGHT,
  TSS_K_STEM_FWD, TSS_K_STEM_BWD_WEIGHT,
  TSS_K_DWCONV_FWD, TSS_K_DWCONV_BWD_DATA, TSS_K_DWCONV_BWD_WEIGHT,
  TSS_K_BN_FINALIZE, TSS_K_BN_BWD_FINALIZE, TSS_K_JOIN_FWD, TSS_K_JOIN_BWD,
  TSS_K_DROPOUT, TSS_K_BIAS_GRAD, TSS_K_ADAMW,
  TSS_K_BILINEAR_FWD, TSS_K_BILINEAR_BWD, TSS_K_BILINEAR_BWD_COLS, TSS_K_BILINEAR_PLANAR_FWD,
  TSS_K_UPSAMPLE_HEAD_FWD, TSS_K_UPSAMPLE_HEAD_BWD_ROWS, TSS_K_UPSAMPLE_HEAD_BWD_COLS,
  TSS_K_POOL_FWD, TSS_K_POOL_BWD, TSS_K_COPY,
  TSS_K_CE_FWD, TSS_K_CE_BWD, TSS_K_ARGMAX, TSS_K_UPSAMPLE_CE_FWD, TSS_K_UPSAMPLE_CE_BWD,
  TSS_K_COUNT
};

/* ---- library / diagnostics --------------------------------------------------------------------------- */
int tss_version(void);                 /* ABI version of this header */
int tss_memset_zero(void* p, long bytes, void* stream);   /* diagnostic: hipMemsetAsync (a memset NODE under capture), see DESIGN.md section 4 */
const char* tss_last_error(void);      /* text of the last HIP error seen by this library (thread-local) */
const char* tss_arch(void);            /* "gfx950" */
#define TSS_OPT_DISABLE_FAST_PATHS 1   /* value 1: bf16 calls use the general kernels only (A/B checks of the lean ones) */
int tss_set_option(int key, int value);
int tss_get_option(int key);   /* current value, -1 for an unknown key */

/* ---- profiler: HIP events around every launch, on the launch stream --------------------------------- */
int tss_prof_enable(int on);           /* 1: record events for every launch from now on; 0: stop */
int tss_prof_reset(void);
int tss_prof_collect(void);            /* synchronises recorded events and folds them into the per-kernel table */
int tss_prof_get(int kernel_id, long* launches, double* total_ms, double* alg_bytes, double* flops);
long tss_prof_records(int* kernel_ids, double* ms, double* alg_bytes, long max_records);  /* per-launch, in launch order */
const char* tss_prof_name(int kernel_id);    /* operator name, e.g. "pwconv_fwd" */
const char* tss_prof_symbol(int kernel_id);  /* device kernel the operator launches, e.g. "convgemm_kernel" */

/* ---- pointwise (1x1) convolution --------------------------------------------------------------------
 * replaces: nn.Conv2d(k=1, bias=False) (+BatchNorm2d, ReLU fused as described above) built by
 *           Conv2dBlock TSS/models/fastscnn.py:164-173, DSConv2dBlock :194, ConvBlock TSS/models/contextnet.py:168-177
 *           and the biased classifier conv TSS/models/fastscnn.py:97, TSS/models/contextnet.py:86.
 * y[p][n] = sum_k act(x[p][k]) * w[n][k] (+ bias[n]);  stats (optional) = partial sums of y and y^2.
 * w_bf16 (optional, bf16 path): a current bf16 copy [N][K] of w written by tss_cast_weights -- staged with plain copies. */
int tss_pwconv_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                   const float* w, const void* w_bf16, const float* bias, void* y, long ldy, double* stats,
                   long P, int K, int N, int dtype, void* stream);
/* Eval mode (frozen statistics, no gradient): the BatchNorm BEHIND the layer, the skip of a residual block and the ReLU after the sum
 * are applied in the epilogue, so a block output is written by its last 1x1 layer and no join pass exists:
 *   y = relu?((x_act W^T + bias - out_mean) * out_scale + out_beta + residual)      out_scale = gamma / sqrt(running_var + eps)
 * replaces: conv3 -> BatchNorm2d -> (+ input) -> ReLU of BottleneckBlock.forward, TSS/models/fastscnn.py:152-161,
 *           TSS/models/contextnet.py:139-147 under model.eval().  bf16; K <= 768 (multiple of 8), N a multiple of 4. */
int tss_pwconv_fwd_joined(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          const float* w, const void* w_bf16, const float* bias,
                          const float* out_mean, const float* out_scale, const float* out_beta,
                          const void* residual, long ldr, int out_relu, void* y, long ldy,
                          long P, int K, int N, int dtype, void* stream);
/* A whole inverted residual with FROZEN statistics as ONE kernel (csrc/bneck.hip, round 4; bf16): 1x1 expand -> BatchNorm -> ReLU ->
 * depthwise 3x3 (stride 1 | 2, padding 1) -> BatchNorm -> ReLU -> 1x1 project -> BatchNorm (+ x when residual) -> ReLU.
 * replaces: BottleneckBlock.forward under model.eval(), TSS/models/fastscnn.py:152-161, TSS/models/contextnet.py:139-147 -- the two
 *           6x-expanded tensors live in LDS only.  x: materialised NHWC [B][H][W][Cin]; y: NHWC [B][Ho][Wo][Cout], Ho = (H-1)/stride + 1.
 * w1 [Cmid][Cin], wdw [Cmid][3][3], w3 [Cout][Cmid]: the f32 parameters; w1_bf16 / w3_bf16: optional current bf16 copies (tss_cast_weights).
 * (mean_i, scale_i, beta_i): the BatchNorm behind layer i as tss_bn_eval_affine / tss_bn_eval_affine_batched write it.
 * Cin <= 128 (multiple of 8), Cmid a multiple of 64, Cout <= 128 (multiple of 4); residual needs stride 1 and Cin == Cout. */
int tss_bneck_eval_supported(int Cin, int Cmid, int Cout, int stride, int residual, int dtype);
int tss_bneck_eval_fwd(const void* x, long ldx, const float* w1, const void* w1_bf16, const float* mean1, const float* scale1,
                       const float* beta1, const float* wdw, const float* mean2, const float* scale2, const float* beta2,
                       const float* w3, const void* w3_bf16, const float* mean3, const float* scale3, const float* beta3,
                       int residual, void* y, long ldy, int B, int H, int W, int Cin, int Cmid, int Cout, int stride, int dtype,
                       void* stream);
/* ---- cross-replica (Sync) BatchNorm with the exchange INSIDE the finalize kernel (csrc/xchg.hip, round 4) ----------------------
 * replaces: apex.parallel.convert_syncbn_model's two collectives per layer (TSS scripts/train_fastscnn.py:144-145) for the ranks of ONE
 * node: every rank owns a mailbox in fine-grained device memory (tss_ipc_alloc, tss_bn_xchg_bytes() bytes), maps the mailboxes of its
 * peers through HIP IPC (tss_ipc_open on their 64-byte handles) and passes the world's pointers, rank-indexed, to the finalize
 * kernels below; block b of the grid reduces the slab rows of its 8 channels, writes them into its cell of every mailbox, waits
 * (bounded: tss_bn_xchg_error) for the peers' cells and adds them up in rank order.  counters: tss_bn_xchg_counters() zero-filled
 * 64-bit words of ordinary device memory per rank (call counters; a replayed HIP graph advances them on the device).
 * Semantics of tss_bn_finalize_sync / tss_bn_bwd_finalize_sync (global statistics forward; global sums for the input gradient,
 * the replica's own for d(gamma), d(beta)).  C <= 768. */
long tss_bn_xchg_bytes(void);
int tss_bn_xchg_counters(void);
int tss_ipc_alloc(long bytes, void** ptr_out, void* handle_out);
int tss_ipc_open(const void* handle, void** ptr_out);
int tss_ipc_close(void* ptr);
int tss_ipc_free(void* ptr);
int tss_bn_xchg_error(const void* mailbox, long* out);
int tss_bn_finalize_xchg(const double* sums, double count, const void* const* peers, int rank, int world, void* counters,
                         const float* gamma, float eps, float momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* mean_out, float* invstd_out, float* scale, int C, void* stream);
int tss_bn_bwd_finalize_xchg(const double* bstats, double count, const void* const* peers, int rank, int world, void* counters,
                             const float* invstd, const float* gamma, int accumulate, float* dgamma, float* dbeta,
                             float* ga, float* gb, float* gce, int C, void* stream);
/* e_in[p][k] = relu'(act(x))[p][k] * sum_n g[p][n] w[n][k],  g = ga*(e-gce) + gb*(yraw-gmu);
 * bstats (optional) = partial sums of e_in and e_in * (xraw - in_mean).  xraw/in_* NULL: plain dX, no mask.
 * wT_bf16 (optional, bf16 path): a current bf16 TRANSPOSE [K][N] of w written by tss_cast_weights.
 * wg_ws / wg_dw (optional): the workspace and dW of an earlier tss_pwconv_bwd_weight(..., defer_reduce = 1) call -- of this
 * layer (wg_P = 0) or of ANY 1x1 layer whose weight gradient has been launched (wg_P, wg_K, wg_N = that call's P, K, N): its
 * slot reduction is carried by the first blocks of this launch instead of a kernel of its own. */
int tss_pwconv_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                        const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                        const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                        void* e_in, long ldei, double* bstats, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                        long P, int K, int N, int dtype, void* stream);
/* dw[n][k] += sum_p g[p][n] * act(x[p][k]).  ws: f32 workspace of tss_pwconv_bwd_weight_ws(P, K, N, dtype) floats for the
 * blocks' partial tiles (summed deterministically by a second kernel, or -- defer_reduce = 1 -- by the backward-data
 * launch of a later tss_pwconv_bwd_data / _radd call, or by tss_pwconv_wg_reduce); ws NULL, or a size of 0: f32 atomics onto dw.
 * fin (optional): a BatchNorm-backward finalize of ANOTHER layer (tss_bn_bwd_finalize's arguments) that rides in front of this
 * launch's grid: the weight gradient of a layer depends on nothing the backward pass computes after that layer's input
 * gradient, so the caller may postpone it until the next finalize is due and save that launch (>= 4.7 us in a replayed graph). */
int tss_pwconv_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                          const float* ga, const float* gb, const float* gce, const float* gmu,
                          const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                          float* dw, float* ws, int defer_reduce, long P, int K, int N, int dtype, const tss_bn_bwd_job* fin,
                          void* stream);
long tss_pwconv_bwd_weight_ws(long P, int K, int N, int dtype);
/* forward of a 1x1 layer over the channel CONCATENATION of nsrc (2..6) bf16 tensors of 128 channels each, every one with its own
 * pending BatchNorm (means / scales / biases[i], any of them NULL) and a common ReLU flag; the concatenated tensor is never written.
 * lds[i] == 0: source i is a single row broadcast to all P pixels.  w: [N][nsrc * 128] f32 (w_bf16: its bf16 copy or NULL).
 * replaces: torch.cat(branches, dim=1) + the 1x1 `project` conv of a DeepLab-style ASPP head (models/aspp.py), eval mode. */
int tss_pwconv_fwd_multi(const void* const* srcs, const long* lds, const float* const* means, const float* const* scales,
                         const float* const* biases, int nsrc, int in_relu, const float* w, const void* w_bf16,
                         const float* bias, void* y, long ldy, long P, int N, int dtype, void* stream);
/* the slot reduction of a tss_pwconv_bwd_weight(..., defer_reduce = 1) call that no later launch carried */
int tss_pwconv_wg_reduce(const float* ws, float* dw, long P, int K, int N, void* stream);

/* ---- dense 3x3 convolution, padding = dilation ------------------------------------------------------
 * replaces: nn.Conv2d(128,128,3,padding=1) of ConvBlock TSS/models/contextnet.py:55 (and any Conv2dBlock k=3, Cin%8==0).
 * fwd takes the weight re-laid out as [9][N][Cin], bwd_data as [9][Cin][N] (tss_permute_w3x3); bwd_weight
 * accumulates straight into the torch layout [N][Cin][3][3]. */
int tss_permute_w3x3(const float* w, float* w_tnc, float* w_tcn, int N, int Cin, void* stream);
/* the same two layouts as bf16 (either may be NULL).  With a bf16 copy and stride 1, dilation 1, contraction in {32, 64, 128},
 * outputs % 16 == 0 and <= 128, bf16 activations, fwd / bwd_data run the LDS-halo kernel of conv3x3.hip (one 64-pixel
 * row segment per tile, nine shifted LDS views, double-buffered weight taps); the f32 layout may then be NULL. */
int tss_permute_w3x3_bf16(const float* w, void* w_tnc_bf16, void* w_tcn_bf16, int N, int Cin, void* stream);
int tss_conv3x3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const void* w_tnc_bf16, void* y, long ldy, double* stats,
                    int B, int Hin, int Win, int Cin, int N, int stride, int dil, int dtype, void* stream);
int tss_conv3x3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* w_tcn_bf16,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int H, int W, int Cin, int N, int dil, int dtype, void* stream);  /* stride 1 */
int tss_conv3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int Hin, int Win, int Cin, int N, int stride, int dil,
                           int dtype, void* stream);
/* bf16 unfold for the dense 3x3 weight gradient: col[p][c*9 + tap] = act(x[p + off(tap)][c]) (0 outside the image), stride 1,
 * padding = dilation.  dW ([N][Cin][3][3]) is then tss_pwconv_bwd_weight(e, ..., x = col, ldx = 9*Cin, no affine, K = 9*Cin). */
int tss_im2col3x3(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                  void* col, int B, int H, int W, int C, int dil, int dtype, void* stream);

/* ---- factorized (1-D) dense convolution, 3 taps along one axis, padding = dilation, stride 1 ------------------------------
 * replaces: nn.Conv2d(C, C, (1,3), padding=(0,d), dilation=(1,d)) / nn.Conv2d(C, C, (3,1), padding=(d,0), dilation=(d,1)) of
 *           FactorizedConvBlock TSS/models/lednet.py:157-180 (SS-nbt block :95-124) and TSS/models/esnet.py:83-166.
 * axis 0: along W (the 1x3 kernel), axis 1: along H (3x1).  fwd takes the weight as [3][N][Cin], bwd_data as [3][Cin][N]
 * (tss_permute_wtaps with T = 3 from torch's [N][Cin][1][3] / [N][Cin][3][1]); bwd_weight accumulates into the torch layout. */
int tss_permute_wtaps(const float* w /* [N][Cin][T] */, float* w_tnc, float* w_tcn, int N, int Cin, int T, void* stream);
int tss_conv1d3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                    int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
int tss_conv1d3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
int tss_conv1d3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
/* forward / backward-data on the layer's own weight tensor (torch's [N][Cin][1][3] / [N][Cin][3][1]; no permuted copy) where the lean bf16
 * kernels cover the shape (tss_conv1d3_lean_supported: square 16 / 32 / 64-channel layers; csrc/fc1d.hip) */
int tss_conv1d3_lean_supported(int Cin, int N, int dtype);
int tss_conv1d3_fwd_w(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                      const float* w, const float* bias, void* y, long ldy, double* stats,
                      int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
int tss_conv1d3_bwd_data_w(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           void* e_in, long ldei, double* bstats,
                           int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
/* weight gradient of a square 16 / 32 / 64-channel three-tap layer in ONE sweep over e, y and x (bf16; csrc/fc1d.hip): every block leaves a
 * row of partial sums (3 * N * Cin floats, torch's [N][Cin][taps] order) in ws[tss_conv1d3_bwd_weight_rows(...)][3*N*Cin]; the rows are
 * added to the gradient by tss_dw_reduce_many.  _rows returns 0 when the shape is not covered (use tss_conv1d3_bwd_weight). */
int tss_conv1d3_bwd_weight_rows(long P, int Cin, int N, int dtype);
int tss_conv1d3_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                 const float* ga, const float* gb, const float* gce, const float* gmu,
                                 const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                                 int in_relu, float* ws, int B, int H, int W, int Cin, int N, int axis, int dil, int dtype, void* stream);
/* weight gradient of a square T-tap layer along one axis -- 64 channels x 5 taps (FCUBlock(64, 5), TSS/models/esnet.py:83-123) or 128 channels
 * x 3 taps (FPCUBlock, esnet.py:126-166) -- in one sweep (bf16; csrc/fcg.hip): per-block rows of partial sums (T * N * Cin floats, torch's
 * [N][Cin][taps] order) in ws[tss_convtap_bwd_weight_rows(...)][T*N*Cin], added by tss_dw_reduce_many.  rows == 0: not covered.
 * (Forward and backward-data of these layers are taken inside tss_conv1d3_* / tss_convkxk_*.) */
int tss_convtap_bwd_weight_rows(long P, int Cin, int N, int T, int dtype);
int tss_convtap_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                 const float* ga, const float* gb, const float* gce, const float* gmu,
                                 const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                                 int in_relu, float* ws, int B, int H, int W, int Cin, int N, int T, int axis, int dil, int dtype, void* stream);
/* weight gradient of a square 32 / 64-channel stride-2 dense 3x3 (padding 1) in one sweep (bf16; csrc/sconv.hip): per-block rows of partial
 * sums (9 * N * Cin floats, torch's [N][Cin][3][3] order) in ws[tss_sconv_bwd_weight_rows(...)][9*N*Cin], added by tss_dw_reduce_many.
 * replaces: the convolution arm of DownsamplingBlock, TSS/models/lednet.py:130-131, TSS/models/esnet.py:54-56.  rows == 0: not covered.
 * (Forward and backward-data of these layers are taken inside tss_conv3x3_fwd / tss_convkxk_fwd / tss_convkxk_bwd_data.) */
int tss_sconv_bwd_weight_rows(int B, int Hin, int Win, int Cin, int N, int dtype);
int tss_sconv_bwd_weight_sweep(const void* e, long lde, const void* yraw, long ldyr,
                               const float* ga, const float* gb, const float* gce, const float* gmu,
                               const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias,
                               int in_relu, float* ws, int B, int Hin, int Win, int Cin, int N, int dtype, void* stream);
/* bf16 unfold for the weight gradient of a three-tap layer: col[p][c*3 + tap] = act(x[p + off(tap)][c]) (0 outside the image); dW in torch's
 * [N][C][1][3] / [N][C][3][1] layout is then tss_pwconv_bwd_weight(e, ..., x = col, ldx = 3*C, no affine, K = 3*C). */
int tss_im2col1d3(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                  void* col, int B, int H, int W, int C, int axis, int dil, int dtype, void* stream);
/* ---- general dense convolution: kh x kw taps (odd sides), padding = dilation * (k - 1) / 2 on each axis, any stride, optional bias ---
 * replaces: the strided 3x3 / 5x5 / 7x7 ConvBlocks of APNModule TSS/models/lednet.py:62-64, the strided 3x3 (with bias) of
 *           DownsamplingBlock lednet.py:130-131 / esnet.py:54-56, and the 1x5 / 5x1 layers of FCUBlock esnet.py:83-113.
 * Weights as [kh*kw][N][Cin] (fwd) / [kh*kw][Cin][N] (bwd_data) from tss_permute_wtaps(T = kh*kw); bwd_weight accumulates into the
 * torch layout [N][Cin][kh][kw].  bwd_data: e is [B][(Hin-1)/stride+1][(Win-1)/stride+1][N], e_in is [B][Hin][Win][Cin]. */
int tss_convkxk_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                    const float* w_tnc, const float* bias, void* y, long ldy, double* stats,
                    int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil, int dtype, void* stream);
int tss_convkxk_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                         const float* ga, const float* gb, const float* gce, const float* gmu, const float* w_tcn,
                         const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                         void* e_in, long ldei, double* bstats,
                         int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil, int dtype, void* stream);
int tss_convkxk_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           float* dw, int B, int Hin, int Win, int Cin, int N, int kh, int kw, int stride, int dil,
                           int dtype, void* stream);

/* nn.ConvTranspose2d(Cin_t, Cout, k, stride, padding = (k-1)/2, output_padding = stride-1) of UpsamplingBlock TSS/models/esnet.py:71-80:
 * x [B][Hout/stride][Wout/stride][Cin_t] -> y [B][Hout][Wout][Cout] (+ bias); w_tcn = tss_permute_wtaps(weight as [N = Cin_t][Cin = Cout]
 * [T = kh*kw]).  Its backward is tss_convkxk_fwd (input gradient) and tss_convkxk_bwd_weight with the roles of x and the gradient swapped. */
int tss_convkxk_transposed_fwd(const void* x, long ldx, const float* w_tcn, const float* bias, void* y, long ldy,
                               int B, int Hout, int Wout, int Cout, int Cin_t, int kh, int kw, int stride, int dtype, void* stream);

/* ---- glue of LEDNet / ESNet (csrc/zoo.hip) -------------------------------------------------------------------------------
 * tss_tensor_stats: BatchNorm statistics (slab rows [tss_stat_slabs()][2C], sums and raw second moments) of a materialised tensor:
 *   the nn.BatchNorm2d after torch.cat([conv(x), pool(x)]) of DownsamplingBlock TSS/models/lednet.py:126-144, esnet.py:47-68;
 * tss_bn_bwd_apply: that BatchNorm's input gradient dz = ga (e - gce) + gb (z - gmu) (gb NULL: ga e, frozen statistics);
 * tss_pool_concat_*: z = cat([y1 + bias (N1 channels), max_pool2d(x, 2) (Cin channels)]); x is addressed by element strides
 *   (an NHWC activation, or the NCHW f32 image: x_f32 = 1); bwd writes dx ([B][Hin][Win][Cin], dtype) from dz[:, N1:];
 * tss_mul_addrows_*: out = u * a + r[image] of APNModule lednet.py:86-90; ws: B * tss_rows_slices(B, HW) * C floats;
 * tss_scale_rows: out = x * m[image][channel] (nn.Dropout2d with the mask m drawn by the caller; its own backward);
 * tss_cat2_add: out = cat([gl, gr], channels) + gs (gs may be NULL): the input gradient of a unit that splits its input with
 *   torch.chunk(input, 2, 1) and also uses it whole (SSnbtBlock TSS/models/lednet.py:112-124). */
int tss_tensor_stats(const void* z, long ldz, long P, int C, double* stats, int dtype, void* stream);
int tss_bn_bwd_apply(const void* e, long lde, const void* z, long ldz, const float* ga, const float* gb, const float* gce,
                     const float* gmu, void* dz, long lddz, long P, int C, int dtype, void* stream);
int tss_pool_concat_fwd(const void* y1, long ld1, const float* bias, int N1, const void* x, int x_f32, long sxb, long sxc, long sxh,
                        long sxw, int Cin, void* z, long ldz, int B, int Hin, int Win, int dtype, void* stream);
int tss_pool_concat_bwd(const void* dz, long lddz, int N1, const void* x, int x_f32, long sxb, long sxc, long sxh, long sxw, int Cin,
                        void* dx, long lddx, int B, int Hin, int Win, int dtype, void* stream);
int tss_rows_slices(int B, long HW);
int tss_mul_addrows_fwd(const void* u, long ldu, const void* a, long lda, const void* r, long ldr, void* out, long ldo, int B, long HW,
                        int C, int dtype, void* stream);
int tss_mul_addrows_bwd(const void* g, long ldg, const void* u, long ldu, const void* a, long lda, void* du, long lddu, void* da,
                        long ldda, void* dr, long lddr, float* ws, int B, long HW, int C, int dtype, void* stream);
int tss_scale_rows(const void* x, long ldx, const float* m, void* out, long ldo, int B, long HW, int C, int dtype, void* stream);
int tss_cat2_add(const void* gl, long ldl, const void* gr, long ldr, const void* gs, long lds, void* out, long ldo, long P, int half,
                 int dtype, void* stream);
/* out[p][c] = c < C ? g[p][c] : 0 for c < CP: the gradient of the channel slice x[:, :C] of a tensor computed with CP channels (19 classes
 * padded to 24: UpsamplingBlock(16, num_classes) TSS/models/esnet.py:45, the `level` layers of APNModule TSS/models/lednet.py:65-69);
 * g may have any pitch >= C. */
int tss_pad_channels(const void* g, long ldg, int C, void* out, long ldo, int CP, long P, int dtype, void* stream);
/* ---- tail of LEDNet's SS-nbt unit in one pass each way (csrc/ssnbt.hip) ------------------------------------------------------------
 * replaces: cat([left(l), right(r)]) -> Dropout2d -> activation(input + x) -> channel_shuffle(x, 2) of SSnbtBlock.forward,
 *           TSS/models/lednet.py:112-124.  left / right: the branches' raw convolution outputs [P][C/2] with their pending BatchNorm as
 *           (mean, scale, shift); m: [B][C] dropout multipliers (0 or 1/(1-p)) or NULL.
 * fwd: out[p][2j+g] = relu(x[p][g C/2 + j] + m[b][g C/2 + j] * bn_g(raw_g[p][j])).
 * bwd: gs = unshuffle(d(out) * [out > 0]) (gradient of the skip); e = gs * m ([P][C] = left | right; NULL with m NULL: e == gs);
 *      stats_l / stats_r: BatchNorm-backward slab rows of the branches (sum e, sum e (raw - mean); [tss_stat_slabs()][C]). */
int tss_ssnbt_tail_fwd(const void* left, long ldl, const float* mean_l, const float* scale_l, const float* shift_l,
                       const void* right, long ldr, const float* mean_r, const float* scale_r, const float* shift_r,
                       const void* x, long ldx, const float* m, void* out, long ldo, int B, long HW, int C, int dtype, void* stream);
int tss_ssnbt_tail_bwd(const void* dout, long lddo, const void* out, long ldo,
                       const void* left, long ldl, const float* mean_l, const void* right, long ldr, const float* mean_r,
                       const float* m, void* e, long lde, void* gs, long ldgs, double* stats_l, double* stats_r,
                       int B, long HW, int C, int dtype, void* stream);

/* channel_shuffle(x, groups) TSS/models/lednet.py:183-188: y[:, j * groups + i] = x[:, i * (C / groups) + j] (its own inverse with
 * groups' = C / groups: the backward is the same entry) */
int tss_channel_shuffle(const void* x, long ldx, void* y, long ldy, long P, int C, int groups, int dtype, void* stream);

/* ---- stem: 3x3 stride-s conv on the NCHW image (Cin*9 <= 64), NHWC output ----------------------------
 * replaces: nn.Conv2d(in_channels,32,3,stride=2,padding=1) TSS/models/fastscnn.py:30, TSS/models/contextnet.py:38,48. */
int tss_stem3x3_fwd(const void* x_nchw, int x_is_f32, const float* w, void* y, long ldy, double* stats,
                    int B, int Cin, int Hin, int Win, int N, int stride, int dtype, void* stream);
int tss_stem3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu,
                           const void* x_nchw, int x_is_f32, float* dw,
                           float* ws /* [tss_stat_slabs()][N*28] f32 workspace, may be NULL (slower path) */,
                           int B, int Cin, int Hin, int Win, int N, int stride, int dtype, void* stream);

/* ---- depthwise 3x3 convolution, padding = dilation, weight [C][3][3] --------------------------------
 * replaces: nn.Conv2d(groups=in_channels) of DWConv2dBlock TSS/models/fastscnn.py:176-185, DSConv2dBlock :191-192,
 *           DWConvBlock TSS/models/contextnet.py:150-165. */
int tss_dwconv3x3_fwd(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                      const float* w, void* y, long ldy, double* stats,
                      int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream);
int tss_dwconv3x3_bwd_data(const void* e, long lde, const void* yraw, long ldyr,
                           const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                           const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                           void* e_in, long ldei, double* bstats, const float* wg_ws, float* wg_dw,
                           int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream);
/* tss_pwconv_bwd_data for a layer whose (materialised) input has a second consumer -- the skip of a residual block: the other
 * gradient of that tensor, radd (bf16 [P][K], pitch ldr), is added in the epilogue instead of by an elementwise launch:
 * e_in = g W + radd.  Lean bf16 path only (tss_pwconv_bwd_data_radd_supported); no producer mask / statistics (input materialised). */
int tss_pwconv_bwd_data_radd_supported(long P, int K, int N, int dtype);
/* ... and for a layer whose materialised input is the OUTPUT of a relu join (a block output, relu(BN(join_y) + skip), TSS/models/
 * fastscnn.py:158-161): that join's backward runs in this launch's epilogue -- e_in = (g W + radd) where join_out > 0, else 0, and
 * join_bstats = slab rows of sum(e_in), sum(e_in * (join_y - join_mean)).  radd optional.  Only valid when e_in is the COMPLETE
 * gradient of the join's output; tss_join_bwd on the summed gradient stays the exact fallback. */
int tss_pwconv_bwd_data_joined(const void* e, long lde, const void* yraw, long ldyr,
                               const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                               void* e_in, long ldei, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                               const void* radd, long ldr, const void* join_out, long ldjo, const void* join_y, long ldjy,
                               const float* join_mean, double* join_bstats, long P, int K, int N, int dtype, void* stream);
int tss_pwconv_bwd_data_radd(const void* e, long lde, const void* yraw, long ldyr,
                             const float* ga, const float* gb, const float* gce, const float* gmu, const float* w, const void* wT_bf16,
                             void* e_in, long ldei, const float* wg_ws, float* wg_dw, long wg_P, int wg_K, int wg_N,
                             const void* radd, long ldr, long P, int K, int N, int dtype, void* stream);
/* 1x1 layer, backward in ONE sweep (csrc/pwbwd.hip; bf16, Cin, Cout <= 128, both multiples of 8): e, yraw and x are read once,
 * e_in written once -- replaces tss_pwconv_bwd_weight + tss_pwconv_bwd_data where their double read of (e, yraw) dominates
 * (tss_pwconv_bwd_fused_preferred: few channels, many pixels).  x is the layer's input (always given); x_pending = 1 when it is a
 * producer's raw output (in_* pending): e_in is then masked and bstats written as tss_pwconv_bwd_data does.  ws receives
 * tss_pwconv_bwd_fused_rows(P, Cin, Cout) rows of Cout*Cin floats (per-block partial sums of dW in the parameter's own
 * [Cout][Cin] order), to be added to dW by tss_dw_reduce_many.  wT_bf16: optional bf16 [Cin][Cout] shadow of w.
 * bias_ws (optional): rows of Cout floats, per-block partial sums of g over the pixels -- the bias gradient of a biased conv
 * (nn.Conv2d(128, classes, 1) of the classifiers, TSS/models/fastscnn.py:97), summed by tss_dw_reduce_many like the others.
 * Cin <= 128, Cout <= 64 (and not both <= 64): Cout may be ragged (19 classes); e / yraw are then read through their pitch
 * (>= Cout rounded up to 8) and whatever sits in the padding is ignored. */
int tss_pwconv_bwd_fused_preferred(long P, int Cin, int Cout, int dtype);
int tss_pwconv_bwd_fused_rows(long P, int Cin, int Cout);
int tss_pwconv_bwd_fused(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                         const float* gmu, const float* w, const void* wT_bf16, const void* x, long ldx, const float* in_mean,
                         const float* in_scale, const float* in_bias, int in_relu, int x_pending, void* e_in, long ldei,
                         double* bstats, float* ws, float* bias_ws, long P, int Cin, int Cout, int dtype, void* stream);
/* The same one sweep for the LARGE 1x1 layers (csrc/pwsweep.hip, round 4; bf16): Cin -> Cout = 128 -> 128, 64 -> 384, 384 -> 64 (and
 * 32 -> 192, 192 -> 32: the inverted residuals of TSS/models/contextnet.py:129-147 at 1/8 of the context image) -- the
 * classifier / fusion layers and the 6x bottleneck expand / project layers of TSS/models/fastscnn.py:138-161,188-199 at 1/8 and 1/16
 * resolution -- where tss_pwconv_bwd_data + tss_pwconv_bwd_weight read (e, yraw) twice.  One 512-thread block per CU keeps the whole
 * [Cout][Cin] weight-gradient tile in registers; ws receives tss_pwconv_bwd_sweep_rows(P, Cin, Cout) rows of Cout*Cin floats (summed
 * by tss_dw_reduce_many).  wT_bf16: bf16 TRANSPOSE [Cin][Cout] of w (required).  x_pending = 1: x is a producer's raw output (in_*
 * pending): e_in is masked and bstats written as tss_pwconv_bwd_data does.  radd (optional, x_pending = 0 only): the other gradient
 * of the layer's input (the skip of a residual block), added in the epilogue as tss_pwconv_bwd_data_radd does.  P must be a multiple
 * of the tile (64 pixels for 128 -> 128 and 32 -> 192, else 32); the expand shapes take a materialised x only, the project shapes a pending one:
 * tss_pwconv_bwd_sweep_preferred says whether the entry covers the layer and pays. */
int tss_pwconv_bwd_sweep_preferred(long P, int Cin, int Cout, int x_pending, int dtype);
int tss_pwconv_bwd_sweep_rows(long P, int Cin, int Cout);
int tss_pwconv_bwd_sweep(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                         const float* gmu, const void* wT_bf16, const void* x, long ldx, const float* in_mean,
                         const float* in_scale, const float* in_bias, int in_relu, int x_pending, const void* radd, long ldr,
                         void* e_in, long ldei, double* bstats, float* ws, long P, int Cin, int Cout, int dtype, void* stream);
/* backward-data AND weight gradient of one layer in a single sweep (bf16; stride/dilation of the strip kernels): e, yraw
 * and x are read once.  x is the layer's input (always given: the weight gradient needs it); x_pending = 1 when it is a
 * producer's raw output whose BatchNorm(+ReLU) is still pending (in_* describe it): e_in is then masked and bstats
 * written, exactly as tss_dwconv3x3_bwd_data does with xraw.  ws: [TSS_STAT_SLABS][C*9] f32 rows of per-block partial
 * sums, added to dw ([C][1][3][3]) by a second small launch.  tss_dwconv3x3_bwd_fused_supported: 1 when this entry
 * covers the shape. */
int tss_dwconv3x3_bwd_fused_supported(int C, int stride, int dil, int dtype);
/* The same sweep without the second launch: the per-block rows stay in ws, *rows_out (host) receives their number, and the caller
 * adds them to the weight gradients of many layers at once with tss_dw_reduce_many (arrays of njobs device pointers / sizes in
 * HOST memory: ws[j] = rows[j] x n[j] floats, dw[j][i] += sum over the rows; n = C * 9).  A training step has one such 5 us
 * reduction per depthwise layer and nothing but the optimizer waits for them. */
int tss_dwconv3x3_bwd_fused_sweep(const void* e, long lde, const void* yraw, long ldyr,
                                  const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                                  const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                                  int x_pending, void* e_in, long ldei, double* bstats, float* ws,
                                  int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream, int* rows_out);
int tss_dw_reduce_many(int njobs, const float* const* ws, float* const* dw, const int* n, const int* rows, void* stream);
/* 1 when the one-sweep backward is also the faster choice for this layer (row-pipelined kernels, csrc/dwroll.hip) */
int tss_dwconv3x3_bwd_fused_preferred(int C, int stride, int dil, int dtype);
int tss_dwconv3x3_bwd_fused(const void* e, long lde, const void* yraw, long ldyr,
                            const float* ga, const float* gb, const float* gce, const float* gmu, const float* w,
                            const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                            int x_pending, void* e_in, long ldei, double* bstats, float* ws, float* dw,
                            int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream);
int tss_dwconv3x3_bwd_weight(const void* e, long lde, const void* yraw, long ldyr,
                             const float* ga, const float* gb, const float* gce, const float* gmu,
                             const void* xraw, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                             float* dw, float* ws /* [tss_stat_slabs()][C*9] f32 workspace */, int defer_reduce,
                             int B, int Hin, int Win, int C, int stride, int dil, int dtype, void* stream);
/* defer_reduce = 1: the workspace rows are summed into dw by the tss_dwconv3x3_bwd_data call of the SAME layer that
 * follows (its wg_ws / wg_dw arguments) instead of a kernel of its own. */

/* ---- bilinear upsample (align_corners=True) + depthwise 3x3 (padding = dilation) as one operator (csrc/updw.hip) ------
 * replaces: nn.UpsamplingBilinear2d(scale_factor) -> DWConv2dBlock(dilation=scale_factor) of FeatureFusionModule.lowres,
 *           TSS/models/fastscnn.py:74-76; F.interpolate(size=...) -> DWConvBlock(dilation=4), TSS/models/contextnet.py:110-122.
 * x: materialised bf16 source [B][Hs][Ws][C]; the [B][Ho][Wo][C] upsampled tensor never exists.  bf16 only.
 * fwd: y raw conv output + its statistics slab rows.  bwd (one sweep over e, yraw): e_up = gradient with respect to the
 * UPSAMPLED map (bf16 [B][Ho][Wo][C]; tss_bilinear_nhwc_bwd folds it back to the source), ws = [tss_updw_ws_rows(...)][C*9] f32 rows
 * of per-block weight-gradient sums, *rows_out of them written (host int), summed by tss_dw_reduce_many.
 * ga..gmu: coefficients of the BatchNorm backward behind the layer (tss_bn_bwd_finalize); yraw NULL: g = ga * e (ga NULL: e). */
int tss_updw_supported(int B, int Hs, int Ws, int Ho, int Wo, int C, int dil, int dtype);
int tss_updw_ws_rows(int B, int Hs, int Ws, int Ho, int Wo, int C, int dil, int dtype);   /* rows of `ws` tss_updw_bwd writes */
int tss_updw_fwd(const void* x, long ldx, int Hs, int Ws, const float* w, void* y, long ldy, double* stats,
                 int B, int Ho, int Wo, int C, int dil, int dtype, void* stream);
int tss_updw_bwd(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                 const float* gmu, const float* w, const void* x, long ldx, int Hs, int Ws, void* e_up, long ldeu, float* ws,
                 int B, int Ho, int Wo, int C, int dil, int dtype, void* stream, int* rows_out);

/* Per-channel statistics buffers (`stats`, `bstats`, `stats_a/b` everywhere in this header) are
 * [tss_stat_slabs()][2*C] f64: every producing kernel writes one partial row per block and zeroes the rest, so
 * the caller allocates them uninitialised; tss_bn_finalize / tss_bn_bwd_finalize sum the rows. */
int tss_stat_slabs(void);

/* ---- BatchNorm2d bookkeeping (eps, momentum, running stats exactly as torch.nn.BatchNorm2d) ----------
 * replaces: nn.BatchNorm2d in every block above (training: batch statistics + running update; eval: running stats). */
int tss_bn_finalize(const double* sums, double count, const float* gamma, float eps,
                    float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* mean_out, float* invstd_out, float* scale, int C, void* stream);
int tss_bn_eval_affine(const float* gamma, const float* running_mean, const float* running_var,
                       float eps, float* mean_out, float* invstd_out, float* scale, int C, void* stream);
/* all eval-mode BatchNorm layers of a model in one launch.  table: njobs x 6 int64 on the device: (gamma or 0, running_mean,
 * running_var, out [3][C] = mean | invstd | scale, C, eps as f32 bits); max_channels = the largest C of the table. */
int tss_bn_eval_affine_batched(const long long* table, int njobs, int max_channels, void* stream);
/* bstats = [sum(e), sum(e*(y-mean))]; writes d(gamma), d(beta) (+= when accumulate) and ga, gb, gce */
int tss_bn_bwd_finalize(const double* bstats, double count, const float* invstd,
                        const float* gamma, int training, int accumulate, float* dgamma, float* dbeta,
                        float* ga, float* gb, float* gce, int C, void* stream);

/* ---- cross-replica (Sync) BatchNorm: apex.parallel.convert_syncbn_model TSS scripts/train_fastscnn.py:144-145 -------
 * tss_slab_reduce sums a replica's slab rows into out[0..2C) and stores `count` at out[2C]; the caller all-reduces the
 * [2C+1] f64 vector over the ranks (one RCCL collective per BatchNorm layer and direction) and hands it to the *_sync
 * finalize kernels: forward = global batch statistics (also into the running statistics), backward = global sums for
 * the input gradient, this replica's slab sums (bstats) for d(gamma), d(beta). */
int tss_slab_reduce(const double* slabs, double count, double* out, int C, void* stream);
int tss_bn_finalize_sync(const double* gsums, const float* gamma, float eps, float momentum, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float* mean_out, float* invstd_out,
                         float* scale, int C, void* stream);
int tss_bn_bwd_finalize_sync(const double* bstats, const double* gsums, const float* invstd, const float* gamma,
                             int accumulate, float* dgamma, float* dbeta, float* ga, float* gb, float* gce, int C,
                             void* stream);

/* ---- join: out = relu?(affA(a) + affB(b)) ------------------------------------------------------------
 * replaces: the trailing BatchNorm2d(+ReLU) of a block, `x + input` / F.relu of BottleneckBlock
 *           TSS/models/fastscnn.py:158-161, TSS/models/contextnet.py:145-147 and F.relu(lowres + highres)
 *           TSS/models/fastscnn.py:89, TSS/models/contextnet.py:126. */
int tss_join_fwd(const void* a, long lda, const float* ma, const float* sa, const float* ba,
                 const void* b, long ldb, const float* mb, const float* sb, const float* bb,
                 void* out, long ldo, int relu, float drop_p, const unsigned long long* seed_slot,
                 long P, int C, int dtype, void* stream);
/* seed_slot != NULL (needs relu): nn.Dropout(drop_p) of the Classifier (TSS/models/fastscnn.py:96, contextnet.py:85) applied to
 * the joined activation in the same pass, with the mask tss_dropout would draw from that slot.
 * e = dout_scale * dout * relu'(out) (written if e != NULL); stats_x (optional) = partial sums of e and e * (x_raw - mean_x).
 * dout_scale = 1/(1 - p) for a join that applied dropout (out > 0 <=> kept and active, so no mask is regenerated), else 1. */
int tss_join_bwd(const void* dout, long lddo, const void* out, long ldo, int relu,
                 const void* a_raw, long lda, const float* mean_a, double* stats_a,
                 const void* b_raw, long ldb, const float* mean_b, double* stats_b,
                 void* e, long lde, float dout_scale, long P, int C, int dtype, void* stream);

/* ---- dropout / bias gradient / optimizer ------------------------------------------------------------
 * replaces: nn.Dropout(0.1) TSS/models/fastscnn.py:96, TSS/models/contextnet.py:85 (Philox; mask recomputed in backward);
 *           torch.optim.AdamW.step() as called by TSS/engine.py:38 (single flat tensor, torch arithmetic). */
int tss_dropout_tick(unsigned long long* counter, unsigned long long* seed_slot, void* stream);
int tss_dropout(const void* x, long ldx, void* y, long ldy, long P, int C, float p,
                const unsigned long long* seed_slot, int dtype, void* stream);
int tss_bias_grad(const void* e, long lde, long P, int N, float* dbias, int dtype, void* stream);
/* nn.Dropout in front of a 1x1 convolution (Classifier: ... -> BN -> ReLU -> Dropout(0.1) -> Conv2d(128, classes, 1),
 * TSS/models/fastscnn.py:96-97, TSS/models/contextnet.py:85-86) applied ON LOAD by that convolution instead of by a pass of its
 * own over the 128-channel activation (forward and backward).
 * tss_dropout_mask: mask = [P][16] bytes (C <= 128: a pixel's bytes are one aligned 16-byte row), bit j of byte (p, v) = channel
 *   8 v + j of pixel p is kept; Philox4x32-10 keyed by *counter (device), 16 bits per element.  It does NOT advance the counter: the consumer does.
 * tss_pwconv_fwd_drop: y = (keep / (1 - p) * act(x)) W^T + bias, act = the pending BatchNorm(+ReLU) of x; advances *counter.
 * tss_pwconv_bwd_fused_drop: tss_pwconv_bwd_fused with the same mask on the weight gradient's activation operand and on e_in;
 *   ws rows: tss_pwconv_bwd_fused_drop_rows(P); yraw must be NULL (no BatchNorm behind the convolution).  bf16; K = Cin <= 128 (multiple of 8), N = Cout <= 64 (>= 8, may be ragged). */
int tss_dropout_mask(const unsigned long long* counter, void* mask, long P, int C, float p, void* stream);
int tss_pwconv_drop_supported(long P, int K, int N, int dtype);
int tss_pwconv_fwd_drop(const void* x, long ldx, const float* in_mean, const float* in_scale, const float* in_bias, int in_relu,
                        const float* w, const void* w_bf16, const float* bias, void* y, long ldy,
                        const void* mask, float drop_p, unsigned long long* counter,
                        long P, int K, int N, int dtype, void* stream);
int tss_pwconv_bwd_fused_drop_supported(long P, int Cin, int Cout, int dtype);
int tss_pwconv_bwd_fused_drop_rows(long P);
int tss_pwconv_bwd_fused_drop(const void* e, long lde, const void* yraw, long ldyr, const float* ga, const float* gb, const float* gce,
                              const float* gmu, const float* w, const void* x, long ldx, const float* in_mean,
                              const float* in_scale, const float* in_bias, int in_relu, int x_pending, const void* mask, float drop_p,
                              void* e_in, long ldei, double* bstats, float* ws, float* bias_ws, long P, int Cin, int Cout, int dtype,
                              void* stream);
/* bf16 shadows of 1x1 weights, all layers in one launch.  table: njobs x 5 int64 on the device:
 * (f32 source [N][K], bf16 copy [N][K], bf16 transpose [K][N], N, K); blocks_per_job x njobs blocks of 256 threads.
 * zero / zero_n (optional): a float buffer cleared by the same launch -- the step's flat gradient buffer (optimizer.zero_grad()
 * of TSS/engine.py:28), which is due at the same point of the step. */
int tss_cast_weights(const long long* table, int njobs, int blocks_per_job, float* zero, long zero_n, void* stream);
/* torch.optim.AdamW.step() on one flat buffer.  state != NULL: [step, 1 - beta1^step, sqrt(1 - beta2^step)] and the learning
 * rate *lr live on the device (a tick kernel advances them: the step can be replayed from a captured graph); state == NULL:
 * lr_host and step_host (1-based) are used instead and the whole step is ONE launch. */
int tss_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                   const float* lr, float beta1, float beta2, float eps, float weight_decay,
                   float* state, float grad_scale, float lr_host, long step_host, void* stream);

/* ---- resampling -------------------------------------------------------------------------------------
 * replaces: F.interpolate(mode='bilinear', align_corners=True) TSS/models/fastscnn.py:63-64,119-120,
 *           nn.UpsamplingBilinear2d :74, TSS/models/contextnet.py:65-67,74-76,119-121;
 *           nn.AdaptiveAvgPool2d TSS/models/fastscnn.py:108; torch.cat :122 (tss_copy_nhwc into a channel slice). */
int tss_bilinear_nhwc_fwd(const void* x, long ldx, void* y, long ldy, int B, int Hin, int Win, int Hout, int Wout,
                          int C, int dtype, void* stream);
int tss_bilinear_nhwc_bwd(const void* dy, long lddy, void* dx, long lddx, float* tmp /* [B*Hin*Wout*C] f32 */,
                          int B, int Hin, int Win, int Hout, int Wout, int C, int dtype, void* stream);
int tss_bilinear_planar_fwd(const void* x, int x_dtype, void* y, int y_dtype, long planes, int Hin, int Win,
                            int Hout, int Wout, void* stream);
/* logits head: NHWC low-res logits (pitch ldl) -> NCHW-contiguous full-res logits, and its backward */
int tss_upsample_head_fwd(const void* low, long ldl, void* y, int B, int N, int h, int w, int H, int W,
                          int dtype, void* stream);
int tss_upsample_head_bwd(const void* dy, const float* gscale, float* tmp, void* dlow, long ldl,
                          int B, int N, int h, int w, int H, int W, int dtype, void* stream);
int tss_adaptive_pool_fwd(const void* x, long ldx, void* y, long ldy, int B, int H, int W, int C, int bins,
                          int dtype, void* stream);
int tss_adaptive_pool_bwd(const void* dy, long lddy, void* dx, long lddx, int B, int H, int W, int C, int bins,
                          int dtype, void* stream);
/* PyramidPoolingModule (TSS/models/fastscnn.py:101-123), the element-wise stages of ALL arms per launch (narms <= 4; the
 * pointer / size arrays are HOST arrays of narms entries):
 *   pool_fwd   : y[a] = AdaptiveAvgPool2d(bins[a])(x), NHWC [B][bins][bins][C]; with few images the windows are cut into S row
 *                slices (tss_ppm_pool_slices) whose partial sums meet in ws, combined by a second small launch
 *   pool_bwd   : dx = sum_a pool_a^T(dy[a])
 *   concat_fwd : out = cat(x, upsample(relu?(bn_a(raw[a]))) for every arm), align_corners=True, raw[a] = [B][bins][bins][ca];
 *                mean/scale/beta[a] NULL: raw[a] is already the activation
 *   concat_bwd : e[a] = relu'(.) * upsample^T(dout[:, C + a*ca : C + (a+1)*ca]) and, when bstats[a] != NULL, the slab rows
 *                (sum e, sum e*(raw - mean)) of the arm's BatchNorm backward (needs B * bins^2 <= tss_stat_slabs()) */
int tss_ppm_pool_slices(int B, int ncells);   /* S: row slices per window pool_fwd uses when given a workspace (1: none needed) */
int tss_ppm_pool_fwd(const void* x, long ldx, void* const* y, const long* ldy, const int* bins, int narms,
                     float* ws /* NULL, or B * sum(bins^2) * S * C floats (partial sums, no initialisation needed) */,
                     int B, int H, int W, int C, int dtype, void* stream);
/* radd (optional, [B][H][W] pixels of C channels, pitch ldr): another gradient of the pooled-from map, added to dx here (the map
 * feeds the pools AND the concat, TSS/models/fastscnn.py:118-121; autograd would add the two gradients with a launch of its own) */
int tss_ppm_pool_bwd(const void* const* dy, const long* lddy, const int* bins, int narms, void* dx, long lddx,
                     const void* radd, long ldr, int B, int H, int W, int C, int dtype, void* stream);
int tss_ppm_concat_fwd(const void* x, long ldx, const void* const* raw, const long* ldr, const int* bins,
                       const float* const* mean, const float* const* scale, const float* const* beta, const int* relu,
                       int narms, void* out, long ldo, int B, int H, int W, int C, int ca, int dtype, void* stream);
int tss_ppm_concat_bwd(const void* dout, long lddo, const void* const* raw, const long* ldr, const int* bins,
                       const float* const* mean, const float* const* scale, const float* const* beta, const int* relu,
                       double* const* bstats, void* const* e, const long* lde, int narms,
                       int B, int H, int W, int C, int ca, int dtype, void* stream);
/* The arms of a pyramid pooling module between the pools and the concat, ALL arms in one launch, one block per arm
 * (replaces, per arm: nn.Conv2d(in, in / 4, 1) + nn.BatchNorm2d of Conv2dBlock TSS/models/fastscnn.py:106-112 -- i.e.
 * tss_pwconv_fwd + tss_bn_finalize forward, tss_bn_bwd_finalize + tss_pwconv_bwd_weight + tss_pwconv_bwd_data backward).
 * x[a]: pooled map [P[a]][C] (P[a] = B * bins^2 <= 512), w[a]: [Ca][C] f32, y[a]: raw conv output [P[a]][Ca],
 * vec[a]: [6][Ca] f32 = mean | invstd | gamma * invstd | ga | gb | gce (forward writes the first three in training mode and
 * updates the running statistics as tss_bn_finalize does; backward reads them and writes the last three).
 * Backward: e[a] = d(loss)/d(BatchNorm output), masked; dw / dgamma / dbeta are overwritten, or added to when accumulate = 1;
 * e_in[a] [P[a]][C] = gradient of the pooled map.  bf16 only; C <= 128 and a multiple of 32, Ca 16 or 32. */
int tss_ppm_arms_supported(int narms, int C, int Ca, const int* P, int dtype);
int tss_ppm_arms_fwd(const void* const* x, const long* ldx, const float* const* w, const float* const* gamma, float* const* running_mean,
                     float* const* running_var, long long* const* num_batches_tracked, void* const* y, const long* ldy, float* const* vec,
                     const int* P, int narms, int C, int Ca, int training, float eps, float momentum, int dtype, void* stream);
int tss_ppm_arms_bwd(const void* const* e, const long* lde, const void* const* y, const long* ldy, const void* const* x, const long* ldx,
                     const float* const* w, const float* const* gamma, float* const* vec, float* const* dw, float* const* dgamma,
                     float* const* dbeta, int accumulate, void* const* e_in, const long* ldei, const int* P, int narms, int C, int Ca,
                     int training, int dtype, void* stream);
int tss_copy_nhwc(const void* x, long ldx, void* y, long ldy, long P, int C, int dtype, void* stream);

/* ---- channel-attention gates (BiSeNet) ---------------------------------------------------------
 * replaces: `x * sigmoid(conv(pool(x)))` of AttentionRefinementModule TSS/models/bisenet.py:144-148 (add_one = 0) and
 *           `x * (1. + attention)` of FeatureFusionModule TSS/models/bisenet.py:128-131 (add_one = 1).
 * x, out, g, dx: [B][HW][C] NHWC maps; a, da: [B][C] (one value per image and channel: the 1x1 convolution's output on the
 * pooled map, BEFORE the sigmoid).  Backward: dx = g * (sigmoid(a) + add_one), da = sigmoid'(a) * sum_pixels(g * x), summed
 * per row slice into ws (B * tss_gate_slices(B, HW) * C floats, not initialised) and then in slice order: no atomics. */
int tss_gate_slices(int B, long HW);
int tss_gate_fwd(const void* x, long ldx, const void* a, long lda, void* out, long ldo, int B, long HW, int C, float add_one,
                 int dtype, void* stream);
int tss_gate_bwd(const void* g, long ldg, const void* x, long ldx, const void* a, long lda, void* dx, long lddx, void* da, long ldda,
                 float* ws, int B, long HW, int C, float add_one, int dtype, void* stream);

/* ---- caller side: loss and evaluation metrics --------------------------------------------------------
 * replaces: nn.CrossEntropyLoss(ignore_index=255) scripts/train_fastscnn.py:132 as called by TSS/engine.py:30;
 *           argmax + ConfusionMatrix update of create_segmentation_evaluator TSS/engine.py:65-77. */
int tss_cross_entropy_fwd(const void* logits, const long long* target, float* lse, double* acc,
                          float* loss, float* inv_count, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream);
int tss_cross_entropy_bwd(const void* logits, const long long* target, const float* lse, const float* inv_count,
                          const float* grad_out, void* dlogits, long B, int C, long HW, int ignore_index,
                          int dtype, void* stream);
/* Online hard example mining cross-entropy: OHEMLoss(ignore_index, thresh_loss = -log 0.7, numel_frac = 0.01) of the
 * reference's training recipe (TSS/losses/ohem_loss.py:10-21, scripts/train_fastscnn.py:133-137).  n_top = int(B*HW *
 * numel_frac).  The (n_top+1)-th largest per-pixel loss is found by a device-side radix select (no sort, no host read-back);
 * workspace = tss_ohem_workspace_bytes() bytes, zeroed once by the caller (left zeroed by every call). */
long tss_ohem_workspace_bytes(void);
/* the selection alone, on n per-pixel losses that already exist (n % 4 == 0) */
int tss_ohem_select(const float* pixel_loss, void* workspace, float* loss, float* params, long n, float thresh_loss, long n_top,
                    void* stream);
int tss_ohem_fwd(const void* logits, const long long* target, float* lse, float* pixel_loss, void* workspace,
                 float* loss, float* params, long B, int C, long HW, int ignore_index, float thresh_loss,
                 long n_top, int dtype, void* stream);
int tss_ohem_bwd(const void* logits, const long long* target, const float* lse, const float* pixel_loss,
                 const float* params, const float* grad_out, void* dlogits, long B, int C, long HW, int ignore_index,
                 int dtype, void* stream);
/* Fused decoder head + loss: cross-entropy (mean over the non-ignored pixels) of the bilinearly upsampled logits,
 * straight from the low-res NHWC logits (replaces F.interpolate TSS/models/fastscnn.py:63-64 + the loss call
 * TSS/engine.py:30 as one operator; the full-resolution logits and their gradient are never materialised).
 * One pass computes the loss AND the unscaled low-res gradient, as per-block tiles and per-block loss rows in `ws`
 * (tss_upsample_ce_ws(...) floats, 16-byte aligned, NOT initialised by the caller: every element that is read was written by
 * the pass); backward = tss_upsample_ce_bwd gathers the tiles of every low-res cell in a fixed order: dlow = sum * grad_out /
 * count ([B][h][w][ldl], pitch padding zeroed).  No atomics anywhere: bit-identical from run to run.  C <= 24, H >= h, W >= w. */
long tss_upsample_ce_ws(int B, int C, int h, int w, int H, int W);
int tss_upsample_ce_fwd(const void* low, long ldl, const long long* target, float* ws, float* loss, float* inv_count,
                        int B, int C, int h, int w, int H, int W, int ignore_index, int dtype, void* stream);
int tss_upsample_ce_bwd(const float* ws, const float* inv_count, const float* grad_out, void* dlow, long ldl,
                        int B, int C, int h, int w, int H, int W, int dtype, void* stream);
/* OHEM on the fused head (TSS/losses/ohem_loss.py:10-21 on F.interpolate(low, x8) without the full-resolution logits): per-pixel
 * cross-entropy from the low-res logits -> tss_ohem_select on that array -> gradient tiles of the selected pixels
 * (tss_upsample_ohem_grad; ws as for tss_upsample_ce_fwd) -> tss_upsample_ce_bwd with *inv_count = 1 gathers them. */
int tss_upsample_pixel_ce(const void* low, long ldl, const long long* target, float* pix, int B, int C, int h, int w, int H, int W,
                          int ignore_index, int dtype, void* stream);
int tss_upsample_ohem_grad(const void* low, long ldl, const long long* target, const float* pix, const float* sel, float* ws,
                           int B, int C, int h, int w, int H, int W, int ignore_index, int dtype, void* stream);
int tss_upsample_head_bwd_cols(const float* tmp, void* dlow, long ldl, int B, int N, int h, int w, int W,
                               int dtype, void* stream);
/* Fused evaluation head: argmax of the bilinearly upsampled logits (+ confusion matrix, rows = truth) from the low-res
 * NHWC logits; the full-resolution logits are never materialised (model(x).argmax(1) of the evaluator TSS/engine.py:65-77). */
int tss_upsample_argmax_confusion(const void* low, long ldl, const long long* target, unsigned char* pred,
                                  unsigned long long* confusion, int B, int C, int h, int w, int H, int W,
                                  int ignore_index, int dtype, void* stream);
int tss_argmax_confusion(const void* logits, const long long* target, unsigned char* pred,
                         unsigned long long* confusion, long B, int C, long HW, int ignore_index,
                         int dtype, void* stream);

/* ---- host side of the step: the batch on the wire ----------------------------------------------------------------------
 * replaces: the float32 image / int64 target the reference's DataLoader produces (albumentations.Normalize + ToTensor,
 *           scripts/train_fastscnn.py:62-68) and copies to the device every iteration (TSS/engine.py:27).  The loader may ship
 *           uint8 pixels (HWC as decoded, or CHW) and uint8 labels instead (5x fewer PCIe bytes); this entry writes
 *           image_out[b][c][h][w] = (image/255 - mean[c]) / std[c] (f32 NCHW) and target_out = (int64) target into the
 *           buffers the step reads.  mean3 / std3 are HOST arrays of C floats (NULL: 0 / 1); either tensor may be NULL. */
int tss_decode_batch_u8(const unsigned char* image, int image_is_hwc, const float* mean3, const float* std3, float* image_out,
                        const unsigned char* target, long long* target_out, long B, int C, long HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSS_HIP_H */

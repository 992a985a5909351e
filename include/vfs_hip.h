/* libvfs_hip.so -- C ABI of the MI355X (gfx950) VFS hot path.
 *
 * The reference (xvjiarui/VFS) has no native code: every function below replaces a torch /
 * mmcv call site of its Python hot path (file:line under /root/reference cited per entry).
 * A maintainer of the reference binds these with ctypes underneath the
 * same registry classes (ResNet / SimSiamHead / CosineSimLoss / SimSiamBaseTracker /
 * VanillaTracker); vfs_amd/ is exactly that binding.
 *
 * Conventions
 *   - the caller owns every buffer (device pointers from its allocator); the library keeps no
 *     device state and never synchronises: all work is enqueued on `stream` (a hipStream_t)
 *   - activations: bf16 NHWC; packed weights: bf16 [Cout][KH][KW][Cin] (forward) and
 *     [Cin][KH][KW][Cout] (dgrad); master parameters / gradients: fp32 in the reference's
 *     layouts (OIHW conv, [out][in] linear) so state_dicts stay interchangeable
 *   - return 0 on success, negative on error; vfs_last_error() gives the message
 *   - thread-safe per stream; one process per GPU.  The operator entry points below read no mutable library state except the A/B
 *     switchboard of include/vfs_hip_tuning.h, which a production caller never touches (every knob defaults to the measured-best path)
 */
#ifndef VFS_HIP_H
#define VFS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vfs_stream_t; /* hipStream_t */
typedef uint16_t vfs_bf16;

const char* vfs_last_error(void);
int vfs_abi_version(void);
/* (Measurement knobs - vfs_set_option - are NOT part of this contract: include/vfs_hip_tuning.h.) */

/* ---- input / parameter layout -------------------------------------------------------------
 * imgs fp32 [B][2][3][T][H][W] (pipelines/formating.py:248-258) -> bf16 NHWC4 frames
 * out[(v*B+b)*T+t][h][w][4], i.e. video2images(imgs[:, v]) per view
 * (sim_siam_base_tracker.py:65-68, common/utils.py:45-53); width padded to even Wp with zeros */
int vfs_imgs_to_nhwc4(const float* imgs, vfs_bf16* out, int B, int V, int T, int H, int W, int Wp,
                      vfs_stream_t stream);

/* table-driven repack of ALL fp32 master weights into the bf16 MFMA layouts in one launch.
 * desc: device array of ntensors records {w, wf, wd, start, Cout, Cin, KH, KW, kind, tile_start}
 * (8-byte pointers/int64 then 6 int32; kind 1 = 7x7 stem -> [64][8][8][4]).  One workgroup per
 * 32x32 (cout x cin) tile of a tensor (per 256 elements of the stem); tile_start = running sum,
 * total_tiles = its end (KH*KW <= 25 except the stem).  kind 2 / 3 = kind 0 with 16-byte accesses for the shapes that hold a
 * ResNet's weights - 2: 1x1 with 64 | Cout, 64 | Cin (one workgroup per 64 x 64 tile), 3: 3x3 with 32 | Cout, 64 | Cin (per 32 x 64
 * tile); w, wf, wd 16-byte aligned; identical results (the table builder of vfs_amd/packing.py chooses them). */
int vfs_pack_weights(const void* desc, int ntensors, long long total_tiles, vfs_stream_t stream);

/* ---- convolution / linear: torch conv2d & linear call sites ---------------------------------
 * forward  (resnet.py:51-73,163-191,267-277 via mmcv ConvModule; sim_siam_head.py:78-111):
 *   y[N,Ho,Wo,Cout] = conv(x[N,H,W,Cin], wf) (+ bias)          Cin % 64 == 0, Cout % 64 == 0
 *   stats (optional): float[ceil(N*Ho*Wo/128)][2][Cout] per-128-pixel-block (sum, sumsq) of the
 *   stored bf16 outputs -- the BatchNorm batch statistics, free in the GEMM epilogue */
int vfs_conv_fwd(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats,
                 int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                 int pad, vfs_stream_t stream);
/* vfs_conv_fwd that ALSO sums its statistics rows in groups of 2^coarse_log2 consecutive rows (round 5): the last workgroup of a
 * group to arrive (device-scope ticket) adds the group's rows in row order (deterministic) into
 *   stats_coarse: float[ceil(rows / 2^coarse_log2)][2][Cout]          rows = what vfs_conv_fwd writes to `stats`
 * so that the consumer finishing the BatchNorm statistics in its prologue (vfs_bn_act_fin, at most 128 rows per group) can be
 * used for the large maps as well and the separate reduction launch between conv and BatchNorm (torch: the batch_norm op's own
 * statistics pass, resnet.py via mmcv ConvModule) disappears.  tickets: uint32 [ceil(rows / 2^L) * ceil(Cout / 64)], zero before
 * the first launch, left at zero.  `stats` is still written (the fine rows are the exchange medium). */
int vfs_conv_fwd_coarse(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, float* stats_coarse,
                        uint32_t* tickets, int coarse_log2, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                        int stride, int pad, vfs_stream_t stream);
/* 7x7/2 stem conv (resnet.py:422-434) on NHWC4 input, wf = [64][8][8][4].  stats rows: one per
 * spatial tile of 8x16 output pixels, N*ceil(Ho/8)*ceil(Wo/16) rows of [2][64] (image-major) */
int vfs_stem_fwd(const vfs_bf16* x4, const vfs_bf16* wf, vfs_bf16* y, float* stats, int N, int H,
                 int Wp, int Ho, int Wo, vfs_stream_t stream);
/* dgrad (autograd of the above): dx[N,H,W,Cin] = conv_transpose(dy[N,Ho,Wo,Cout], wd) (+ add) */
int vfs_conv_dgrad(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, int N,
                   int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                   int pad, vfs_stream_t stream);
/* the same input gradient, plus the BatchNorm-backward statistics of the unit whose OUTPUT this gradient
 * belongs to, from the epilogue (saves bn_bwd_reduce's pass over dx): bn_partial float[ceil(M/128)][2][Cin]
 * rows {sum g*mask, sum g*mask*xhat} per 128 output pixels, M = N*H*W; bn_x = that unit's raw conv output
 * [M][Cin], bn_y its activation for the ReLU mask (residual units) or NULL (bn_relu: mask recomputed
 * from bn_x), bnp float[G][4][Cin], bn_mpg pixels per group (multiple of 128).  stride 1 only.
 * Feed bn_partial to vfs_bn_bwd_sums_paramgrad / vfs_bn_reduce_partials with bpg = bn_mpg/128. */
int vfs_conv_dgrad_bn(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add,
                      const vfs_bf16* bn_x, const vfs_bf16* bn_y, const float* bnp, float* bn_partial,
                      int bn_mpg, int bn_relu, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                      int KW, int stride, int pad, vfs_stream_t stream);
/* vfs_conv_dgrad / vfs_conv_dgrad_bn whose `add` operand is gated by the bit-packed ReLU mask of its own tensor (add_mask: the
 * mask_bits of vfs_bn_act_mask for the [N,H,W,Cin] block output; NULL = plain add; Cin % 64 == 0): dx = dgrad + add * (y > 0).
 * The identity branch of a residual block (resnet.py:105-111,224-230: out = relu(bn(x) + identity)) then adds the
 * block-output gradient itself; the masked copy g * (y > 0) that vfs_bn_bwd_apply can emit (gm) is never written or read. */
int vfs_conv_dgrad_maskadd(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, const uint8_t* add_mask,
                           int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                           vfs_stream_t stream);
int vfs_conv_dgrad_bn_maskadd(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, const uint8_t* add_mask,
                              const vfs_bf16* bn_x, const vfs_bf16* bn_y, const float* bnp, float* bn_partial,
                              int bn_mpg, int bn_relu, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                              int KW, int stride, int pad, vfs_stream_t stream);
/* Split-K variants for problems with few output pixels and a long reduction - the SimSiam head's Linear
 * layers (sim_siam_head.py:78-111: 2048x2048 on 64 rows fill 16 workgroups otherwise): the K loop is cut into
 * ksplit slices that run as separate workgroups; the last slice to finish adds the fp32 partial tiles in
 * slice order (deterministic) and runs the usual epilogue (bias, statistics rows).  ks_ws: float workspace,
 * 1024 + tiles * ksplit * 128 * BC floats with tiles = ceil(M/128) * ceil(Cout/BC) <= 1024, BC = 128 when
 * Cout % 128 == 0 else 64; its first 1024 words (tickets) must be zero before the first launch and are left
 * zero.  Stride-1 problems only for the dgrad; ksplit == 1 is the plain kernel. */
int vfs_conv_fwd_splitk(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats,
                        float* ks_ws, int ksplit, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH,
                        int KW, int stride, int pad, vfs_stream_t stream);
int vfs_conv_dgrad_splitk(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, float* ks_ws,
                          int ksplit, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                          int stride, int pad, vfs_stream_t stream);
/* forward conv with dilated taps (resnet.py:51-58,172-179: ConvModule(padding=dilation, dilation=dilation) of a ResNet built
 * with dilations != 1 - the frozen backbone of the SiamFC probe, projects/siamfc-pytorch/siamfc/default_config_base.py:40-49).
 * Forward only: there is no dgrad / wgrad for dilated layers.  Ho = (H + 2 pad - dilation (KH - 1) - 1) / stride + 1. */
int vfs_conv_fwd_dilated(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, int N,
                         int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                         int dilation, vfs_stream_t stream);
/* conv-BN-ReLU -> conv without materialising the activation: x_raw is the RAW output of the producer unit,
 * in_bnp its float[G][4][Cin] {scale, shift, mean, invstd}, in_npg images per group; relu(x*scale+shift)
 * (rounded to bf16 exactly as vfs_bn_act) is applied while the operand is staged, padding stays zero.
 * 3x3 / stride 1 / pad 1 on halo-tile-eligible shapes, and (round 6: the conv2 -> conv3 edge of a bottleneck block,
 * resnet.py:221-230) 1x1 / stride 1 / pad 0 with Cin % 64 == 0 and groups of whole 128-pixel tiles (in_npg*H*W % 128 == 0);
 * VFS_ERR_SHAPE otherwise.  The matching weight gradient reads the same raw tensor.  Replaces vfs_bn_act + vfs_conv_fwd /
 * vfs_conv_wgrad. */
int vfs_conv_fwd_bnin(const vfs_bf16* x_raw, const float* in_bnp, int in_npg, const vfs_bf16* wf, vfs_bf16* y,
                      const float* bias, float* stats, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                      int KH, int KW, int stride, int pad, vfs_stream_t stream);
int vfs_conv_wgrad_bnin(const vfs_bf16* dy, const vfs_bf16* x_raw, const float* in_bnp, int in_npg, float* partial,
                        float* grad, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                        int stride, int pad, int nsplit, int pix_per_split, vfs_stream_t stream);
/* wgrad: grad[Cout][Cin][KH][KW] (fp32, reference OIHW layout) += sum_pixels dy * im2col(x).
 * partial: workspace float[nsplit][Cout][KH*KW*Cin]; pix_per_split % 64 == 0 and
 * nsplit*pix_per_split >= N*Ho*Wo.  Deterministic (fixed-order split-K reduction).
 * grad == NULL (here and in vfs_conv_wgrad_bnin / vfs_stem_wgrad / vfs_stem_wgrad_fused): only the partials are written;
 * the caller keeps `partial` alive and reduces many layers at once with vfs_wgrad_reduce_table. */
int vfs_conv_wgrad(const vfs_bf16* dy, const vfs_bf16* x, float* partial, float* grad, int N, int H,
                   int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                   int nsplit, int pix_per_split, vfs_stream_t stream);
int vfs_stem_wgrad(const vfs_bf16* dy, const vfs_bf16* x4, float* partial, float* grad, int N, int H,
                   int Wp, int Ho, int Wo, int nsplit, int pix_per_split, vfs_stream_t stream);
/* Round 6 - vfs_conv_wgrad / vfs_conv_wgrad_bnin with the split-K reduction INSIDE the launch (one launch per layer instead of
 * two): the last workgroup of a (k-column, cout) tile to arrive - a device-scope ticket per tile - adds the tile's partials in
 * ascending split order (run-to-run bit-identical, whichever workgroup arrives last) into grad.  Same operands as vfs_conv_wgrad
 * (torch autograd of F.conv2d / F.linear wrt the weight, resnet.py:163-191,221-230; sim_siam_head.py:78-111); in_bnp / in_npg as in
 * vfs_conv_wgrad_bnin or NULL / 0; grad must be given.  partial: workspace of nsplit * Cout * roundup(KH*KW*Cin, 128) floats (the
 * launch's own layout: whole 128-byte lines per wave, written and read back with device-scope accesses; its contents mean nothing
 * to the caller).  tickets: unsigned[vfs_wgrad_tickets()] in device memory, ZERO before the
 * first launch; every launch leaves it zero, so one array serves all launches of a stream (launches that may overlap on different
 * streams need their own).  Gradients equal the two-launch form's up to the fp32 summation order of the splits. */
int vfs_wgrad_tickets(void);
int vfs_conv_wgrad_inl(const vfs_bf16* dy, const vfs_bf16* x, const float* in_bnp, int in_npg, float* partial, float* grad,
                       unsigned* tickets, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad, int nsplit, int pix_per_split, vfs_stream_t stream);
/* db[Cout] += column sums of dy[M][Cout] (linear bias gradient) */
int vfs_bias_grad(const vfs_bf16* dy, float* db, int M, int C, vfs_stream_t stream);

/* ---- BatchNorm / SyncBN + activation (torch batch_norm in ConvModule; configs/r*_*.py:9,15) ------
 * G = number of independent BN batches inside the tensor (the two views), bpg partial blocks
 * per group.  sums: double[G][2][C]; between reduce_partials and finalize the host all-reduces
 * `sums` across ranks for SyncBN (count = global element count per channel per group).
 * bnp: float[G][4][C] = {scale, shift, mean, invstd}. */
int vfs_bn_reduce_partials(const float* partial, double* sums, double* scratch, int G, int bpg, int C,
                           vfs_stream_t stream);
/* scratch (or NULL): 64 uint32 ticket counters, zero before the first use and left at zero by every
 * call, followed by double[G][128][2][C].  With it, large row counts are reduced by many workgroups
 * in ONE launch (the workgroup that draws the last ticket of a 32-channel block finishes it). */
int vfs_bn_finalize(const double* sums, const float* gamma, const float* beta, float* bnp,
                    float* running_mean, float* running_var, int G, int C, double count, float eps,
                    float momentum, vfs_stream_t stream);
/* vfs_bn_stats_finalize + vfs_bn_act in ONE launch, resp. vfs_bn_bwd_sums_paramgrad + vfs_bn_bwd_apply, for SMALL
 * statistics row counts (bpg <= ~128 rows per group: ResNet-50's 16x16 / 8x8 stages): every workgroup of the apply
 * pass reduces the partial rows of its own (group, <= 64-channel slab) in its prologue - same order everywhere, so the
 * coefficients are identical - and the first workgroup of a slab writes bnp / sums / running statistics (forward) or
 * sums / dgamma / dbeta (backward).  Removes a dependent ~6 us launch per BatchNorm layer and direction.
 * vfs_bn_act_fin with partial == NULL (SyncBN): `sums` already holds the all-reduced totals and count the global element
 * count; the kernel then replaces vfs_bn_finalize + vfs_bn_act for any tensor size. */
int vfs_bn_act_fin(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp,
                   double* sums, float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres,
                   const float* rbnp, vfs_bf16* y, long long M, int C, int mpg, int relu, double count, float eps,
                   float momentum, vfs_stream_t stream);
int vfs_bn_bwd_apply_fin(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, const float* partial,
                         int bpg, double* sums, float* dgamma, float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M, int C,
                         int mpg, double count, int relu, vfs_stream_t stream);
/* the same result as vfs_bn_stats_finalize, computed from the stored bf16 output raw [G*rows_per_group][C] instead of the conv
 * kernels' 128-pixel statistics rows: for SMALL groups that are not multiples of 128 rows (the head's BN1d layers,
 * sim_siam_head.py:78-111, 32 rows per view on the ResNet-50 config), so that ONE conv launch covers all groups */
/* Round 6 - nn.Linear + BatchNorm1d (training statistics) + [ReLU] of the SimSiam head in ONE launch (sim_siam_head.py:78-111: the
 * projector / predictor units Linear -> BN -> ReLU on M = G * mpg <= 256 rows, G <= 4 views, K % 128 == 0, C % 16 == 0):
 * raw = x wf^T + bias (bf16, kept for the backward), statistics of the stored values per view, act = [relu](raw * scale + shift),
 * bnp / sums / running statistics as vfs_bn_stats_raw_finalize writes them - for <= 128 rows (where vfs_conv_fwd runs the skinny GEMM)
 * the same bits as vfs_conv_fwd + vfs_bn_stats_raw_finalize + vfs_bn_act, above that equal up to the K order of the GEMM; two
 * dependent launches less per unit.  Single-GPU path (SyncBN keeps the exchange). */
int vfs_linear_bn_act(const vfs_bf16* x, const vfs_bf16* wf, const float* bias, const float* gamma, const float* beta,
                      vfs_bf16* raw, vfs_bf16* act, float* bnp, double* sums, float* running_mean, float* running_var, int M,
                      int K, int C, int mpg, int relu, double count, float eps, float momentum, vfs_stream_t stream);
/* Round 6 - vfs_bn_bwd_reduce + vfs_bn_bwd_apply_fin in ONE launch for groups with a single statistics row (mpg | 512, or mpg < 16:
 * the BatchNorm1d layers of the SimSiam head, sim_siam_head.py:78-111, and tiny maps): the apply pass computes the row {sum g*mask,
 * sum g*mask*xhat} from (g, x, mask) in its prologue in vfs_bn_bwd_reduce's summation order - the same bits - and writes sums,
 * dgamma, dbeta, dx (gm) as vfs_bn_bwd_apply_fin does.  torch autograd of batch_norm (+ relu) on a [M][C] tensor. */
int vfs_bn_bwd_apply_raw(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, double* sums, float* dgamma,
                         float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg, double count, int relu,
                         vfs_stream_t stream);
int vfs_bn_stats_raw_finalize(const vfs_bf16* raw, double* sums, const float* gamma, const float* beta, float* bnp,
                              float* running_mean, float* running_var, int G, int rows_per_group, int C, double count,
                              float eps, float momentum, vfs_stream_t stream);
/* single-process fast paths (no SyncBN all-reduce in between): vfs_bn_reduce_partials fused with
 * vfs_bn_finalize, resp. with vfs_bn_param_grad (sums are still written for the apply pass) */
int vfs_bn_stats_finalize(const float* partial, double* sums, double* scratch, const float* gamma,
                          const float* beta, float* bnp, float* running_mean, float* running_var, int G,
                          int bpg, int C, double count, float eps, float momentum, vfs_stream_t stream);
int vfs_bn_bwd_sums_paramgrad(const float* partial, double* sums, double* scratch, float* dgamma,
                              float* dbeta, int G, int bpg, int C, vfs_stream_t stream);
int vfs_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float* bnp, int C, float eps, vfs_stream_t stream);
/* y = [relu](x*scale+shift [+res] [+ rres*rscale+rshift])   (resnet.py:102-111,221-230) */
int vfs_bn_act(const vfs_bf16* x, const float* bnp, const vfs_bf16* res, const vfs_bf16* rres,
               const float* rbnp, vfs_bf16* y, long long M, int C, int mpg, int relu,
               vfs_stream_t stream);
/* vfs_bn_act / vfs_bn_act_fin with a second output: mask_bits, M*C/8 bytes, bit i of the byte of (pixel m, channels c..c+7) =
 * (y[m][c+i] > 0); bytes are stored SLAB-major, uint8 [C/64][M][8] ([M][C/8] when C < 64), so that a kernel that owns <= 64
 * channels of a pixel range reads contiguous bytes.
 * The BatchNorm backward of a residual unit (resnet.py:102-111,221-230: out = relu(bn(x) + identity)) needs y only as
 * that mask, in two passes (statistics, apply): pass the mask as `y` with relu = 2 to vfs_bn_bwd_reduce / vfs_bn_bwd_apply /
 * vfs_bn_bwd_apply_fin / vfs_conv_dgrad_bn (bn_relu = 2) and they read 1/16 of the bytes - bit-identical gradients. */
int vfs_bn_act_mask(const vfs_bf16* x, const float* bnp, const vfs_bf16* res, const vfs_bf16* rres,
                    const float* rbnp, vfs_bf16* y, uint8_t* mask_bits, long long M, int C, int mpg, int relu,
                    vfs_stream_t stream);
int vfs_bn_act_fin_mask(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp,
                        double* sums, float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres,
                        const float* rbnp, vfs_bf16* y, uint8_t* mask_bits, long long M, int C, int mpg, int relu,
                        double count, float eps, float momentum, vfs_stream_t stream);
/* stem: y = maxpool3x3/2/1(relu(bn(x)))  (resnet.py:435), idx = argmax code per element (tap 0..8 of the 3x3 window; 0xFF where
 * the pooled activation is not positive - no gradient flows there, and the kernels that route the gradient by the code
 * (vfs_maxpool_relu_bwd, vfs_stem_pool_bn_bwd_reduce with xpool, vfs_stem_pool_bn_bwd_apply, vfs_stem_wgrad_fused) take the ReLU mask
 * from it: their yp argument is kept for the signature, they do not read it); xpool (optional) = the RAW x at each argmax: the BatchNorm backward of the stem then
 * never re-reads x */
int vfs_bn_relu_maxpool(const vfs_bf16* x, const float* bnp, vfs_bf16* y, uint8_t* idx, vfs_bf16* xpool,
                        int N, int H, int W, int C, int Hp, int Wp, int npg, vfs_stream_t stream);
int vfs_maxpool_relu_bwd(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx, vfs_bf16* ga,
                         int N, int H, int W, int C, int Hp, int Wp, vfs_stream_t stream);
/* BN backward: pass 1 -> partial[nblk][2][C] (S1 = sum gm, S2 = sum gm*xhat, gm = g*mask);
 * then vfs_bn_reduce_partials -> sums (all-reduce for SyncBN) -> pass 2.
 * mask: y != NULL -> (y > 0) (residual units); y == NULL && relu -> (x*scale+shift > 0); else 1 */
int vfs_bn_bwd_reduce(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp,
                      float* partial, long long M, int C, int mpg, int ppb, int relu,
                      vfs_stream_t stream);
int vfs_bn_bwd_apply(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp,
                     const double* sums, vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg,
                     double count, int relu, vfs_stream_t stream);
int vfs_bn_param_grad(const double* sums, float* dgamma, float* dbeta, int G, int C,
                      vfs_stream_t stream);
/* stem: BN backward through max-pool + ReLU without materialising the full-resolution gradient
 * (same results as vfs_maxpool_relu_bwd followed by vfs_bn_bwd_reduce / vfs_bn_bwd_apply);
 * pass 1 -> partial[ceil(N*Hp*Wp/ppb)][2][C], pass 2 -> dx[N][H][W][C] */
int vfs_stem_pool_bn_bwd_reduce(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx,
                                const vfs_bf16* x, const vfs_bf16* xpool, const float* bnp, float* partial,
                                int N, int H, int W, int C, int Hp, int Wp, int npg, int ppb,
                                vfs_stream_t stream); /* xpool (from vfs_bn_relu_maxpool) replaces the gather from x */
/* stem weight gradient with the BN-backward apply pass folded into its operand load (the
 * full-resolution dx is never materialised): grad[64][3][7][7] += ...; partial: float[nblocks][64][224],
 * nblocks = ceil(ntiles / ceil(ntiles / nblocks)) with ntiles = N*ceil(Ho/8)*ceil(Wo/16) */
int vfs_stem_wgrad_fused(const vfs_bf16* x4, const vfs_bf16* xraw, const vfs_bf16* gp, const vfs_bf16* yp,
                         const uint8_t* idx, const float* bnp, const double* sums, float* partial,
                         float* grad, int N, int Hin, int Win, int Ho, int Wo, int Hp, int Wp, int npg,
                         double count, int nblocks, vfs_stream_t stream);
int vfs_stem_pool_bn_bwd_apply(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx,
                               const vfs_bf16* x, const float* bnp, const double* sums, vfs_bf16* dx,
                               int N, int H, int W, int C, int Hp, int Wp, int npg, double count,
                               vfs_stream_t stream);

/* ---- head: AdaptiveAvgPool2d((1,1)) + flatten (sim_siam_head.py:117,154-157) ---------------- */
int vfs_avgpool_fwd(const vfs_bf16* x, vfs_bf16* y, int N, int HW, int C, vfs_stream_t stream);
int vfs_avgpool_bwd(const vfs_bf16* g, vfs_bf16* gx, int N, int HW, int C, vfs_stream_t stream);

/* ---- CosineSimLoss over all temporal rolls (sim_loss.py:42-63, sim_siam_head.py:165-174,
 * sim_siam_base_tracker.py:31-56).  loss/gloss: float[K][N]; K = T (intra_video) or 1 */
int vfs_cosine_loss_fwd(const vfs_bf16* p1, const vfs_bf16* z1, const vfs_bf16* p2,
                        const vfs_bf16* z2, float* loss, int N, int C, int T, int K, int negative,
                        float weight, vfs_stream_t stream);
/* BaseTracker._parse_losses (trackers/base.py:76-110) of the K loss rows: means[k] = mean(loss[k][:]) for k < K,
 * means[K] = sum of the K means (the 'loss' entry); double accumulation, fixed order.  means: float[K+1] */
int vfs_loss_means(const float* loss, float* means, int K, int N, vfs_stream_t stream);
int vfs_cosine_loss_bwd(const vfs_bf16* p1, const vfs_bf16* z1, const vfs_bf16* p2,
                        const vfs_bf16* z2, const float* gloss, vfs_bf16* dp1, vfs_bf16* dp2, int N,
                        int C, int T, int K, int negative, float weight, vfs_stream_t stream);

/* ---- CosineSimLoss on spatial inputs, incl. pairwise=True / mask / with_norm=False (sim_loss.py:42-63) -------------
 * fp32 [B][C][S] operands = the reference's [B,C,*] tensors flattened (`.flatten(2)`); Sa / Sl: positions of cls_score / label.
 * colnorm: inv[b][s] = 1 / max(||x[b][:][s]||_2, 1e-12)   (F.normalize(p=2, dim=1), :44-45); pass inv = NULL for with_norm=False.
 * fwd: pairwise = 1: prod = einsum('bci,bcj->bij') on the matrix cores (fp32 MFMA), * mask[B][Sa][Sl] when given, mean over
 *      (i, j) (:48-56); pairwise = 0 (Sa == Sl): sum over C per position, mean over positions (:57-58);
 *      loss[b] = weight * (negative ? -mean : 2 - 2 * mean) (:59-62, losses/base.py:37).  partial: float scratch
 *      [B][ceil(Sa/32) * ceil(Sl/32)].
 * bwd: d[B][C][Sself] = gradient wrt the NORMALISED operand of one side given gloss[B]; `other` is the opposite operand
 *      (raw) with its inverse norms; mask_transposed = 1 when this side is the einsum's j operand (label).
 * norm_bwd: dx = (d - x^ <x^, d>) * inv per position (backward of F.normalize); inv = NULL: dx = d. */
int vfs_simloss_colnorm(const float* x, float* inv, int B, int C, int S, vfs_stream_t stream);
int vfs_simloss_fwd(const float* a, const float* l, const float* inva, const float* invl, const float* mask,
                    float* partial, float* loss, int B, int C, int Sa, int Sl, int pairwise, int negative,
                    float weight, vfs_stream_t stream);
int vfs_simloss_bwd(const float* other, const float* invo, const float* mask, int mask_transposed,
                    const float* gloss, float* d, int B, int C, int Sself, int Sother, int pairwise,
                    int negative, float weight, vfs_stream_t stream);
int vfs_simloss_norm_bwd(const float* x, const float* inv, const float* d, float* dx, int B, int C, int S,
                         vfs_stream_t stream);

/* ---- SyncBN statistic exchange over xGMI (configs/r*_*.py:9,15 SyncBN under MMDistributedDataParallel, apis/train.py:58-66):
 * a latency-bound all-reduce of <= 8192 doubles among the <= 8 ranks of one node as ONE small kernel - peer stores into
 * IPC-mapped windows + epoch flags (csrc/p2p.hip) - instead of a collective-library call per BatchNorm layer and direction.
 *   window_bytes: size of a window, the largest n and world supported
 *   alloc / free: a zeroed window in fine-grained device memory (hipExtMallocWithFlags)
 *   export: 64-byte IPC handle of the own window (hipIpcGetMemHandle); import / unimport: map / unmap a peer's window
 *   allreduce_f64: buf[0..n) <- sum over ranks, in rank order (bit-identical on every rank).  peers: DEVICE array of `world`
 *     window pointers (peers[rank] = the own window); state: 4 x uint64 in device memory, zero-initialised ({exchange
 *     counter, error flag, -, -}); phase 3 = push + wait (1 / 2: the halves, for protocol tests); spin_limit: polls before the
 *     kernel gives up, sets state[1] = 1 and returns garbage (a lost peer must not hang the GPU). */
int vfs_p2p_window_bytes(long long* bytes, int* max_doubles, int* max_world);
/* the SyncBN reductions with the exchange as their TAIL: vfs_bn_reduce_partials (forward: statistics rows -> sums) and
 * vfs_bn_bwd_sums_paramgrad (backward: rows -> sums + LOCAL dgamma / dbeta) whose last workgroup runs the window exchange on
 * sums[G][2][C] (G*2*C <= 8192) - no launch and no collective call between "local sums" and "sums over all ranks".
 * state: 4 x uint64, zero-initialised ({exchange counter, error flag, workgroup ticket, -}); other arguments as above. */
int vfs_bn_reduce_partials_xchg(const float* partial, double* sums, double* scratch, int G, int bpg, int C,
                                const void* peers, int rank, int world, void* state, long long spin_limit,
                                vfs_stream_t stream);
int vfs_bn_bwd_sums_paramgrad_xchg(const float* partial, double* sums, double* scratch, float* dgamma,
                                   float* dbeta, int G, int bpg, int C, const void* peers, int rank, int world,
                                   void* state, long long spin_limit, vfs_stream_t stream);
/* Round 6 - the exchange FOLDED into the apply passes of vfs_bn_act_fin_mask / vfs_bn_bwd_apply_fin (small row counts: the 16x16 /
 * 8x8 stages): `partial` holds the LOCAL statistics rows; the first workgroup of every 64-channel slab sums them, exchanges the
 * slab's G*2*64 sums with the peers' workgroups of the same slab through the windows and releases the slab's other workgroups, which
 * wait on a device-scope word instead of summing rows.  sums / bnp / running statistics (forward) and sums (backward) come from the
 * totals over the ranks, count = GLOBAL element count; dgamma / dbeta stay local sums.  Replaces vfs_bn_reduce_partials_xchg +
 * vfs_bn_act_fin (forward) and vfs_bn_bwd_sums_paramgrad_xchg + vfs_bn_bwd_apply (backward): one dependent launch less per BatchNorm
 * layer and direction - the SyncBN step costs what the single-GPU step costs plus the exchange latency.  G*2*min(C,64) <= 256
 * (two views), C <= 4096, G*2*C <= 8192.  state: (4 + 64) x uint64 in device memory, zero-initialised ({exchange counter, error
 * flag, workgroup ticket, chain counter, slab_ready[64]}); peers / rank / world / spin_limit as vfs_bn_reduce_partials_xchg.
 * seq (0 .. 4094) numbers the folded exchanges of one launch chain: the exchange's epoch is state[3] * 4096 + seq + 1, and every rank
 * must issue the same (chain, seq) sequence - vfs_p2p_chain_start(state) (one single-thread launch: state[3] += 1) goes to the head
 * of every chain that is recorded once and replayed (the recorded seq values repeat, the chain counter does not); a caller that
 * launches eagerly may simply count seq up and never start a chain.  Same torch call sites: SyncBatchNorm forward / backward
 * (configs/r*_*.py:9,15; apis/train.py:58-66). */
int vfs_p2p_chain_start(void* state, vfs_stream_t stream);
int vfs_bn_act_fin_xchg(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp,
                        double* sums, float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres,
                        const float* rbnp, vfs_bf16* y, uint8_t* mask_bits, long long M, int C, int mpg, int relu,
                        double count, float eps, float momentum, const void* peers, int rank, int world, void* state,
                        long long spin_limit, int seq, vfs_stream_t stream);
int vfs_bn_bwd_apply_fin_xchg(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, const float* partial,
                              int bpg, double* sums, float* dgamma, float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M,
                              int C, int mpg, double count, int relu, const void* peers, int rank, int world, void* state,
                              long long spin_limit, int seq, vfs_stream_t stream);
int vfs_p2p_alloc(void** window);
int vfs_p2p_free(void* window);
int vfs_p2p_export(void* window, void* handle64);
int vfs_p2p_import(const void* handle64, void** window);
int vfs_p2p_unimport(void* window);
int vfs_p2p_allreduce_f64(double* buf, int n, const void* peers, int rank, int world, void* state, int phase,
                          long long spin_limit, vfs_stream_t stream);

/* ---- optimizer: torch.optim.SGD(lr, momentum, weight_decay) (configs/r*_*.py:134) on flat arenas --- */
/* skip_flag: NULL, or a device word (8 bytes) read when the kernel RUNS: non-zero -> the step changes nothing.  The trackers pass
 * the error word of the SyncBN window exchange (vfs_p2p_allreduce_f64's state[1]): a step whose statistics were poisoned by a
 * peer that never arrived must not reach the weights or the momentum (the host learns of it with the step's log values). */
int vfs_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, float lr,
                 float momentum, float weight_decay, const void* skip_flag, vfs_stream_t stream);
int vfs_scale(float* x, long long n, float scale, vfs_stream_t stream);
/* bf16 gradient buckets for the data-parallel all-reduce (opt-in, VFS_GRAD_BF16=1; the reference's DDP - apis/train.py:62-66 -
 * reduces fp32): dst = bf16(src * scale) before the collective, dst = float(src) after it; buffers 16-byte aligned */
int vfs_f32_to_bf16(const float* src, vfs_bf16* dst, long long n, float scale, vfs_stream_t stream);
int vfs_bf16_to_f32(const vfs_bf16* src, float* dst, long long n, vfs_stream_t stream);

/* ---- DAVIS label propagation (VanillaTracker.forward_test, vanilla_tracker.py:80-206) -------
 * F.normalize(dim=channel) of NHWC rows, once per frame when it enters the bank
 * (local_attention.py:277-279 does it on every propagation step) */
int vfs_l2norm_rows(const vfs_bf16* x, vfs_bf16* y, long long P, int C, vfs_stream_t stream);
/* masked_attention_efficient (local_attention.py:237-348) + spatial_neighbor 'circle'
 * (affinity_utils.py:144-156): fbank [frames][H*W][C] normalised bf16, sbank [frames][H*W][CO]
 * fp32; key frames kslot[0..nkeys) in the reference's order (first frame first, duplicates
 * allowed); out [H*W][CO].  radius = neighbor_range // 2 (<= 0: no mask), the first non_mask_len
 * key frames are never masked (test_cfg.with_first_neighbor=False -> 1), topk <= 10, nkeys <= 64,
 * C % 64 == 0.  workspace: at least vfs_labelprop_workspace_bytes(H, W) bytes (per-split partial top-k lists - key frames,
 * and for vfs_labelprop_f32 the 64-key blocks of a frame's window as well, are split over workgroups because a DAVIS
 * frame has only 8x14 query tiles); workspace_bytes = what the caller allocated: a smaller buffer is REFUSED
 * (VFS_ERR_ARG) instead of written past (the size grew between library versions - ABI version 2) */
int vfs_labelprop_workspace_bytes(int H, int W, long long* bytes);
int vfs_labelprop(const vfs_bf16* fbank, const float* sbank, float* out, void* workspace, long long workspace_bytes, int qframe,
                  const int* kslot, int nkeys, int H, int W, int C, int CO, int radius, int non_mask_len,
                  int topk, float temperature, vfs_stream_t stream);
/* bilinear upsample (align_corners=False) + per-channel min-max normalisation where max > 0 +
 * argmax -> uint8 [Ho][Wo] (vanilla_tracker.py:162-181); partial: workspace float[64*CO*2] */
int vfs_seg_postprocess(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho,
                        int Wo, vfs_stream_t stream);
/* F.one_hot of the resized first-frame label map into the seg bank (vanilla_tracker.py:96-100) */
int vfs_onehot(const uint8_t* labels, float* out, int P, int CO, vfs_stream_t stream);

/* split-K partials of MANY layers -> their gradients in ONE launch.  desc: device array of nrecords 56-byte records
 * {const float* partial; float* grad; int nsplit, Cout, Ktot, Cin, KH, KW, stem, block_start, nblocks, pad;} (stem = 1:
 * the 7x7 stem's k-layout, Ktot 256 for vfs_stem_wgrad / 224 for vfs_stem_wgrad_fused with Cin = 3); a record is served
 * by workgroups [block_start, block_start + nblocks), 128 elements each per pass; total_blocks = sum of nblocks.
 * grad += sum over the splits in a fixed order (deterministic). */
int vfs_wgrad_reduce_table(const void* desc, int nrecords, int total_blocks, vfs_stream_t stream);

/* ---- fp32 evaluation path ("exact" precision; csrc/exact_f32.hip) ------------------------------------
 * The reference evaluates in fp32 and its outputs are INTEGER label maps, so the default forward_test path stores
 * and computes in fp32 with bit-defined arithmetic: every dot product is one ascending chain acc = fma(a_k, b_k, acc)
 * on v_mfma_f32_32x32x2_f32, every other step a single correctly rounded fp32 operation, exp() an explicit
 * polynomial, top-k ties to the lowest candidate index.  oracle/exact_oracle.c states the same arithmetic in C and
 * the results agree bit for bit (tests/test_exact_f32.py); the bf16 entry points above remain as the fast mode.
 *
 * mmcv ConvModule in eval mode (conv -> BN(running statistics) [-> + identity] [-> ReLU]; resnet.py:51-73,102-111,
 * 163-191,221-230): y[N,Ho,Wo,Cout] = [relu]( fma(conv(x, w), scale, shift) [+ res] ), fp32 NHWC, w [Cout][KH][KW][Cin]
 * (the chain runs kh, kw, cin ascending), Cin % 4 == 0 (3-channel frames as NHWC4 with zero weights on channel 3),
 * scale/shift [Cout] or both NULL, any stride / padding / dilation */
int vfs_conv_f32_fwd(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y,
                     int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                     int dilation, int relu, vfs_stream_t stream);
/* imgs fp32 [B][V][3][T][H][W] -> fp32 NHWC4 frames out[(v*B+b)*T+t][h][w][4], channel 3 = 0 (common/utils.py:45-53) */
int vfs_imgs_to_nhwc4_f32(const float* imgs, float* out, int B, int V, int T, int H, int W, vfs_stream_t stream);
/* nn.MaxPool2d(3, 2, 1) (resnet.py:435), fp32 NHWC, C % 4 == 0 */
int vfs_maxpool_f32(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, vfs_stream_t stream);
/* F.normalize(dim=channel, eps=1e-12) of fp32 rows [P][C] (local_attention.py:277-279), C % 4 == 0 */
int vfs_l2norm_rows_f32(const float* x, float* y, long long P, int C, vfs_stream_t stream);
/* vfs_labelprop on an fp32 bank: score = chain(key . query) / temperature, top-k by (score desc, candidate id asc) with
 * id = key_position * H*W + pixel, softmax weights exp(s - s_max) / sum in sorted order; the first non_mask_len key
 * frames are not masked (test_cfg.with_first_neighbor=False -> 1, local_attention.py:303-309); workspace as vfs_labelprop */
int vfs_labelprop_f32(const float* fbank, const float* sbank, float* out, void* workspace, long long workspace_bytes, int qframe,
                      const int* kslot, int nkeys, int H, int W, int C, int CO, int radius, int non_mask_len, int topk,
                      float temperature, vfs_stream_t stream);
/* Two-pass form of vfs_labelprop_f32 with IDENTICAL results (csrc/labelprop2.hip): every in-window candidate is first scored on the
 * bf16 matrix path from a split copy of the bank (x = hi + lo, three products; |s~ - s| <= 3 * 2^-16 + 4 * C * 2^-24 for unit
 * rows), the candidates within twice that bound of the query's 10th-best s~ - the only ones that can be in the exact top 10 -
 * are rescored with the defining fp32 chain and go through the same total order / softmax / value sum.
 *   vfs_split_rows_bf16x2: hl [P][C/16][16 hi | 16 lo] bf16 from unit rows x [P][C] fp32 (C % 16 == 0), once per frame;
 *   vfs_labelprop_f32_2pass: hlbank = the split copy of fbank (same frame indexing).  unit_rows = 0 (test_cfg.with_norm=False),
 *   hlbank = NULL or C not in {256, 512, 1024} -> the dense kernel; a candidate list that overflows -> the dense kernel redoes
 *   the frame, decided on the device.  workspace: vfs_labelprop_f32_2pass_workspace_bytes(H, W) for the full list capacity (4608
 *   entries per query: 237 MB at 60 x 107), or vfs_labelprop_f32_2pass_workspace_bytes_for(H, W, entries_per_query >= 16) - the launch
 *   derives the capacity from workspace_bytes; fewer entries trade memory for dense redos of the first frames of a clip. */
int vfs_split_rows_bf16x2(const float* x, vfs_bf16* hl, long long P, int C, vfs_stream_t stream);
int vfs_labelprop_f32_2pass_workspace_bytes(int H, int W, long long* bytes);
int vfs_labelprop_f32_2pass_workspace_bytes_for(int H, int W, int entries_per_query, long long* bytes);
int vfs_labelprop_f32_2pass(const float* fbank, const vfs_bf16* hlbank, const float* sbank, float* out, void* workspace,
                            long long workspace_bytes, int qframe, const int* kslot, int nkeys, int H, int W, int C, int CO,
                            int radius, int non_mask_len, int topk, float temperature, int unit_rows, vfs_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=False) of a C-channel fp32 map between layouts (NCHW [C][H][W] or NHWC
 * [H][W][C], chosen per side): one-hot reference maps -> feature resolution, soft label maps -> original resolution
 * (vanilla_tracker.py:101-111,162-166 when ref_seg_map is 4-D) */
int vfs_bilinear_resize_f32(const float* src, float* dst, int C, int H, int W, int Ho, int Wo, int src_nhwc, int dst_nhwc,
                            vfs_stream_t stream);
/* vfs_seg_postprocess with every step a single fp32 operation (no contraction) */
int vfs_seg_postprocess_exact(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                              vfs_stream_t stream);

/* ---- DAVIS-2017 semi-supervised J&F (datasets/davis_dataset.py:68-140 -> davis2017.evaluation) -----
 * pred / gt: uint8 label maps [T][H][W]; frames 1..T-2 are evaluated (first = given, last excluded);
 * objects 1..nobj (<= 32), gt label 255 = void when use_void; radius = ceil(0.008*||(H,W)||) of the
 * boundary-match disk.  counts: int32 [T-2][nobj][6] = {|pred&gt|, |pred|gt|, #pred boundary,
 * #gt boundary, matched pred boundary, matched gt boundary}; scratch: uint32 [2][T-2][H][W].
 * J = c0/c1 (1 if c1 == 0); F from c2..c5 with the package's empty-boundary conventions. */
int vfs_davis_counts(const uint8_t* pred, const uint8_t* gt, int* counts, void* scratch, int T, int H, int W,
                     int nobj, int radius, int use_void, vfs_stream_t stream);

/* ---- training input pipeline (configs/r*_*.py:48-91: RandomResizedCrop -> Resize -> Flip -> Normalize ->
 * FormatShape NCTHW; pipelines/augmentations.py:171-334,487-596,600-707,711-794) in one pass over the decoded
 * frames.  src uint8 [B*V*T][Hs][Ws][3] RGB in pipeline order (b, v, t); boxes int32 [F][4] = left, top,
 * right, bottom; flips uint8 [F]; outputs (either may be NULL): imgs fp32 [B][V][3][T][Ho][Wo] (what
 * train_step takes), x4 bf16 NHWC4 [V*B*T][Ho][Wp][4] (what vfs_stem_fwd reads).  Bilinear = cv2.resize
 * INTER_LINEAR on 8-bit data (fixed point), normalisation = mmcv.imnormalize_ (double arithmetic). */
int vfs_crop_resize_flip_norm(const uint8_t* src, const int* boxes, const uint8_t* flips, float* imgs, vfs_bf16* x4,
                              int B, int V, int T, int Hs, int Ws, int Ho, int Wo, int Wp, double mean_r,
                              double mean_g, double mean_b, double std_r, double std_g, double std_b,
                              vfs_stream_t stream);

/* ---- SiamFC probe head (projects/siamfc-pytorch/siamfc/heads.py:16-23,51-58 `_fast_xcorr`): response maps
 * out[m][i][j] = scale * sum_{u,v,c} z[m % nz][u][v][c] * x[m][i+u][j+v][c]; z bf16 [nz][Hz][Wz][C], x bf16 [nx][H][W][C] (NHWC),
 * out fp32 [nx][H-Hz+1][W-Wz+1]; nx % nz == 0, C % 8 == 0.  Forward only (inference of a trained probe). */
int vfs_xcorr_fwd(const vfs_bf16* z, const vfs_bf16* x, float* out, int nz, int nx, int Hz, int Wz, int H, int W, int C,
                  float scale, vfs_stream_t stream);

/* training the probe (siamfc_tracker_base.py:364-387 `train_step`): backward of the cross-correlation.  g fp32
 * [nx][H-Hz+1][W-Wz+1] = dL/d(responses); dz bf16 [nz][Hz][Wz][C] (summed over the search features that share an exemplar),
 * dx bf16 [nx][H][W][C]; either may be NULL.  The gradients feed vfs_conv_wgrad / vfs_bias_grad of the 1x1 convs. */
int vfs_xcorr_bwd(const vfs_bf16* z, const vfs_bf16* x, const float* g, vfs_bf16* dz, vfs_bf16* dx, int nz, int nx, int Hz, int Wz,
                  int H, int W, int C, float scale, vfs_stream_t stream);
/* the probe's losses on n response values (projects/siamfc-pytorch/siamfc/losses.py): mode 0 BalancedLoss (:24-41, param =
 * neg_weight), mode 1 FocalLoss (:44-65, param = gamma; the normaliser mean(avg_weight) is differentiated through, as autograd
 * does).  loss[0] = value; grad (may be NULL) = scale * dL/d(responses). */
int vfs_siamfc_loss(const float* responses, const float* labels, float* loss, float* grad, int n, int mode, float param, float scale,
                    vfs_stream_t stream);
/* torch.optim.Adam (amsgrad off; default_config_base.py:33, siamfc_tracker_base.py:139-145) on flat fp32 arrays, step >= 1 */
int vfs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, vfs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

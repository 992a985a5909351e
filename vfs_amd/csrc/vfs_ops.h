// Argument structs + host launchers of the element-wise / reduction kernels (bn.hip, misc.hip).
#pragma once
#include "vfs_common.h"
#include "vfs_p2p.h"

// y = [relu]( x*scale + shift  [+ res]  [+ rres*rscale + rshift] )
struct BnActArgs {
  const bf16_t* x;      // [M][C] raw conv output
  const float* bnp;     // [G][4][C]
  const bf16_t* res;    // optional materialised residual [M][C]
  const bf16_t* rres;   // optional RAW residual (downsample conv output) ...
  const float* rbnp;    // ... with its own BN parameters [G][4][C]
  bf16_t* y;            // [M][C]
  long long M;
  int C, mpg, relu;     // mpg = pixels per group
  unsigned char* mbits = nullptr;   // optional second output: the bit-packed mask y > 0, [M][C/8] (vfs_common.h mask8_of)
  int wide = 0;         // plain (non-FIN) launch: whole pixel rows per workgroup (set by the launcher, bn.hip slab_geom)
};

// running = (1 - momentum) * running + momentum * statistic; a NaN statistic (the poison of a failed SyncBN exchange) is kept out
__device__ __forceinline__ void bn_running_update(float& rm, float& rv, float momentum, double mean, double unbiased) {
  if (mean != mean || unbiased != unbiased) return;
  const float keep = 1.f - momentum;
  rm = __builtin_fmaf(momentum, (float)mean, keep * rm);
  rv = __builtin_fmaf(momentum, (float)unbiased, keep * rv);
}


// Optional in-kernel statistics finalisation for the apply passes (bn_act / bn_bwd_apply, SMALL row counts): the
// workgroup reduces the partial rows of its own (group, channel slab) in its prologue - every workgroup gets the same
// coefficients, in the same summation order - instead of waiting for a separate 6-us launch between the conv and the
// apply pass.  The leader workgroup of a slab (blockIdx.x == 0) reduces every group and writes the outputs.
struct BnFin {
  const float* partial = nullptr;   // [G][bpg][2][C] rows (forward: sum x, sum x^2; backward: S1, S2)
  int bpg = 0, G = 0;
  // forward outputs
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* bnp = nullptr;             // [G][4][C]
  float* running_mean = nullptr;
  float* running_var = nullptr;
  float eps = 0.f, momentum = 0.f;
  double count = 1.0;
  // both
  double* sums = nullptr;           // [G][2][C]
  // backward outputs (accumulated)
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  // SyncBN, round 6: the window exchange folded into this launch (vfs_p2p.h, "exchange FOLDED"): x.peers set -> `partial` holds the
  // LOCAL statistics rows, the slab leads exchange their sums, `sums` and every coefficient come from the sums over the ranks and
  // `count` is the global element count; dgamma / dbeta stay local sums (they travel with the gradient buckets)
  P2PTail x;
};

// stem: y = maxpool3x3/s2/p1( relu( x*scale + shift ) ), argmax position (0..8, first maximum in
// scan order as torch's CPU max_pool2d) saved per element for the backward pass
struct BnPoolArgs {
  const bf16_t* x;   // [N][H][W][C]
  const float* bnp;  // [G][4][C]
  bf16_t* y;         // [N][Hp][Wp][C]
  uint8_t* idx;      // [N][Hp][Wp][C] or null
  bf16_t* xpool;     // [N][Hp][Wp][C] or null: raw x at the argmax
  int N, H, W, C, Hp, Wp, npg;  // npg = images per group
};

// backward of the stem max-pool (+ReLU): ga[n][h][w][c] = sum over the <=4 windows containing
// (h,w) whose argmax is (h,w) of gp[window] * (yp[window] > 0); gather form, no atomics
struct PoolBwdArgs {
  const bf16_t* gp;    // [N][Hp][Wp][C] gradient wrt pooled output
  const bf16_t* yp;    // pooled output (ReLU mask: pooled max > 0)
  const uint8_t* idx;  // argmax positions
  bf16_t* ga;          // [N][H][W][C]
  int N, H, W, C, Hp, Wp;
};

// ------------------------------------------------------------------------------------------
// BN backward, pass 1: per-channel  S1 = sum gm,  S2 = sum gm * xhat   with gm = g * (y > 0)
// one partial[2][C] per workgroup (PIX_PER_BLOCK pixels of one group)
struct BnBwdArgs {
  const bf16_t* g;     // [M][C] gradient wrt the unit's output
  const bf16_t* y;     // [M][C] unit output for the ReLU mask, or null (mask from x when relu, else none); relu == VFS_MASK_BITS: uint8 [M][C/8] bit mask
  const bf16_t* x;     // [M][C] raw conv output
  const float* bnp;    // [G][4][C]
  const double* sums;  // pass 2: [G][2][C] (S1,S2), all-reduced for SyncBN
  float* partial;      // pass 1 output [nblk][2][C]
  bf16_t* dx;          // pass 2 output [M][C]
  bf16_t* gm;          // pass 2 optional output: masked gradient (identity branch)
  long long M;
  int C, mpg, ppb;     // pixels per group, pixels per block (pass 1; mpg % ppb == 0)
  int relu;            // with y == null: recompute the ReLU mask as x*scale+shift > 0; VFS_MASK_BITS (2): y is the bit-packed mask
  double count;        // pass 2: elements per channel per group (global for SyncBN)
  int wide = 0;        // pass 2, plain (non-FIN) launch: whole pixel rows per workgroup (set by the launcher)
};

#define VFS_BN_MAX_CHUNKS 128
#define VFS_BN_TICKETS 64     // 32-bit ticket counters at the head of the BatchNorm reduction scratch (one per 32 channels)
// stem: BN backward through max-pool + ReLU (bn.hip)
struct StemBwdArgs {
  const bf16_t* gp;    // [N][Hp][Wp][C] gradient wrt the pooled output
  const bf16_t* yp;    // pooled output (ReLU mask)
  const uint8_t* idx;  // argmax codes
  const bf16_t* x;     // [N][H][W][C] raw stem conv output
  const bf16_t* xp;    // pass 1, optional: [N][Hp][Wp][C] raw x at the argmax (replaces the gather from x)
  const float* bnp;    // [G][4][C]
  const double* sums;  // pass 2
  float* partial;      // pass 1: [nblk][2][C]
  bf16_t* dx;          // pass 2: [N][H][W][C]
  int N, H, W, C, Hp, Wp, npg, ppb;   // images per group, pooled pixels per block (pass 1)
  double count;
};

int vfs_stem_pool_bn_bwd_reduce_launch(const StemBwdArgs& a, int nblk, hipStream_t s);
int vfs_stem_pool_bn_bwd_apply_launch(const StemBwdArgs& a, hipStream_t s);
int vfs_stem_wgrad_fused_launch(const StemBwdArgs& a, const bf16_t* x4, int Hin, int Win, float* partial, int nblocks,
                                hipStream_t stream);
int vfs_bn_reduce_partials_launch(const float* partial, double* sums, double* scratch, int G, int bpg, int C, hipStream_t s,
                                  const P2PTail* tail = nullptr);
int vfs_bn_reduce_fused_launch(int mode, const float* partial, double* sums, double* scratch, int G, int bpg, int C,
                               const float* gamma, const float* beta, float* bnp, float* rm, float* rv, double count, float eps,
                               float momentum, float* dgamma, float* dbeta, hipStream_t s, const P2PTail* tail = nullptr);
int vfs_bn_finalize_launch(const double* sums, const float* gamma, const float* beta, float* bnp, float* rm, float* rv,
                           int G, int C, double count, float eps, float momentum, hipStream_t s);
int vfs_bn_eval_params_launch(const float* gamma, const float* beta, const float* rm, const float* rv, float* bnp, int C,
                              float eps, hipStream_t s);
int vfs_bn_act_launch(const BnActArgs& a, hipStream_t s);
int vfs_bn_relu_maxpool_launch(const BnPoolArgs& a, hipStream_t s);
int vfs_maxpool_relu_bwd_launch(const PoolBwdArgs& a, hipStream_t s);
int vfs_bn_bwd_reduce_launch(const BnBwdArgs& a, int nblk, hipStream_t s);
int vfs_bn_bwd_apply_launch(const BnBwdArgs& a, hipStream_t s);
int vfs_bn_param_grad_launch(const double* sums, float* dgamma, float* dbeta, int G, int C, hipStream_t s);
int vfs_wgrad_reduce_launch(const float* partial, float* grad, int nsplit, int Cout, int Ktot, int Cin, int KH, int KW,
                            int stem, hipStream_t stream);
// one record per layer of the table-driven split-K reduction (conv_wgrad.hip; 56 bytes, mirrored by vfs_amd/packing.py)
struct WgradReduceDesc {
  const float* partial;   // [nsplit][Cout][Ktot]
  float* grad;            // OIHW fp32 gradient (+=)
  int nsplit, Cout, Ktot, Cin, KH, KW;
  int stem;               // 1: 7x7 stem k-layout ((r*8 + s+1)*4 + c)
  int block_start;        // first workgroup of this record (128 elements per workgroup)
  int nblocks;            // workgroups serving this record
  int pad_;
};
int vfs_wgrad_reduce_table_launch(const WgradReduceDesc* tab, int n, int total_blocks, hipStream_t stream);

// ---- misc.hip ------------------------------------------------------------------------------
// imgs fp32 [B][V][3][T][H][W] (reference layout, FormatShape 'NCTHW') -> bf16 NHWC4
// out[(v*B + b)*T + t][h][w][0..3], channel 3 = 0, width padded to Wp (even) with zeros
int vfs_imgs_to_nhwc4_launch(const float* imgs, bf16_t* out, int B, int V, int T, int H, int W, int Wp, hipStream_t s);

// table-driven fp32 OIHW master weights -> bf16 packed copies (forward KRSC + dgrad CRSK)
struct PackDesc {
  const float* w;   // [Cout][Cin][KH][KW]
  bf16_t* wf;       // kind 0: [Cout][KH][KW][Cin]   kind 1 (stem): [Cout][8][8][4] (pre-zeroed)
  bf16_t* wd;       // kind 0: [Cin][KH][KW][Cout] or null
  long long start;  // first global element index of this tensor (informational)
  int Cout, Cin, KH, KW;
  int kind;         // 0: conv / linear   1: 7x7 stem   2: 1x1 with 64 | Cout, 64 | Cin (64 x 64 tiles)   3: 3x3 with 32 | Cout, 64 | Cin (32 x 64 tiles);
                    // 2 and 3 are kind 0 with 16-byte accesses (w, wf, wd 16-byte aligned): same results, chosen by the table builder
  int tile_start;   // first workgroup of this tensor: 32x32 (cout x cin) tiles (kind 0), 256-element pieces (stem), the tiles above (2, 3)
};
int vfs_pack_weights_launch(const PackDesc* table, int ntensors, long long total_tiles, hipStream_t s);

int vfs_avgpool_fwd_launch(const bf16_t* x, bf16_t* y, int N, int HW, int C, hipStream_t s);
int vfs_avgpool_bwd_launch(const bf16_t* g, bf16_t* gx, int N, int HW, int C, hipStream_t s);
int vfs_bias_grad_launch(const bf16_t* dy, float* db, int M, int C, hipStream_t s);

// SimSiam cosine loss, all T temporal rolls at once (sim_siam_base_tracker.py:31-56,
// sim_siam_head.py:165-174, sim_loss.py:42-63)
struct LossArgs {
  const bf16_t *p1, *z1, *p2, *z2;  // [N][C], N = B*T
  float* loss;                      // fwd out [K][N]   (K = T if intra_video else 1)
  const float* gloss;               // bwd in  [K][N] upstream gradient of each loss element
  bf16_t *dp1, *dp2;                // bwd out [N][C]
  int N, C, T, K, negative;
  float weight;
};
int vfs_bn_stats_raw_launch(const bf16_t* x, double* sums, int G, int rows, int C, const float* gamma, const float* beta, float* bnp,
                            float* rm, float* rv, double count, float eps, float momentum, hipStream_t s);
// SiamFC cross-correlation (xcorr.hip)
struct XcorrArgs {
  const bf16_t* z;   // [nz][Hz][Wz][C] exemplar features
  const bf16_t* x;   // [nx][H][W][C] search features
  float* out;        // [nx][H-Hz+1][W-Wz+1]
  int nz, nx, Hz, Wz, H, W, C;
  float scale;
};
// p2p.hip: SyncBN statistic exchange through IPC-mapped windows
int vfs_p2p_window_bytes_host(long long* bytes, int* max_doubles, int* max_world);
int vfs_p2p_alloc_host(void** ptr);
int vfs_p2p_free_host(void* ptr);
int vfs_p2p_export_host(void* ptr, void* handle64);
int vfs_p2p_import_host(const void* handle64, void** ptr);
int vfs_p2p_unimport_host(void* ptr);
// nn.Linear + BatchNorm1d + ReLU in one launch (conv_pw.hip: linear_bn_act_kernel)
struct LinBnArgs {
  const bf16_t* x;      // [M][K]
  const bf16_t* w;      // [C][K]
  const float* bias;    // [C] or null
  const float* gamma;
  const float* beta;
  bf16_t* raw;          // [M][C]
  bf16_t* act;          // [M][C]
  float* bnp;           // [G][4][C]
  double* sums;         // [G][2][C]
  float* rm;
  float* rv;
  int M, K, C, G, mpg, relu;
  double count;
  float eps, momentum;
};
int vfs_linear_bn_act_launch(const LinBnArgs& a, hipStream_t stream);      // conv_pw.hip
int vfs_p2p_chain_start_launch(unsigned long long* state, hipStream_t s);
int vfs_p2p_allreduce_f64_launch(double* buf, int n, void* const* peers, int rank, int world, unsigned long long* state, int phase,
                                 unsigned long long spin_limit, hipStream_t s);
// simloss.hip: CosineSimLoss on spatial inputs (pairwise affinity on the matrix cores, fp32)
int vfs_simloss_colnorm_launch(const float* x, float* inv, int B, int C, int S, hipStream_t s);
int vfs_simloss_fwd_launch(const float* a, const float* l, const float* inva, const float* invl, const float* mask, float* partial,
                           float* loss, int B, int C, int Sa, int Sl, int pairwise, int negative, float weight, hipStream_t s);
int vfs_simloss_bwd_launch(const float* other, const float* invo, const float* mask, int mask_transposed, const float* gloss, float* d,
                           int B, int C, int Sself, int Sother, int pairwise, int negative, float weight, hipStream_t s);
int vfs_simloss_norm_bwd_launch(const float* x, const float* inv, const float* d, float* dx, int B, int C, int S, hipStream_t s);
int vfs_xcorr_fwd_launch(const XcorrArgs& a, hipStream_t s);
struct XcorrBwdArgs {
  const bf16_t* z;   // [nz][Hz][Wz][C]
  const bf16_t* x;   // [nx][H][W][C]
  const float* g;    // [nx][H-Hz+1][W-Wz+1] gradient wrt the responses
  bf16_t* dz;        // [nz][Hz][Wz][C] or null
  bf16_t* dx;        // [nx][H][W][C] or null
  int nz, nx, Hz, Wz, H, W, C;
  float scale;
};
int vfs_xcorr_bwd_launch(const XcorrBwdArgs& a, hipStream_t s);
int vfs_siamfc_loss_launch(const float* x, const float* tgt, float* loss_out, float* grad, int n, int mode, float param, float scale,
                           hipStream_t s);
int vfs_adam_launch(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step,
                    hipStream_t s);
int vfs_cosine_loss_fwd_launch(const LossArgs& a, hipStream_t s);
int vfs_bn_act_fin_launch(const BnActArgs& a, const BnFin& f, hipStream_t s);
int vfs_bn_bwd_apply_raw_launch(const BnBwdArgs& a, const BnFin& f, hipStream_t s);
int vfs_bn_bwd_apply_fin_launch(const BnBwdArgs& a, const BnFin& f, hipStream_t s);
int vfs_loss_means_launch(const float* loss, float* means, int K, int N, hipStream_t s);
int vfs_cosine_loss_bwd_launch(const LossArgs& a, hipStream_t s);

// fused SGD over the flat parameter arena (torch.optim.SGD, dampening 0, no nesterov)
int vfs_sgd_launch(float* p, const float* g, float* buf, long long n, float lr, float momentum, float wd, const unsigned long long* skip,
                   hipStream_t s);
int vfs_scale_launch(float* p, long long n, float scale, hipStream_t s);
int vfs_f32_to_bf16_launch(const float* src, bf16_t* dst, long long n, float scale, hipStream_t s);
int vfs_bf16_to_f32_launch(const bf16_t* src, float* dst, long long n, hipStream_t s);

// ---- labelprop.hip -------------------------------------------------------------------------
#define LP_MAX_KEYS 64      // key frames per propagation step (precede_frames + the first frame); round 3: 24 -> 64
#define LP_MAX_CLASSES 256
#define LP_POST_BLOCKS 64
#define LP_MAX_SPLIT 96     // partial top-k lists per query (key-frame splits x window sub-splits): the workspace rows
#define LP_MAX_FSPLIT 24    // key-frame splits
struct LabelPropArgs {
  const bf16_t* fbank;  // [frames][H*W][C] L2-normalised bf16 features (the clip's feature bank)
  const float* sbank;   // [frames][H*W][CO] fp32 value logits (frame 0 = one-hot labels)
  float* out;           // [H*W][CO] propagated logits of the query frame
  float* pval;          // workspace [LP_MAX_SPLIT][H*W][10] partial top-k values ...
  int* pidx;            // ... and key ids
  int qframe;           // bank index of the query frame
  int nkeys;            // number of key frames, in the reference's order (first frame first)
  int kslot[LP_MAX_KEYS];
  int H, W, C, CO;
  int radius;           // neighbor_range // 2 (mask: distance < radius); <= 0: no spatial mask
  int non_mask_len;     // leading key frames WITHOUT the spatial mask (with_first_neighbor=False: 1)
  int topk;             // <= 10
  float inv_temp;
};
int vfs_l2norm_rows_launch(const bf16_t* x, bf16_t* y, long long P, int C, hipStream_t s);
int vfs_labelprop_launch(const LabelPropArgs& a, hipStream_t s);
int vfs_seg_postprocess_launch(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                               hipStream_t s);
int vfs_onehot_launch(const uint8_t* lab, float* out, int P, int CO, hipStream_t s);

// ---- exact_f32.hip: the fp32 evaluation path (bit-defined arithmetic, see the file header) -----------
// unsigned division by a launch constant (Granlund-Montgomery, any 32-bit numerator): q = (t + ((n - t) >> s1)) >> s2, t = mulhi(n, m)
struct VfsFastDiv {
  unsigned m, s1, s2;
};
inline VfsFastDiv vfs_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  VfsFastDiv f;
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1; f.s2 = l > 1 ? l - 1 : 0;
  return f;
}
struct ConvF32Args {
  const float* x;      // [N][H][W][Cin] fp32 NHWC, Cin % 4 == 0
  const float* w;      // [Cout][KH][KW][Cin]
  const float* scale;  // [Cout] or null: y = fmaf(acc, scale, shift) (BatchNorm in eval mode / bias)
  const float* shift;  // [Cout]
  const float* res;    // [N][Ho][Wo][Cout] identity branch or null
  float* y;            // [N][Ho][Wo][Cout]
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, relu;
  VfsFastDiv d_hw, d_wo, d_nt, d_c4, d_kw;      // set by the launcher: divisions by Ho * Wo, Wo, channel tiles, Cin / 4, KW
  int dbg;             // what-if timing (WRONG results), set by the launcher from the conv_f32_dbg option: 1 no gather, 2 no LDS stores, 4 no MFMAs, 8 staggered start, 16 no output stores, 32 no residual loads
};
struct LabelPropF32Args {
  const float* fbank;  // [frames][H*W][C] L2-normalised fp32 features
  const float* sbank;  // [frames][H*W][CO] fp32 value logits
  float* out;          // [H*W][CO]
  float* pval;         // workspace [LP_MAX_SPLIT][H*W][10] ...
  int* pidx;           // ... partial top-k lists
  int qframe, nkeys;
  int kslot[LP_MAX_KEYS];
  int H, W, C, CO, radius, topk;
  int non_mask_len;    // leading key frames WITHOUT the spatial mask (with_first_neighbor=False: 1)
  float temperature;
  const int* run_flag = nullptr;   // device word: when given and zero, the dense kernels exit at once (fallback of the two-pass path)
};
// two-pass exact label propagation (labelprop2.hip): bf16 hi/lo prefilter on the matrix cores + exact rescoring of the survivors
#define LP2_MAX_SPLIT 24     // key-frame splits of pass 1 (= candidate lists per query)
#define LP2_MAX_CAP 192      // list entries per (split, query): the workspace is sized for this (seeded thresholds: a handful are used)
#define LP2_LIST_MAX 2048    // longest single list
#define LP2_KEYTAB 2048      // key positions of a trimmed window the score kernel tabulates in LDS (8 KB); larger windows stay rectangles
#define LP2_RING 3           // LDS stages per wave of pass 1 (RING - 1 in flight)
#define LP2_BLOCK_QUEUE 8   // scores of one key block queued per query for its running top 10
struct Lp2Args {
  const float* fbank;        // [frames][H*W][C] unit rows, fp32 (pass 2: the defining arithmetic)
  const bf16_t* hl;          // [frames][H*W][C/16][hi 16 | lo 16] bf16 split copy of fbank (vfs_split_rows_bf16x2)
  const float* sbank;
  float* out;
  unsigned long long* lists; // [nsplit][H*W][cap] {score bits (low), candidate id (high)}
  int* counts;               // [nsplit][H*W]
  int* flags;                // [0]: != 0 -> a list overflowed, the dense kernel redoes the frame
  int* gthr;                 // [H*W] running lower bound of every query's 10th-best s~ (monotone int code of the float), shared by the splits
  int qframe, nkeys;
  int kslot[LP_MAX_KEYS];
  int H, W, C, CO, radius, topk, non_mask_len;
  float temperature, margin;
  int cap, nsplit, xcd_order, dbg, trim;
  int entries;               // list entries per query the workspace holds (shared out among the splits: cap <= entries / nsplit)
};
int vfs_split_rows_bf16x2_launch(const float* x, bf16_t* hl, long long P, int C, hipStream_t s);
int vfs_labelprop_f32_2pass_launch(Lp2Args a, hipStream_t s);
bool vfs_lp2_eligible(int C);
int vfs_conv_f32_launch(const ConvF32Args& a, hipStream_t s);
int vfs_imgs_to_nhwc4_f32_launch(const float* imgs, float* out, int B, int V, int T, int H, int W, hipStream_t s);
int vfs_maxpool_f32_launch(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
int vfs_l2norm_rows_f32_launch(const float* x, float* y, long long P, int C, hipStream_t s);
int vfs_labelprop_f32_launch(const LabelPropF32Args& a, hipStream_t s);
int vfs_bilinear_resize_f32_launch(const float* src, float* dst, int C, int H, int W, int Ho, int Wo, int src_nhwc, int dst_nhwc,
                                   hipStream_t s);
int vfs_seg_postprocess_exact_launch(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                                     hipStream_t s);

// DAVIS J&F ingredients (davis.hip)
struct DavisArgs {
  const uint8_t* pred;   // [T][H][W] predicted labels
  const uint8_t* gt;     // [T][H][W] ground-truth labels (255 = void when use_void)
  unsigned* bp;          // scratch: boundary words of the prediction [T-2][H][W]
  unsigned* bg;          // scratch: boundary words of the ground truth
  int* counts;           // [T-2][nobj][6] = inters, union, n_fg, n_gt, fg_match, gt_match
  int T, H, W, nobj, radius, use_void;
};
int vfs_davis_counts_launch(const DavisArgs& a, hipStream_t s);

// training input pipeline (pipeline.hip)
struct PipelineArgs {
  const uint8_t* src;    // [F][Hs][Ws][3] decoded RGB frames, F = B*V*T in pipeline order (b, v, t)
  const int* boxes;      // [F][4] crop box left, top, right, bottom (img[top:bottom, left:right])
  const uint8_t* flips;  // [F] horizontal flip after the resize
  float* imgs;           // optional out: fp32 [B][V][3][T][Ho][Wo]
  bf16_t* x4;            // optional out: bf16 NHWC4 [V*B*T][Ho][Wp][4]
  int B, V, T, Hs, Ws, Ho, Wo, Wp;
  double mean[3], stdinv[3];
};
int vfs_crop_resize_flip_norm_launch(const PipelineArgs& a, hipStream_t s);


// extern "C" entry points of libvfs_hip.so (declared in include/vfs_hip.h).
#include <string.h>

#include "../../include/vfs_hip.h"
#include "../../include/vfs_hip_tuning.h"
#include "vfs_conv.h"
#include "vfs_ops.h"
#include "vfs_p2p.h"
#include "vfs_wgrad_tail.h"

static thread_local char g_err[512] = "";

int vfs_set_error(int code, const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}
int vfs_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[400];
    snprintf(buf, sizeof(buf), "%s: launch failed: %s", what, hipGetErrorString(e));
    return vfs_set_error(VFS_ERR_LAUNCH, buf);
  }
  return VFS_OK;
}

#define S(s) ((hipStream_t)(s))

int vfs_option_halo = 1;
int vfs_option_stem_blocks = 0;
extern int vfs_option_bn_ticket, vfs_option_bn_chunk_rows, vfs_option_bn_wide, vfs_option_bn_wide_min_mb;
int vfs_option_stem_direct = 1;
extern int vfs_option_igemm_xcd, vfs_option_igemm_narrow_below;
extern int vfs_option_igemm_ring_mfma32, vfs_option_igemm_ring_gather, vfs_option_igemm_bc, vfs_option_igemm_onek, vfs_option_igemm_ring_tiles, vfs_option_igemm_ring_upfront, vfs_option_igemm_ring_fbn, vfs_option_wgrad_lin, vfs_option_wgrad_lin2, vfs_option_wgrad_ring, vfs_option_wgrad_xcd, vfs_option_halo_xcd, vfs_option_igemm_mfma_stats, vfs_option_lpx_target, vfs_option_lpx_wgs, vfs_option_lpx_minb, vfs_option_lp2, vfs_option_lp2_fpb, vfs_option_lp2_cap, vfs_option_lp2_xcd, vfs_option_lp2_trim, vfs_option_lp2_dbg, vfs_option_conv_f32_variant, vfs_option_conv_f32_dbg;

static ConvGeom make_geom(int N, int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride, int pad, int Ktot) {
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = Ktot;
  g.M = N * Ho * Wo;
  return g;
}

extern "C" {

const char* vfs_last_error(void) { return g_err; }
int vfs_abi_version(void) { return 2; }      // 2: vfs_sgd_step(skip_flag), vfs_labelprop*(workspace_bytes)
int vfs_set_option(const char* name, int value) {
  if (!strcmp(name, "halo")) { vfs_option_halo = value; return VFS_OK; }
  if (!strcmp(name, "halo_min_fill")) { vfs_option_halo_min_fill = value; return VFS_OK; }
  if (!strcmp(name, "stem_blocks")) { vfs_option_stem_blocks = value; return VFS_OK; }
  if (!strcmp(name, "bn_ticket")) { vfs_option_bn_ticket = value; return VFS_OK; }
  if (!strcmp(name, "bn_wide")) { vfs_option_bn_wide = value; return VFS_OK; }
  if (!strcmp(name, "bn_wide_min_mb")) { vfs_option_bn_wide_min_mb = value; return VFS_OK; }
  if (!strcmp(name, "bn_chunk_rows")) { vfs_option_bn_chunk_rows = value > 0 ? value : 64; return VFS_OK; }
  if (!strcmp(name, "stem_direct")) { vfs_option_stem_direct = value; return VFS_OK; }
  if (!strcmp(name, "igemm_bc")) { vfs_option_igemm_bc = value; return VFS_OK; }
  if (!strcmp(name, "igemm_xcd")) { vfs_option_igemm_xcd = value; return VFS_OK; }
  if (!strcmp(name, "igemm_narrow_below")) { vfs_option_igemm_narrow_below = value; return VFS_OK; }
  if (!strcmp(name, "igemm_onek")) { vfs_option_igemm_onek = value; return VFS_OK; }
  if (!strcmp(name, "igemm_ring_tiles")) { vfs_option_igemm_ring_tiles = value; return VFS_OK; }
  if (!strcmp(name, "lpx_target")) { vfs_option_lpx_target = value; return VFS_OK; }
  if (!strcmp(name, "lpx_wgs")) { vfs_option_lpx_wgs = value; return VFS_OK; }
  if (!strcmp(name, "lpx_minb")) { vfs_option_lpx_minb = value; return VFS_OK; }
  if (!strcmp(name, "conv_f32_dbg")) { vfs_option_conv_f32_dbg = value; return VFS_OK; }
  if (!strcmp(name, "conv_f32_variant")) { vfs_option_conv_f32_variant = value; return VFS_OK; }
  if (!strcmp(name, "lp2")) { vfs_option_lp2 = value; return VFS_OK; }
  if (!strcmp(name, "lp2_fpb")) { vfs_option_lp2_fpb = value; return VFS_OK; }
  if (!strcmp(name, "lp2_trim")) { vfs_option_lp2_trim = value; return VFS_OK; }
  if (!strcmp(name, "lp2_xcd")) { vfs_option_lp2_xcd = value; return VFS_OK; }
  if (!strcmp(name, "lp2_dbg")) { vfs_option_lp2_dbg = value; return VFS_OK; }
  if (!strcmp(name, "lp2_cap")) { vfs_option_lp2_cap = value <= 0 ? 0 : (value < 16 ? 16 : value); return VFS_OK; }
  if (!strcmp(name, "igemm_mfma_stats")) { vfs_option_igemm_mfma_stats = value; return VFS_OK; }
  if (!strcmp(name, "wgrad_lin2")) { vfs_option_wgrad_lin2 = value; return VFS_OK; }
  if (!strcmp(name, "wgrad_ring")) { vfs_option_wgrad_ring = value; return VFS_OK; }
  if (!strcmp(name, "wgrad_lin")) { vfs_option_wgrad_lin = value; return VFS_OK; }
  if (!strcmp(name, "wgrad_xcd")) { vfs_option_wgrad_xcd = value; return VFS_OK; }
  if (!strcmp(name, "halo_deep_max")) { vfs_option_halo_deep_max = value; return VFS_OK; }
  if (!strcmp(name, "halo_xcd")) { vfs_option_halo_xcd = value; return VFS_OK; }
  if (!strcmp(name, "igemm_ring_fbn")) { vfs_option_igemm_ring_fbn = value; return VFS_OK; }
  if (!strcmp(name, "igemm_ring_mfma32")) { vfs_option_igemm_ring_mfma32 = value; return VFS_OK; }
  if (!strcmp(name, "igemm_ring_gather")) { vfs_option_igemm_ring_gather = value; return VFS_OK; }
  if (!strcmp(name, "igemm_skinny")) { vfs_option_igemm_skinny = value; return VFS_OK; }
  if (!strcmp(name, "igemm_pw")) { vfs_option_igemm_pw = value; return VFS_OK; }
  if (!strcmp(name, "igemm_pw_min_tiles")) { vfs_option_igemm_pw_min_tiles = value; return VFS_OK; }
  if (!strcmp(name, "igemm_ring_upfront")) { vfs_option_igemm_ring_upfront = value; return VFS_OK; }
  return vfs_set_error(VFS_ERR_ARG, "vfs_set_option: unknown option");
}

int vfs_imgs_to_nhwc4(const float* imgs, vfs_bf16* out, int B, int V, int T, int H, int W, int Wp, vfs_stream_t stream) {
  if (Wp < W || (Wp & 1)) return vfs_set_error(VFS_ERR_SHAPE, "imgs_to_nhwc4: Wp must be even and >= W");
  return vfs_imgs_to_nhwc4_launch(imgs, out, B, V, T, H, W, Wp, S(stream));
}

int vfs_pack_weights(const void* desc, int ntensors, long long total, vfs_stream_t stream) {
  return vfs_pack_weights_launch((const PackDesc*)desc, ntensors, total, S(stream));
}

int vfs_conv_fwd(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, int N, int H, int W,
                 int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, vfs_stream_t stream) {
  ConvArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.src = x; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = bias; a.stats = stats; a.Cout = Cout; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
}
int vfs_conv_fwd_coarse(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, float* stats_coarse,
                        uint32_t* tickets, int coarse_log2, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                        int stride, int pad, vfs_stream_t stream) {
  if (!stats || !stats_coarse || !tickets || coarse_log2 < 1 || coarse_log2 > 8)
    return vfs_set_error(VFS_ERR_ARG, "conv_fwd_coarse: stats, stats_coarse, tickets and 1 <= coarse_log2 <= 8");
  ConvArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.src = x; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = bias; a.stats = stats; a.Cout = Cout; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  a.stats_coarse = stats_coarse; a.stats_tickets = tickets; a.coarse_log2 = coarse_log2;
  return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
}
int vfs_conv_fwd_dilated(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, int N, int H, int W,
                         int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int dilation, vfs_stream_t stream) {
  if (dilation < 1) return vfs_set_error(VFS_ERR_ARG, "conv_fwd_dilated: dilation >= 1");
  if (Ho != (H + 2 * pad - dilation * (KH - 1) - 1) / stride + 1 || Wo != (W + 2 * pad - dilation * (KW - 1) - 1) / stride + 1)
    return vfs_set_error(VFS_ERR_SHAPE, "conv_fwd_dilated: output size does not match (H + 2 pad - dilation (K - 1) - 1) / stride + 1");
  ConvArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.g.dil = dilation;
  a.src = x; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = bias; a.stats = stats; a.Cout = Cout; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
}
int vfs_conv_fwd_splitk(const vfs_bf16* x, const vfs_bf16* wf, vfs_bf16* y, const float* bias, float* stats, float* ks_ws, int ksplit,
                        int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                        vfs_stream_t stream) {
  if (ksplit < 1 || (ksplit > 1 && !ks_ws)) return vfs_set_error(VFS_ERR_ARG, "conv_fwd_splitk: workspace");
  ConvArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.src = x; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = bias; a.stats = stats; a.Cout = Cout; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  a.ks_ws = ks_ws; a.ksplit = ksplit;
  return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
}
int vfs_conv_dgrad_splitk(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, float* ks_ws, int ksplit, int N,
                          int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                          vfs_stream_t stream) {
  if (ksplit < 1 || (ksplit > 1 && !ks_ws)) return vfs_set_error(VFS_ERR_ARG, "conv_dgrad_splitk: workspace");
  if (stride != 1) return vfs_set_error(VFS_ERR_SHAPE, "conv_dgrad_splitk: stride 1 only");
  ConvArgs a;
  a.g = make_geom(N, Ho, Wo, Cout, H, W, KH, KW, stride, pad, KH * KW * Cout);
  a.src = dy; a.wgt = wd; a.out = dx; a.add = add; a.bias = nullptr; a.stats = nullptr; a.Cout = Cin; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  a.ks_ws = ks_ws; a.ksplit = ksplit;
  return vfs_conv_igemm_dispatch(a, GATHER_DGRAD, S(stream));
}
int vfs_conv_fwd_bnin(const vfs_bf16* x_raw, const float* in_bnp, int in_npg, const vfs_bf16* wf, vfs_bf16* y, const float* bias,
                      float* stats, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                      vfs_stream_t stream) {
  if (!in_bnp || in_npg <= 0) return vfs_set_error(VFS_ERR_ARG, "conv_fwd_bnin: BatchNorm parameters of the input");
  ConvArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.src = x_raw; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = bias; a.stats = stats; a.Cout = Cout; a.bn = BnBwdFuse{};
  a.in_bnp = in_bnp; a.in_npg = in_npg;
  const bool smallw = vfs_small_map(H, W);
  // round 6: also the 1x1 / stride-1 forward of the implicit-GEMM kernel (the conv2 -> conv3 edge), groups of whole 128-pixel tiles
  const bool pw = KH * KW == 1 && stride == 1 && pad == 0 && H == Ho && W == Wo && Cin % 64 == 0 && ((long long)in_npg * H * W) % 128 == 0;
  if (pw) return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
  if (!vfs_option_halo || Cin % 64 || (size_t)N * H * W * Cin * 2 >= 0xFFFFFFF0ull || !vfs_conv_halo_eligible(a, GATHER_FWD) ||
      (smallw && in_npg % 2))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_fwd_bnin: the 3x3/stride-1 halo-tile kernel and the 1x1/stride-1 kernel fold the input BatchNorm");
  return vfs_conv_igemm_dispatch(a, GATHER_FWD, S(stream));
}

int vfs_stem_fwd(const vfs_bf16* x4, const vfs_bf16* wf, vfs_bf16* y, float* stats, int N, int H, int Wp, int Ho, int Wo,
                 vfs_stream_t stream) {
  if (Wp & 1) return vfs_set_error(VFS_ERR_SHAPE, "stem_fwd: padded width must be even");
  ConvArgs a;
  a.g = make_geom(N, H, Wp, 4, Ho, Wo, 7, 7, 2, 3, 256);
  a.src = x4; a.wgt = wf; a.out = y; a.add = nullptr; a.bias = nullptr; a.stats = stats; a.Cout = 64; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  if (vfs_option_stem_direct && (size_t)N * H * Wp * 8 < 0xFFFFFFF0ull) return vfs_stem_fwd_direct_launch(a, S(stream));
  return vfs_conv_igemm_dispatch(a, GATHER_STEM, S(stream));
}

static int set_add_mask(ConvArgs& a, const vfs_bf16* add, const uint8_t* add_mask, int N, int H, int W, int Cin) {
  if (!add_mask) return VFS_OK;
  if (!add || Cin % 64) return vfs_set_error(VFS_ERR_SHAPE, "conv_dgrad: add_mask needs an add operand and Cin % 64 == 0");
  a.add_mask = add_mask; a.add_rows = (long long)N * H * W;
  return VFS_OK;
}
int vfs_conv_dgrad_maskadd(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, const uint8_t* add_mask, int N,
                           int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, vfs_stream_t stream) {
  // gather source = dy [N,Ho,Wo,Cout]; destination grid = dx [N,H,W,Cin]
  ConvArgs a;
  a.g = make_geom(N, Ho, Wo, Cout, H, W, KH, KW, stride, pad, KH * KW * Cout);
  a.src = dy; a.wgt = wd; a.out = dx; a.add = add; a.bias = nullptr; a.stats = nullptr; a.Cout = Cin; a.bn = BnBwdFuse{};
  a.in_bnp = nullptr; a.in_npg = 0;
  if (int rc = set_add_mask(a, add, add_mask, N, H, W, Cin)) return rc;
  return vfs_conv_igemm_dispatch(a, GATHER_DGRAD, S(stream));
}
int vfs_conv_dgrad(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, int N, int H, int W, int Cin,
                   int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, vfs_stream_t stream) {
  return vfs_conv_dgrad_maskadd(dy, wd, dx, add, nullptr, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, stream);
}
int vfs_conv_dgrad_bn_maskadd(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, const uint8_t* add_mask,
                              const vfs_bf16* bn_x, const vfs_bf16* bn_y, const float* bnp, float* bn_partial, int bn_mpg, int bn_relu,
                              int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                              vfs_stream_t stream) {
  if (stride != 1) return vfs_set_error(VFS_ERR_SHAPE, "conv_dgrad_bn: stride 1 only (strided dgrads run per parity class)");
  if (!bn_x || !bnp || !bn_partial || bn_mpg <= 0) return vfs_set_error(VFS_ERR_ARG, "conv_dgrad_bn: null statistics operand");
  const long long M = (long long)N * H * W;
  ConvArgs a;
  a.g = make_geom(N, Ho, Wo, Cout, H, W, KH, KW, stride, pad, KH * KW * Cout);
  a.src = dy; a.wgt = wd; a.out = dx; a.add = add; a.bias = nullptr; a.stats = nullptr; a.Cout = Cin;
  a.in_bnp = nullptr; a.in_npg = 0;
  if (int rc = set_add_mask(a, add, add_mask, N, H, W, Cin)) return rc;
  a.bn.x = bn_x; a.bn.y = bn_y; a.bn.bnp = bnp; a.bn.partial = bn_partial; a.bn.mpg = bn_mpg; a.bn.relu = bn_relu;
  // a statistics row must belong to ONE group: spatial tiles never straddle images (halo kernels: groups of whole
  // images), linear blocks are 128 pixels
  const bool tiles = vfs_option_halo && Cout % 64 == 0 && vfs_conv_halo_eligible(a, GATHER_DGRAD);
  if (bn_mpg < M && (tiles ? bn_mpg % ((long long)H * W) != 0 : bn_mpg % 128 != 0))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_dgrad_bn: groups must be whole images (tile kernels) / multiples of 128 pixels");
  return vfs_conv_igemm_dispatch(a, GATHER_DGRAD, S(stream));
}
int vfs_conv_dgrad_bn(const vfs_bf16* dy, const vfs_bf16* wd, vfs_bf16* dx, const vfs_bf16* add, const vfs_bf16* bn_x,
                      const vfs_bf16* bn_y, const float* bnp, float* bn_partial, int bn_mpg, int bn_relu, int N, int H, int W,
                      int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, vfs_stream_t stream) {
  return vfs_conv_dgrad_bn_maskadd(dy, wd, dx, add, nullptr, bn_x, bn_y, bnp, bn_partial, bn_mpg, bn_relu, N, H, W, Cin, Ho, Wo, Cout,
                                   KH, KW, stride, pad, stream);
}

int vfs_conv_wgrad(const vfs_bf16* dy, const vfs_bf16* x, float* partial, float* grad, int N, int H, int W, int Cin, int Ho,
                   int Wo, int Cout, int KH, int KW, int stride, int pad, int nsplit, int pix_per_split, vfs_stream_t stream) {
  WgradArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.dy = dy; a.x = x; a.partial = partial; a.Cout = Cout; a.pix_per_split = pix_per_split; a.nsplit = nsplit;
  a.in_bnp = nullptr; a.in_npg = 0;
  int rc;
  if (vfs_option_halo && vfs_wgrad_halo_eligible(a, GATHER_FWD)) {
    rc = vfs_wgrad_halo_dispatch(a, S(stream), &nsplit);   // may use fewer splits than offered
  } else {
    rc = vfs_conv_wgrad_dispatch(a, GATHER_FWD, S(stream));
  }
  if (rc) return rc;
  if (!grad) {      // the caller reduces the partials later (vfs_wgrad_reduce_table): its table needs the split count it offered
    if (nsplit != a.nsplit) return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad: deferred reduction needs a split plan the kernel takes as offered");
    return VFS_OK;
  }
  return vfs_wgrad_reduce_launch(partial, grad, nsplit, Cout, a.g.Ktot, Cin, KH, KW, 0, S(stream));
}
int vfs_conv_wgrad_bnin(const vfs_bf16* dy, const vfs_bf16* x_raw, const float* in_bnp, int in_npg, float* partial, float* grad, int N,
                        int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int nsplit,
                        int pix_per_split, vfs_stream_t stream) {
  if (!in_bnp || in_npg <= 0) return vfs_set_error(VFS_ERR_ARG, "conv_wgrad_bnin: BatchNorm parameters of the input");
  WgradArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.dy = dy; a.x = x_raw; a.partial = partial; a.Cout = Cout; a.pix_per_split = pix_per_split; a.nsplit = nsplit;
  a.in_bnp = in_bnp; a.in_npg = in_npg;
  if (KH * KW == 1 && stride == 1 && pad == 0 && H == Ho && W == Wo) {      // round 6: the 1x1 / stride-1 kernel (register-staged) folds it too
    int rc1 = vfs_conv_wgrad_dispatch(a, GATHER_FWD, S(stream));
    if (rc1 || !grad) return rc1;
    return vfs_wgrad_reduce_launch(partial, grad, nsplit, Cout, a.g.Ktot, Cin, KH, KW, 0, S(stream));
  }
  if (!vfs_option_halo || !vfs_wgrad_halo_eligible(a, GATHER_FWD) || (vfs_small_map(H, W) && in_npg % 2))
    return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad_bnin: the 3x3/stride-1 halo-tile kernel and the 1x1/stride-1 kernel fold the input BatchNorm");
  int rc = vfs_wgrad_halo_dispatch(a, S(stream), &nsplit);
  if (rc) return rc;
  if (!grad) {
    if (nsplit != a.nsplit) return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad_bnin: deferred reduction needs a split plan the kernel takes as offered");
    return VFS_OK;
  }
  return vfs_wgrad_reduce_launch(partial, grad, nsplit, Cout, a.g.Ktot, Cin, KH, KW, 0, S(stream));
}

int vfs_wgrad_tickets(void) { return VFS_WGRAD_TICKETS; }
int vfs_conv_wgrad_inl(const vfs_bf16* dy, const vfs_bf16* x, const float* in_bnp, int in_npg, float* partial, float* grad,
                       unsigned* tickets, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad, int nsplit, int pix_per_split, vfs_stream_t stream) {
  if (!grad || !tickets || !partial) return vfs_set_error(VFS_ERR_ARG, "conv_wgrad_inl: partial, grad and tickets must be given");
  if (in_bnp && in_npg <= 0) return vfs_set_error(VFS_ERR_ARG, "conv_wgrad_inl: images per BatchNorm group of the input");
  WgradArgs a;
  a.g = make_geom(N, H, W, Cin, Ho, Wo, KH, KW, stride, pad, KH * KW * Cin);
  a.dy = dy; a.x = x; a.partial = partial; a.Cout = Cout; a.pix_per_split = pix_per_split; a.nsplit = nsplit;
  a.in_bnp = in_bnp; a.in_npg = in_bnp ? in_npg : 0;
  a.grad = grad; a.tickets = tickets;
  if (vfs_option_halo && vfs_wgrad_halo_eligible(a, GATHER_FWD) && !(in_bnp && vfs_small_map(H, W) && in_npg % 2))
    return vfs_wgrad_halo_dispatch(a, S(stream), &nsplit);
  if (in_bnp) return vfs_set_error(VFS_ERR_SHAPE, "conv_wgrad_inl: only the 3x3/stride-1 halo-tile kernel folds the input BatchNorm");
  return vfs_conv_wgrad_dispatch(a, GATHER_FWD, S(stream));
}

int vfs_stem_wgrad(const vfs_bf16* dy, const vfs_bf16* x4, float* partial, float* grad, int N, int H, int Wp, int Ho, int Wo,
                   int nsplit, int pix_per_split, vfs_stream_t stream) {
  WgradArgs a;
  a.g = make_geom(N, H, Wp, 4, Ho, Wo, 7, 7, 2, 3, 256);
  a.dy = dy; a.x = x4; a.partial = partial; a.Cout = 64; a.pix_per_split = pix_per_split; a.nsplit = nsplit;
  int rc = vfs_conv_wgrad_dispatch(a, GATHER_STEM, S(stream));
  if (rc || !grad) return rc;
  return vfs_wgrad_reduce_launch(partial, grad, nsplit, 64, 256, 3, 7, 7, 1, S(stream));
}

int vfs_wgrad_reduce_table(const void* desc, int nrecords, int total_blocks, vfs_stream_t stream) {
  if (nrecords > 0 && !desc) return vfs_set_error(VFS_ERR_ARG, "wgrad_reduce_table: null table");
  return vfs_wgrad_reduce_table_launch((const WgradReduceDesc*)desc, nrecords, total_blocks, S(stream));
}

int vfs_bias_grad(const vfs_bf16* dy, float* db, int M, int C, vfs_stream_t stream) {
  return vfs_bias_grad_launch(dy, db, M, C, S(stream));
}

int vfs_bn_reduce_partials(const float* partial, double* sums, double* scratch, int G, int bpg, int C, vfs_stream_t stream) {
  return vfs_bn_reduce_partials_launch(partial, sums, scratch, G, bpg, C, S(stream));
}
int vfs_bn_finalize(const double* sums, const float* gamma, const float* beta, float* bnp, float* running_mean,
                    float* running_var, int G, int C, double count, float eps, float momentum, vfs_stream_t stream) {
  return vfs_bn_finalize_launch(sums, gamma, beta, bnp, running_mean, running_var, G, C, count, eps, momentum, S(stream));
}
int vfs_bn_stats_finalize(const float* partial, double* sums, double* scratch, const float* gamma, const float* beta, float* bnp,
                          float* running_mean, float* running_var, int G, int bpg, int C, double count, float eps, float momentum,
                          vfs_stream_t stream) {
  return vfs_bn_reduce_fused_launch(0, partial, sums, scratch, G, bpg, C, gamma, beta, bnp, running_mean, running_var, count, eps,
                                    momentum, nullptr, nullptr, S(stream));
}
int vfs_bn_stats_raw_finalize(const vfs_bf16* raw, double* sums, const float* gamma, const float* beta, float* bnp, float* running_mean,
                              float* running_var, int G, int rows_per_group, int C, double count, float eps, float momentum,
                              vfs_stream_t stream) {
  if (!raw || !sums || !gamma || !beta || !bnp) return vfs_set_error(VFS_ERR_ARG, "bn_stats_raw_finalize: null buffer");
  return vfs_bn_stats_raw_launch(raw, sums, G, rows_per_group, C, gamma, beta, bnp, running_mean, running_var, count, eps, momentum,
                                 S(stream));
}
int vfs_linear_bn_act(const vfs_bf16* x, const vfs_bf16* wf, const float* bias, const float* gamma, const float* beta, vfs_bf16* raw,
                      vfs_bf16* act, float* bnp, double* sums, float* running_mean, float* running_var, int M, int K, int C, int mpg,
                      int relu, double count, float eps, float momentum, vfs_stream_t stream) {
  if (!x || !wf || !gamma || !beta || !raw || !act || !bnp || !sums) return vfs_set_error(VFS_ERR_ARG, "linear_bn_act: null buffer");
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "linear_bn_act: M % mpg");
  LinBnArgs a;
  a.x = x; a.w = wf; a.bias = bias; a.gamma = gamma; a.beta = beta; a.raw = raw; a.act = act; a.bnp = bnp; a.sums = sums;
  a.rm = running_mean; a.rv = running_var; a.M = M; a.K = K; a.C = C; a.G = M / mpg; a.mpg = mpg; a.relu = relu; a.count = count;
  a.eps = eps; a.momentum = momentum;
  return vfs_linear_bn_act_launch(a, S(stream));
}
int vfs_bn_bwd_sums_paramgrad(const float* partial, double* sums, double* scratch, float* dgamma, float* dbeta, int G, int bpg,
                              int C, vfs_stream_t stream) {
  return vfs_bn_reduce_fused_launch(1, partial, sums, scratch, G, bpg, C, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 0.f, 0.f,
                                    dgamma, dbeta, S(stream));
}
int vfs_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* bnp,
                       int C, float eps, vfs_stream_t stream) {
  return vfs_bn_eval_params_launch(gamma, beta, running_mean, running_var, bnp, C, eps, S(stream));
}
int vfs_bn_act_mask(const vfs_bf16* x, const float* bnp, const vfs_bf16* res, const vfs_bf16* rres, const float* rbnp, vfs_bf16* y,
                    uint8_t* mask_bits, long long M, int C, int mpg, int relu, vfs_stream_t stream) {
  BnActArgs a;
  a.x = x; a.bnp = bnp; a.res = res; a.rres = rres; a.rbnp = rbnp; a.y = y; a.M = M; a.C = C; a.mpg = mpg; a.relu = relu;
  a.mbits = mask_bits;
  return vfs_bn_act_launch(a, S(stream));
}
int vfs_bn_act(const vfs_bf16* x, const float* bnp, const vfs_bf16* res, const vfs_bf16* rres, const float* rbnp, vfs_bf16* y,
               long long M, int C, int mpg, int relu, vfs_stream_t stream) {
  return vfs_bn_act_mask(x, bnp, res, rres, rbnp, y, nullptr, M, C, mpg, relu, stream);
}
int vfs_bn_act_fin_mask(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp, double* sums,
                        float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres, const float* rbnp, vfs_bf16* y,
                        uint8_t* mask_bits, long long M, int C, int mpg, int relu, double count, float eps, float momentum,
                        vfs_stream_t stream) {
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "bn_act_fin: M % mpg");
  BnActArgs a;
  a.x = x; a.bnp = bnp; a.res = res; a.rres = rres; a.rbnp = rbnp; a.y = y; a.M = M; a.C = C; a.mpg = mpg; a.relu = relu;
  a.mbits = mask_bits;
  BnFin f;
  f.partial = partial; f.bpg = bpg; f.G = (int)(M / mpg); f.gamma = gamma; f.beta = beta; f.bnp = bnp; f.sums = sums;
  f.running_mean = running_mean; f.running_var = running_var; f.count = count; f.eps = eps; f.momentum = momentum;
  return vfs_bn_act_fin_launch(a, f, S(stream));
}
static P2PTail make_tail(const void* peers, int rank, int world, void* state, long long spin_limit) {
  P2PTail x;
  x.peers = (void* const*)peers; x.rank = rank; x.world = world; x.state = (unsigned long long*)state;
  x.spin_limit = (unsigned long long)(spin_limit < 1 ? 1 : spin_limit);
  return x;
}
int vfs_bn_act_fin_xchg(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp, double* sums,
                        float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres, const float* rbnp, vfs_bf16* y,
                        uint8_t* mask_bits, long long M, int C, int mpg, int relu, double count, float eps, float momentum,
                        const void* peers, int rank, int world, void* state, long long spin_limit, int seq, vfs_stream_t stream) {
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "bn_act_fin_xchg: M % mpg");
  if (seq < 0 || seq >= 4095) return vfs_set_error(VFS_ERR_ARG, "bn_act_fin_xchg: 0 <= seq < 4095");
  if (!peers || !state || !partial) return vfs_set_error(VFS_ERR_ARG, "bn_act_fin_xchg: statistics rows, peers and state must be given");
  BnActArgs a;
  a.x = x; a.bnp = bnp; a.res = res; a.rres = rres; a.rbnp = rbnp; a.y = y; a.M = M; a.C = C; a.mpg = mpg; a.relu = relu;
  a.mbits = mask_bits;
  BnFin f;
  f.partial = partial; f.bpg = bpg; f.G = (int)(M / mpg); f.gamma = gamma; f.beta = beta; f.bnp = bnp; f.sums = sums;
  f.running_mean = running_mean; f.running_var = running_var; f.count = count; f.eps = eps; f.momentum = momentum;
  f.x = make_tail(peers, rank, world, state, spin_limit);
  f.x.seq = seq;
  return vfs_bn_act_fin_launch(a, f, S(stream));
}
int vfs_bn_bwd_apply_fin_xchg(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, const float* partial, int bpg,
                              double* sums, float* dgamma, float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg,
                              double count, int relu, const void* peers, int rank, int world, void* state, long long spin_limit,
                              int seq, vfs_stream_t stream) {
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_fin_xchg: M % mpg");
  if (seq < 0 || seq >= 4095) return vfs_set_error(VFS_ERR_ARG, "bn_bwd_apply_fin_xchg: 0 <= seq < 4095");
  if (!peers || !state || !partial) return vfs_set_error(VFS_ERR_ARG, "bn_bwd_apply_fin_xchg: statistics rows, peers and state must be given");
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = y; a.x = x; a.bnp = bnp; a.dx = dx; a.gm = gm; a.M = M; a.C = C; a.mpg = mpg; a.count = count; a.relu = relu;
  BnFin f;
  f.partial = partial; f.bpg = bpg; f.G = (int)(M / mpg); f.sums = sums; f.dgamma = dgamma; f.dbeta = dbeta;
  f.x = make_tail(peers, rank, world, state, spin_limit);
  f.x.seq = seq;
  return vfs_bn_bwd_apply_fin_launch(a, f, S(stream));
}
int vfs_p2p_chain_start(void* state, vfs_stream_t stream) {
  if (!state) return vfs_set_error(VFS_ERR_ARG, "p2p_chain_start: null state");
  return vfs_p2p_chain_start_launch((unsigned long long*)state, S(stream));
}
int vfs_bn_act_fin(const vfs_bf16* x, const float* partial, int bpg, const float* gamma, const float* beta, float* bnp, double* sums,
                   float* running_mean, float* running_var, const vfs_bf16* res, const vfs_bf16* rres, const float* rbnp, vfs_bf16* y,
                   long long M, int C, int mpg, int relu, double count, float eps, float momentum, vfs_stream_t stream) {
  return vfs_bn_act_fin_mask(x, partial, bpg, gamma, beta, bnp, sums, running_mean, running_var, res, rres, rbnp, y, nullptr, M, C, mpg,
                             relu, count, eps, momentum, stream);
}
int vfs_bn_relu_maxpool(const vfs_bf16* x, const float* bnp, vfs_bf16* y, uint8_t* idx, vfs_bf16* xpool, int N, int H, int W, int C,
                        int Hp, int Wp, int npg, vfs_stream_t stream) {
  BnPoolArgs a;
  a.x = x; a.bnp = bnp; a.y = y; a.idx = idx; a.xpool = xpool; a.N = N; a.H = H; a.W = W; a.C = C; a.Hp = Hp; a.Wp = Wp; a.npg = npg;
  return vfs_bn_relu_maxpool_launch(a, S(stream));
}
int vfs_maxpool_relu_bwd(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx, vfs_bf16* ga, int N, int H, int W, int C,
                         int Hp, int Wp, vfs_stream_t stream) {
  PoolBwdArgs a;
  a.gp = gp; a.yp = yp; a.idx = idx; a.ga = ga; a.N = N; a.H = H; a.W = W; a.C = C; a.Hp = Hp; a.Wp = Wp;
  return vfs_maxpool_relu_bwd_launch(a, S(stream));
}
int vfs_bn_bwd_reduce(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, float* partial, long long M,
                      int C, int mpg, int ppb, int relu, vfs_stream_t stream) {
  if (ppb <= 0 || mpg % ppb) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_reduce: pixels-per-group % pixels-per-block");
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = y; a.x = x; a.bnp = bnp; a.partial = partial; a.M = M; a.C = C; a.mpg = mpg; a.ppb = ppb; a.relu = relu;
  return vfs_bn_bwd_reduce_launch(a, (int)((M + ppb - 1) / ppb), S(stream));
}
int vfs_bn_bwd_apply(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, const double* sums,
                     vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg, double count, int relu, vfs_stream_t stream) {
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = y; a.x = x; a.bnp = bnp; a.sums = sums; a.dx = dx; a.gm = gm; a.M = M; a.C = C; a.mpg = mpg; a.count = count;
  a.relu = relu;
  return vfs_bn_bwd_apply_launch(a, S(stream));
}
int vfs_bn_bwd_apply_fin(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, const float* partial, int bpg,
                         double* sums, float* dgamma, float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg,
                         double count, int relu, vfs_stream_t stream) {
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_fin: M % mpg");
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = y; a.x = x; a.bnp = bnp; a.dx = dx; a.gm = gm; a.M = M; a.C = C; a.mpg = mpg; a.count = count; a.relu = relu;
  BnFin f;
  f.partial = partial; f.bpg = bpg; f.G = (int)(M / mpg); f.sums = sums; f.dgamma = dgamma; f.dbeta = dbeta;
  return vfs_bn_bwd_apply_fin_launch(a, f, S(stream));
}
int vfs_bn_bwd_apply_raw(const vfs_bf16* g, const vfs_bf16* y, const vfs_bf16* x, const float* bnp, double* sums, float* dgamma,
                         float* dbeta, vfs_bf16* dx, vfs_bf16* gm, long long M, int C, int mpg, double count, int relu,
                         vfs_stream_t stream) {
  if (mpg <= 0 || M % mpg) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_raw: M % mpg");
  int gcd = mpg, r = 512;      // ONE statistics row per group: the shapes the two-launch form serves with ppb == mpg (gcd(mpg, 512) == mpg or < 16)
  while (r) { const int q = gcd % r; gcd = r; r = q; }
  if (mpg > 512 || !(gcd == mpg || gcd < 16)) return vfs_set_error(VFS_ERR_SHAPE, "bn_bwd_apply_raw: one statistics row per group (mpg <= 512 and gcd(mpg, 512) == mpg or < 16)");
  const int ppb = mpg;
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = y; a.x = x; a.bnp = bnp; a.dx = dx; a.gm = gm; a.M = M; a.C = C; a.mpg = mpg; a.ppb = ppb; a.count = count; a.relu = relu;
  BnFin f;
  f.partial = nullptr; f.bpg = 1; f.G = (int)(M / mpg); f.sums = sums; f.dgamma = dgamma; f.dbeta = dbeta;
  return vfs_bn_bwd_apply_raw_launch(a, f, S(stream));
}
int vfs_stem_pool_bn_bwd_reduce(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx, const vfs_bf16* x, const vfs_bf16* xpool,
                                const float* bnp, float* partial, int N, int H, int W, int C, int Hp, int Wp, int npg, int ppb,
                                vfs_stream_t stream) {
  const long long mpg = (long long)npg * Hp * Wp;
  if (ppb <= 0 || mpg % ppb) return vfs_set_error(VFS_ERR_SHAPE, "stem_pool_bn_bwd_reduce: pooled pixels per group % ppb");
  if (!gp || !idx || !bnp || !partial || (!x && !xpool) || (!xpool && !yp)) return vfs_set_error(VFS_ERR_ARG, "stem_pool_bn_bwd_reduce: null buffer");
  StemBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gp = gp; a.yp = yp; a.idx = idx; a.x = x; a.xp = xpool; a.bnp = bnp; a.partial = partial;
  a.N = N; a.H = H; a.W = W; a.C = C; a.Hp = Hp; a.Wp = Wp; a.npg = npg; a.ppb = ppb;
  const long long P = (long long)N * Hp * Wp;
  return vfs_stem_pool_bn_bwd_reduce_launch(a, (int)((P + ppb - 1) / ppb), S(stream));
}
int vfs_stem_pool_bn_bwd_apply(const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx, const vfs_bf16* x, const float* bnp,
                               const double* sums, vfs_bf16* dx, int N, int H, int W, int C, int Hp, int Wp, int npg,
                               double count, vfs_stream_t stream) {
  StemBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gp = gp; a.yp = yp; a.idx = idx; a.x = x; a.bnp = bnp; a.sums = sums; a.dx = dx;
  a.N = N; a.H = H; a.W = W; a.C = C; a.Hp = Hp; a.Wp = Wp; a.npg = npg; a.count = count;
  return vfs_stem_pool_bn_bwd_apply_launch(a, S(stream));
}
int vfs_stem_wgrad_fused(const vfs_bf16* x4, const vfs_bf16* xraw, const vfs_bf16* gp, const vfs_bf16* yp, const uint8_t* idx,
                         const float* bnp, const double* sums, float* partial, float* grad, int N, int Hin, int Win, int Ho,
                         int Wo, int Hp, int Wp, int npg, double count, int nblocks, vfs_stream_t stream) {
  StemBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gp = gp; a.yp = yp; a.idx = idx; a.x = xraw; a.bnp = bnp; a.sums = sums;
  a.N = N; a.H = Ho; a.W = Wo; a.C = 64; a.Hp = Hp; a.Wp = Wp; a.npg = npg; a.count = count;
  if ((size_t)N * Hin * Win * 8 >= 0xFFFFFFF0ull) return vfs_set_error(VFS_ERR_SHAPE, "stem_wgrad_fused: input >= 4 GiB");
  int rc = vfs_stem_wgrad_fused_launch(a, x4, Hin, Win, partial, nblocks, S(stream));
  if (rc || !grad) return rc;
  return vfs_wgrad_reduce_launch(partial, grad, nblocks, 64, 224, 3, 7, 7, 1, S(stream));
}
int vfs_bn_param_grad(const double* sums, float* dgamma, float* dbeta, int G, int C, vfs_stream_t stream) {
  return vfs_bn_param_grad_launch(sums, dgamma, dbeta, G, C, S(stream));
}

int vfs_avgpool_fwd(const vfs_bf16* x, vfs_bf16* y, int N, int HW, int C, vfs_stream_t stream) {
  return vfs_avgpool_fwd_launch(x, y, N, HW, C, S(stream));
}
int vfs_avgpool_bwd(const vfs_bf16* g, vfs_bf16* gx, int N, int HW, int C, vfs_stream_t stream) {
  return vfs_avgpool_bwd_launch(g, gx, N, HW, C, S(stream));
}

int vfs_cosine_loss_fwd(const vfs_bf16* p1, const vfs_bf16* z1, const vfs_bf16* p2, const vfs_bf16* z2, float* loss, int N,
                        int C, int T, int K, int negative, float weight, vfs_stream_t stream) {
  if (N % T) return vfs_set_error(VFS_ERR_SHAPE, "cosine_loss: N % T");
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.p1 = p1; a.z1 = z1; a.p2 = p2; a.z2 = z2; a.loss = loss; a.N = N; a.C = C; a.T = T; a.K = K; a.negative = negative;
  a.weight = weight;
  return vfs_cosine_loss_fwd_launch(a, S(stream));
}
int vfs_xcorr_fwd(const vfs_bf16* z, const vfs_bf16* x, float* out, int nz, int nx, int Hz, int Wz, int H, int W, int C, float scale,
                  vfs_stream_t stream) {
  if (!z || !x || !out) return vfs_set_error(VFS_ERR_ARG, "xcorr_fwd: null buffer");
  XcorrArgs a;
  a.z = z; a.x = x; a.out = out; a.nz = nz; a.nx = nx; a.Hz = Hz; a.Wz = Wz; a.H = H; a.W = W; a.C = C; a.scale = scale;
  return vfs_xcorr_fwd_launch(a, S(stream));
}
int vfs_xcorr_bwd(const vfs_bf16* z, const vfs_bf16* x, const float* g, vfs_bf16* dz, vfs_bf16* dx, int nz, int nx, int Hz, int Wz, int H, int W,
                  int C, float scale, vfs_stream_t stream) {
  if (!z || !x || !g || (!dz && !dx)) return vfs_set_error(VFS_ERR_ARG, "xcorr_bwd: null buffer");
  XcorrBwdArgs a;
  a.z = z; a.x = x; a.g = g; a.dz = dz; a.dx = dx; a.nz = nz; a.nx = nx; a.Hz = Hz; a.Wz = Wz; a.H = H; a.W = W; a.C = C; a.scale = scale;
  return vfs_xcorr_bwd_launch(a, S(stream));
}
int vfs_siamfc_loss(const float* responses, const float* labels, float* loss, float* grad, int n, int mode, float param, float scale,
                    vfs_stream_t stream) {
  if (!responses || !labels || !loss) return vfs_set_error(VFS_ERR_ARG, "siamfc_loss: null buffer");
  return vfs_siamfc_loss_launch(responses, labels, loss, grad, n, mode, param, scale, S(stream));
}
int vfs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, vfs_stream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq) return vfs_set_error(VFS_ERR_ARG, "adam_step: null buffer");
  return vfs_adam_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, S(stream));
}
int vfs_loss_means(const float* loss, float* means, int K, int N, vfs_stream_t stream) {
  if (!loss || !means) return vfs_set_error(VFS_ERR_ARG, "loss_means: null buffer");
  return vfs_loss_means_launch(loss, means, K, N, S(stream));
}
int vfs_cosine_loss_bwd(const vfs_bf16* p1, const vfs_bf16* z1, const vfs_bf16* p2, const vfs_bf16* z2, const float* gloss,
                        vfs_bf16* dp1, vfs_bf16* dp2, int N, int C, int T, int K, int negative, float weight,
                        vfs_stream_t stream) {
  if (N % T) return vfs_set_error(VFS_ERR_SHAPE, "cosine_loss: N % T");
  LossArgs a;
  memset(&a, 0, sizeof(a));
  a.p1 = p1; a.z1 = z1; a.p2 = p2; a.z2 = z2; a.gloss = gloss; a.dp1 = dp1; a.dp2 = dp2;
  a.N = N; a.C = C; a.T = T; a.K = K; a.negative = negative; a.weight = weight;
  return vfs_cosine_loss_bwd_launch(a, S(stream));
}

int vfs_bn_reduce_partials_xchg(const float* partial, double* sums, double* scratch, int G, int bpg, int C, const void* peers, int rank,
                                int world, void* state, long long spin_limit, vfs_stream_t stream) {
  if (!partial || !sums || !peers || !state || spin_limit <= 0) return vfs_set_error(VFS_ERR_ARG, "bn_reduce_partials_xchg: bad argument");
  const P2PTail x = make_tail(peers, rank, world, state, spin_limit);
  return vfs_bn_reduce_partials_launch(partial, sums, scratch, G, bpg, C, S(stream), &x);
}
int vfs_bn_bwd_sums_paramgrad_xchg(const float* partial, double* sums, double* scratch, float* dgamma, float* dbeta, int G, int bpg, int C,
                                   const void* peers, int rank, int world, void* state, long long spin_limit, vfs_stream_t stream) {
  if (!partial || !sums || !peers || !state || spin_limit <= 0) return vfs_set_error(VFS_ERR_ARG, "bn_bwd_sums_paramgrad_xchg: bad argument");
  const P2PTail x = make_tail(peers, rank, world, state, spin_limit);
  return vfs_bn_reduce_fused_launch(1, partial, sums, scratch, G, bpg, C, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, 0.f, 0.f,
                                    dgamma, dbeta, S(stream), &x);
}
int vfs_p2p_window_bytes(long long* bytes, int* max_doubles, int* max_world) {
  if (!bytes || !max_doubles || !max_world) return vfs_set_error(VFS_ERR_ARG, "p2p_window_bytes: null");
  return vfs_p2p_window_bytes_host(bytes, max_doubles, max_world);
}
int vfs_p2p_alloc(void** window) { return window ? vfs_p2p_alloc_host(window) : vfs_set_error(VFS_ERR_ARG, "p2p_alloc: null"); }
int vfs_p2p_free(void* window) { return window ? vfs_p2p_free_host(window) : VFS_OK; }
int vfs_p2p_export(void* window, void* handle64) {
  return (window && handle64) ? vfs_p2p_export_host(window, handle64) : vfs_set_error(VFS_ERR_ARG, "p2p_export: null");
}
int vfs_p2p_import(const void* handle64, void** window) {
  return (window && handle64) ? vfs_p2p_import_host(handle64, window) : vfs_set_error(VFS_ERR_ARG, "p2p_import: null");
}
int vfs_p2p_unimport(void* window) { return window ? vfs_p2p_unimport_host(window) : VFS_OK; }
int vfs_p2p_allreduce_f64(double* buf, int n, const void* peers, int rank, int world, void* state, int phase, long long spin_limit,
                          vfs_stream_t stream) {
  if (!buf || !peers || !state || !(phase & 3) || spin_limit <= 0) return vfs_set_error(VFS_ERR_ARG, "p2p_allreduce_f64: bad argument");
  return vfs_p2p_allreduce_f64_launch(buf, n, reinterpret_cast<void* const*>(peers), rank, world,
                                      reinterpret_cast<unsigned long long*>(state), phase, (unsigned long long)spin_limit, S(stream));
}

int vfs_simloss_colnorm(const float* x, float* inv, int B, int C, int S, vfs_stream_t stream) {
  if (!x || !inv || B <= 0 || C <= 0 || S <= 0) return vfs_set_error(VFS_ERR_ARG, "simloss_colnorm: bad argument");
  return vfs_simloss_colnorm_launch(x, inv, B, C, S, S(stream));
}
int vfs_simloss_fwd(const float* a, const float* l, const float* inva, const float* invl, const float* mask, float* partial, float* loss,
                    int B, int C, int Sa, int Sl, int pairwise, int negative, float weight, vfs_stream_t stream) {
  if (!a || !l || !partial || !loss || B <= 0 || C <= 0 || Sa <= 0 || Sl <= 0) return vfs_set_error(VFS_ERR_ARG, "simloss_fwd: bad argument");
  if (!pairwise && Sa != Sl) return vfs_set_error(VFS_ERR_SHAPE, "simloss_fwd: non-pairwise operands must have the same positions");
  if (!pairwise && mask) return vfs_set_error(VFS_ERR_ARG, "simloss_fwd: a mask needs pairwise (sim_loss.py:46-47)");
  return vfs_simloss_fwd_launch(a, l, inva, invl, mask, partial, loss, B, C, Sa, Sl, pairwise, negative, weight, S(stream));
}
int vfs_simloss_bwd(const float* other, const float* invo, const float* mask, int mask_transposed, const float* gloss, float* d, int B,
                    int C, int Sself, int Sother, int pairwise, int negative, float weight, vfs_stream_t stream) {
  if (!other || !gloss || !d || B <= 0 || C <= 0 || Sself <= 0 || Sother <= 0) return vfs_set_error(VFS_ERR_ARG, "simloss_bwd: bad argument");
  if (!pairwise && Sself != Sother) return vfs_set_error(VFS_ERR_SHAPE, "simloss_bwd: non-pairwise operands must have the same positions");
  return vfs_simloss_bwd_launch(other, invo, mask, mask_transposed, gloss, d, B, C, Sself, Sother, pairwise, negative, weight, S(stream));
}
int vfs_simloss_norm_bwd(const float* x, const float* inv, const float* d, float* dx, int B, int C, int S, vfs_stream_t stream) {
  if (!x || !d || !dx || B <= 0 || C <= 0 || S <= 0) return vfs_set_error(VFS_ERR_ARG, "simloss_norm_bwd: bad argument");
  return vfs_simloss_norm_bwd_launch(x, inv, d, dx, B, C, S, S(stream));
}

int vfs_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, float lr, float momentum,
                 float weight_decay, const void* skip_flag, vfs_stream_t stream) {
  return vfs_sgd_launch(params, grads, momentum_buf, n, lr, momentum, weight_decay, static_cast<const unsigned long long*>(skip_flag), S(stream));
}
int vfs_scale(float* x, long long n, float scale, vfs_stream_t stream) { return vfs_scale_launch(x, n, scale, S(stream)); }
int vfs_f32_to_bf16(const float* src, vfs_bf16* dst, long long n, float scale, vfs_stream_t stream) {
  return vfs_f32_to_bf16_launch(src, dst, n, scale, S(stream));
}
int vfs_bf16_to_f32(const vfs_bf16* src, float* dst, long long n, vfs_stream_t stream) { return vfs_bf16_to_f32_launch(src, dst, n, S(stream)); }

int vfs_l2norm_rows(const vfs_bf16* x, vfs_bf16* y, long long P, int C, vfs_stream_t stream) {
  return vfs_l2norm_rows_launch(x, y, P, C, S(stream));
}
int vfs_labelprop_workspace_bytes(int H, int W, long long* bytes) {
  if (!bytes || H <= 0 || W <= 0) return vfs_set_error(VFS_ERR_ARG, "labelprop_workspace_bytes: bad argument");
  *bytes = (long long)LP_MAX_SPLIT * H * W * 10 * 8;
  return VFS_OK;
}
static int lp_workspace_ok(const void* workspace, long long workspace_bytes, int H, int W) {
  long long need = 0;
  if (vfs_labelprop_workspace_bytes(H, W, &need) != VFS_OK) return 0;
  return workspace != nullptr && workspace_bytes >= need;
}
int vfs_labelprop(const vfs_bf16* fbank, const float* sbank, float* out, void* workspace, long long workspace_bytes, int qframe,
                  const int* kslot, int nkeys, int H, int W, int C, int CO, int radius, int non_mask_len, int topk, float temperature,
                  vfs_stream_t stream) {
  if (nkeys < 1 || nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop: 1 <= nkeys <= 64");
  if (!lp_workspace_ok(workspace, workspace_bytes, H, W))
    return vfs_set_error(VFS_ERR_ARG, "labelprop: workspace smaller than vfs_labelprop_workspace_bytes(H, W)");
  if (non_mask_len < 0 || non_mask_len >= nkeys + (radius <= 0)) return vfs_set_error(VFS_ERR_ARG, "labelprop: 0 <= non_mask_len < nkeys");
  LabelPropArgs a;
  a.fbank = fbank; a.sbank = sbank; a.out = out; a.qframe = qframe; a.nkeys = nkeys;
  a.pval = (float*)workspace;
  a.pidx = workspace ? (int*)((float*)workspace + (size_t)LP_MAX_SPLIT * H * W * 10) : nullptr;
  for (int i = 0; i < LP_MAX_KEYS; ++i) a.kslot[i] = i < nkeys ? kslot[i] : 0;   // kslot is a HOST array
  a.H = H; a.W = W; a.C = C; a.CO = CO; a.radius = radius; a.non_mask_len = non_mask_len; a.topk = topk; a.inv_temp = 1.0f / temperature;
  return vfs_labelprop_launch(a, S(stream));
}
int vfs_seg_postprocess(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                        vfs_stream_t stream) {
  return vfs_seg_postprocess_launch(seg, partial, label, H, W, CO, Ho, Wo, S(stream));
}
int vfs_onehot(const uint8_t* labels, float* out, int P, int CO, vfs_stream_t stream) {
  return vfs_onehot_launch(labels, out, P, CO, S(stream));
}

// ---- fp32 evaluation path (exact_f32.hip) ----
int vfs_conv_f32_fwd(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int N, int H,
                     int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int dilation, int relu,
                     vfs_stream_t stream) {
  if (!x || !w || !y) return vfs_set_error(VFS_ERR_ARG, "conv_f32_fwd: null buffer");
  ConvF32Args a;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = KH; a.KW = KW;
  a.stride = stride; a.pad = pad; a.dil = dilation; a.relu = relu;
  return vfs_conv_f32_launch(a, S(stream));
}
int vfs_imgs_to_nhwc4_f32(const float* imgs, float* out, int B, int V, int T, int H, int W, vfs_stream_t stream) {
  return vfs_imgs_to_nhwc4_f32_launch(imgs, out, B, V, T, H, W, S(stream));
}
int vfs_maxpool_f32(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, vfs_stream_t stream) {
  return vfs_maxpool_f32_launch(x, y, N, H, W, C, Ho, Wo, S(stream));
}
int vfs_l2norm_rows_f32(const float* x, float* y, long long P, int C, vfs_stream_t stream) {
  return vfs_l2norm_rows_f32_launch(x, y, P, C, S(stream));
}
int vfs_labelprop_f32(const float* fbank, const float* sbank, float* out, void* workspace, long long workspace_bytes, int qframe,
                      const int* kslot, int nkeys, int H, int W, int C, int CO, int radius, int non_mask_len, int topk, float temperature,
                      vfs_stream_t stream) {
  if (nkeys < 1 || nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32: 1 <= nkeys <= 64");
  if (!lp_workspace_ok(workspace, workspace_bytes, H, W))
    return vfs_set_error(VFS_ERR_ARG, "labelprop_f32: workspace smaller than vfs_labelprop_workspace_bytes(H, W)");
  if (non_mask_len < 0 || non_mask_len >= nkeys + (radius <= 0)) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32: 0 <= non_mask_len < nkeys");
  LabelPropF32Args a;
  a.fbank = fbank; a.sbank = sbank; a.out = out; a.qframe = qframe; a.nkeys = nkeys;
  a.pval = (float*)workspace;
  a.pidx = workspace ? (int*)((float*)workspace + (size_t)LP_MAX_SPLIT * H * W * 10) : nullptr;
  for (int i = 0; i < LP_MAX_KEYS; ++i) a.kslot[i] = i < nkeys ? kslot[i] : 0;   // kslot is a HOST array
  a.H = H; a.W = W; a.C = C; a.CO = CO; a.radius = radius; a.topk = topk; a.temperature = temperature;
  a.non_mask_len = non_mask_len;
  return vfs_labelprop_f32_launch(a, S(stream));
}
int vfs_split_rows_bf16x2(const float* x, vfs_bf16* hl, long long P, int C, vfs_stream_t stream) {
  if (!x || !hl) return vfs_set_error(VFS_ERR_ARG, "split_rows_bf16x2: null buffer");
  return vfs_split_rows_bf16x2_launch(x, hl, P, C, S(stream));
}
static long long lp2_lists_bytes(int H, int W, int entries) { return (long long)entries * H * W * 8; }
static long long lp2_counts_bytes(int H, int W) { return ((long long)(LP2_MAX_SPLIT + 1) * H * W * 4 + 15) / 16 * 16; }      // counts + thresholds
// workspace: [dense kernel's partial lists][candidate lists: `entries` per query, shared out among the key-frame splits in use]
// [counts + thresholds][16 bytes of flags].  Round 6: the list area follows the workspace the caller passes (vfs_labelprop_f32_2pass
// derives the entries per query from workspace_bytes) - 4608 entries (237 MB at 60 x 107) is what never overflowed on the bench
// clips, fewer entries trade memory for dense redos of the first frames of a clip (cold thresholds, long lists).
int vfs_labelprop_f32_2pass_workspace_bytes_for(int H, int W, int entries_per_query, long long* bytes) {
  long long dense = 0;
  if (!bytes || entries_per_query < 16 || vfs_labelprop_workspace_bytes(H, W, &dense) != VFS_OK)
    return vfs_set_error(VFS_ERR_ARG, "labelprop_f32_2pass_workspace_bytes: bad argument (at least 16 list entries per query)");
  if (entries_per_query > LP2_MAX_SPLIT * LP2_MAX_CAP) entries_per_query = LP2_MAX_SPLIT * LP2_MAX_CAP;
  *bytes = dense + lp2_lists_bytes(H, W, entries_per_query) + lp2_counts_bytes(H, W) + 16;
  return VFS_OK;
}
int vfs_labelprop_f32_2pass_workspace_bytes(int H, int W, long long* bytes) {
  return vfs_labelprop_f32_2pass_workspace_bytes_for(H, W, LP2_MAX_SPLIT * LP2_MAX_CAP, bytes);
}
int vfs_labelprop_f32_2pass(const float* fbank, const vfs_bf16* hlbank, const float* sbank, float* out, void* workspace,
                            long long workspace_bytes, int qframe, const int* kslot, int nkeys, int H, int W, int C, int CO, int radius,
                            int non_mask_len, int topk, float temperature, int unit_rows, vfs_stream_t stream) {
  if (!hlbank || !unit_rows || !vfs_lp2_eligible(C) || (long long)H * W >= (1 << 21))      // (2^21 positions: the float index arithmetic of pass 1)
    return vfs_labelprop_f32(fbank, sbank, out, workspace, workspace_bytes, qframe, kslot, nkeys, H, W, C, CO, radius, non_mask_len, topk,
                             temperature, stream);
  if (nkeys < 1 || nkeys > LP_MAX_KEYS) return vfs_set_error(VFS_ERR_SHAPE, "labelprop_f32_2pass: 1 <= nkeys <= 64");
  if (non_mask_len < 0 || non_mask_len >= nkeys + (radius <= 0)) return vfs_set_error(VFS_ERR_ARG, "labelprop_f32_2pass: 0 <= non_mask_len < nkeys");
  long long need = 0, dense = 0;
  if (vfs_labelprop_f32_2pass_workspace_bytes_for(H, W, 16, &need) != VFS_OK || vfs_labelprop_workspace_bytes(H, W, &dense) != VFS_OK || !workspace ||
      workspace_bytes < need)
    return vfs_set_error(VFS_ERR_ARG, "labelprop_f32_2pass: workspace smaller than vfs_labelprop_f32_2pass_workspace_bytes_for(H, W, 16)");
  Lp2Args p;
  p.fbank = fbank; p.hl = hlbank; p.sbank = sbank; p.out = out;
  const long long entries = (workspace_bytes - dense - lp2_counts_bytes(H, W) - 16) / ((long long)H * W * 8);
  p.entries = (int)(entries > LP2_MAX_SPLIT * LP2_MAX_CAP ? LP2_MAX_SPLIT * LP2_MAX_CAP : entries);
  char* ws = (char*)workspace + dense;
  p.lists = (unsigned long long*)ws;
  p.counts = (int*)(ws + lp2_lists_bytes(H, W, p.entries));
  p.gthr = p.counts + (size_t)LP2_MAX_SPLIT * H * W;
  p.flags = (int*)(ws + lp2_lists_bytes(H, W, p.entries) + lp2_counts_bytes(H, W));
  p.qframe = qframe; p.nkeys = nkeys;
  for (int i = 0; i < LP_MAX_KEYS; ++i) p.kslot[i] = i < nkeys ? kslot[i] : 0;
  p.H = H; p.W = W; p.C = C; p.CO = CO; p.radius = radius; p.topk = topk; p.non_mask_len = non_mask_len; p.temperature = temperature;
  p.margin = 0.f; p.cap = 0; p.nsplit = 0; p.xcd_order = 0; p.dbg = 0;
  int rc = vfs_labelprop_f32_2pass_launch(p, S(stream));
  if (rc) return rc;
  // the dense kernel as the overflow fallback: its workgroups read the flag and leave when no list overflowed
  LabelPropF32Args a;
  a.fbank = fbank; a.sbank = sbank; a.out = out; a.qframe = qframe; a.nkeys = nkeys;
  a.pval = (float*)workspace;
  a.pidx = (int*)((float*)workspace + (size_t)LP_MAX_SPLIT * H * W * 10);
  for (int i = 0; i < LP_MAX_KEYS; ++i) a.kslot[i] = p.kslot[i];
  a.H = H; a.W = W; a.C = C; a.CO = CO; a.radius = radius; a.topk = topk; a.temperature = temperature; a.non_mask_len = non_mask_len;
  a.run_flag = p.flags;
  return vfs_labelprop_f32_launch(a, S(stream));
}
int vfs_bilinear_resize_f32(const float* src, float* dst, int C, int H, int W, int Ho, int Wo, int src_nhwc, int dst_nhwc,
                            vfs_stream_t stream) {
  if (!src || !dst) return vfs_set_error(VFS_ERR_ARG, "bilinear_resize_f32: null buffer");
  return vfs_bilinear_resize_f32_launch(src, dst, C, H, W, Ho, Wo, src_nhwc, dst_nhwc, S(stream));
}
int vfs_seg_postprocess_exact(const float* seg, float* partial, uint8_t* label, int H, int W, int CO, int Ho, int Wo,
                              vfs_stream_t stream) {
  return vfs_seg_postprocess_exact_launch(seg, partial, label, H, W, CO, Ho, Wo, S(stream));
}

int vfs_davis_counts(const uint8_t* pred, const uint8_t* gt, int* counts, void* scratch, int T, int H, int W, int nobj, int radius,
                     int use_void, vfs_stream_t stream) {
  if (T >= 3 && nobj > 0 && (!pred || !gt || !counts || !scratch)) return vfs_set_error(VFS_ERR_ARG, "davis_counts: null buffer");
  DavisArgs a;
  a.pred = pred; a.gt = gt; a.counts = counts;
  a.bp = (unsigned*)scratch;
  a.bg = scratch ? (unsigned*)scratch + (size_t)(T > 2 ? T - 2 : 0) * H * W : nullptr;
  a.T = T; a.H = H; a.W = W; a.nobj = nobj; a.radius = radius; a.use_void = use_void;
  return vfs_davis_counts_launch(a, S(stream));
}

int vfs_crop_resize_flip_norm(const uint8_t* src, const int* boxes, const uint8_t* flips, float* imgs, vfs_bf16* x4, int B, int V, int T,
                              int Hs, int Ws, int Ho, int Wo, int Wp, double mean_r, double mean_g, double mean_b, double std_r,
                              double std_g, double std_b, vfs_stream_t stream) {
  if (!src || !boxes || !flips || (!imgs && !x4)) return vfs_set_error(VFS_ERR_ARG, "crop_resize_flip_norm: null buffer");
  if (x4 && Wp < Wo) return vfs_set_error(VFS_ERR_SHAPE, "crop_resize_flip_norm: Wp < Wo");
  PipelineArgs a;
  a.src = src; a.boxes = boxes; a.flips = flips; a.imgs = imgs; a.x4 = x4;
  a.B = B; a.V = V; a.T = T; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo; a.Wp = Wp;
  a.mean[0] = mean_r; a.mean[1] = mean_g; a.mean[2] = mean_b;
  a.stdinv[0] = 1.0 / std_r; a.stdinv[1] = 1.0 / std_g; a.stdinv[2] = 1.0 / std_b;
  return vfs_crop_resize_flip_norm_launch(a, S(stream));
}

}  // extern "C"

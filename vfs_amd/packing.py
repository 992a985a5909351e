"""Host helpers shared by the engine and the tests: pack-table construction, wgrad split choice."""
import os

import numpy as np
import torch

PACK_DTYPE = np.dtype([('w', '<u8'), ('wf', '<u8'), ('wd', '<u8'), ('start', '<i8'), ('Cout', '<i4'),
                       ('Cin', '<i4'), ('KH', '<i4'), ('KW', '<i4'), ('kind', '<i4'), ('tile_start', '<i4')])
assert PACK_DTYPE.itemsize == 56
FAST_PACK = os.environ.get('VFS_FAST_PACK', '1') == '1'      # 16-byte tile variants of the weight packing (A/B switch)


def build_pack_table(entries, device):
    """entries: list of (w_fp32[Cout,Cin,KH,KW] tensor, wf bf16 tensor, wd bf16 tensor|None, kind).
    Returns (table uint8 tensor on device, ntensors, total_tiles): one workgroup per 32x32
    (cout x cin) tile of a tensor, per 256 elements for the stem, per 64x64 / 32x64 tile for the aligned 1x1 / 3x3 shapes
    (kinds 2 / 3, csrc/misc.hip)."""
    arr = np.zeros(len(entries), PACK_DTYPE)
    start = 0
    tiles = 0
    for i, (w, wf, wd, kind) in enumerate(entries):
        shp = list(w.shape) + [1, 1]
        assert kind == 1 or shp[2] * shp[3] <= 25
        aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (w, wf, wd))
        ntiles = (w.numel() + 255) // 256 if kind == 1 else ((shp[0] + 31) // 32) * ((shp[1] + 31) // 32)
        if kind == 0 and aligned and FAST_PACK:      # the two shapes that hold a ResNet's weights: 16-byte accesses (csrc/misc.hip)
            if shp[2] * shp[3] == 1 and shp[0] % 64 == 0 and shp[1] % 64 == 0:
                kind, ntiles = 2, (shp[0] // 64) * (shp[1] // 64)
            elif shp[2] == 3 and shp[3] == 3 and shp[0] % 32 == 0 and shp[1] % 64 == 0:
                kind, ntiles = 3, (shp[0] // 32) * (shp[1] // 64)
        arr[i] = (w.data_ptr(), wf.data_ptr(), 0 if wd is None else wd.data_ptr(), start,
                  shp[0], shp[1], shp[2], shp[3], kind, tiles)
        start += w.numel()
        tiles += ntiles
    t = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
    return t, len(entries), tiles


REDUCE_DTYPE = np.dtype([('partial', '<u8'), ('grad', '<u8'), ('nsplit', '<i4'), ('Cout', '<i4'), ('Ktot', '<i4'), ('Cin', '<i4'),
                         ('KH', '<i4'), ('KW', '<i4'), ('stem', '<i4'), ('block_start', '<i4'), ('nblocks', '<i4'), ('pad', '<i4')])
assert REDUCE_DTYPE.itemsize == 56


def build_reduce_table(entries, device, max_blocks_per_record=2048):
    """entries: list of (partial tensor, grad tensor, nsplit, Cout, Ktot, Cin, KH, KW, stem) -> (table uint8 tensor on the
    device, nrecords, total_blocks) for vfs_wgrad_reduce_table (csrc/conv_wgrad.hip: 128 elements per workgroup and pass)."""
    arr = np.zeros(len(entries), REDUCE_DTYPE)
    start = 0
    for i, (partial, grad, nsplit, cout, ktot, cin, kh, kw, stem) in enumerate(entries):
        nb = min(max_blocks_per_record, (cout * ktot + 127) // 128)
        arr[i] = (partial.data_ptr(), grad.data_ptr(), nsplit, cout, ktot, cin, kh, kw, stem, start, nb, 0)
        start += nb
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(entries), start


def small_map(H, W):
    """mirror of vfs_small_map (csrc/vfs_conv.h): whole images of <= 8x8 pixels, two per halo tile"""
    return H <= 8 and W <= 8 and H * W * 100 >= 64 * HALO_MIN_FILL


def wgrad_halo_eligible(N, H, W, Cin, Cout, k, stride, pad):
    """mirror of vfs_wgrad_halo_eligible (csrc/conv_wgrad_halo.hip)"""
    if k != 3 or stride != 1 or pad != 1 or Cin % 64 or Cout % 64:
        return False
    if small_map(H, W):
        return True
    cover = ((H + 7) // 8 * 8) * ((W + 15) // 16 * 16)
    return H * W * 100 >= cover * HALO_MIN_FILL


HALO_MIN_FILL = 70      # mirror of vfs_option_halo_min_fill (VFS_OPTS=halo_min_fill=.. sets both, see Engine)


def conv_halo_eligible(N, H, W, Cin, Cout, k, stride, pad):
    """mirror of vfs_conv_halo_eligible (csrc/conv_halo.hip), forward / stride-1 dgrad"""
    if k != 3 or stride != 1 or pad != 1 or Cin % 64 or Cout % 64:
        return False
    if small_map(H, W) and Cout % 128 == 0:
        return N % 2 == 0
    th, tw = (8 if Cout % 128 == 0 else 16), 16
    cover = ((H + th - 1) // th * th) * ((W + tw - 1) // tw * tw)
    return H * W * 100 >= cover * HALO_MIN_FILL


def halo_stats_rows(N, H, W, Cout):
    """statistics rows the halo kernels emit for an [N,H,W,Cout] output: one per 128 tile pixels (ragged edge tiles
    included), tiles enumerated image-major - so a group of whole images owns a contiguous block of rows"""
    if small_map(H, W) and Cout % 128 == 0:
        return N // 2
    if Cout % 128 == 0:
        return N * ((H + 7) // 8) * ((W + 15) // 16)
    return N * ((H + 15) // 16) * ((W + 15) // 16) * 2


def conv_stats_rows(N, G, H, W, Cin, Cout, k, stride, pad, Ho, Wo, halo=True):
    """rows per statistics group of the per-block (sum, sum of squares) rows a forward conv - or a stride-1 dgrad seen
    as a conv producing [N,Ho,Wo,Cout] - writes in ONE launch, or None when the groups do not own whole rows (the
    caller then launches per group): spatial tiles for the halo kernels, linear 128-pixel blocks otherwise."""
    if halo and conv_halo_eligible(N, H, W, Cin, Cout, k, stride, pad):
        total = halo_stats_rows(N, Ho, Wo, Cout)
        return total // G if (N % G == 0 and total % G == 0) else None
    mpg = (N // G) * Ho * Wo
    if G == 1 or mpg % 128 == 0:
        return (mpg + 127) // 128
    return None


def bn_fold_eligible(N, G, H, W, Cin, Cout, k, stride, pad):
    """can conv(k, stride, pad) read the RAW output of its producer unit and apply BatchNorm + ReLU while
    staging (vfs_conv_fwd_bnin / vfs_conv_wgrad_bnin)?  Both halo kernels must take the shape and a tile
    must not straddle two statistics groups (8x8 images are tiled in pairs)."""
    if k == 1 and stride == 1 and pad == 0:
        # round 6: the 1x1 / stride-1 consumer (conv2 -> conv3 of a bottleneck block) on the register-staged implicit-GEMM forward and
        # weight-gradient kernels: groups of whole 128-pixel tiles
        return (os.environ.get('VFS_BNACT_FUSE_1X1', '1') == '1' and Cin % 64 == 0 and Cout % 64 == 0 and N % G == 0 and G <= 8
                and ((N // G) * H * W) % 128 == 0)
    if not (conv_halo_eligible(N, H, W, Cin, Cout, k, stride, pad) and wgrad_halo_eligible(N, H, W, Cin, Cout, k, stride, pad)):
        return False
    if small_map(H, W) and (N // G) % 2:
        return False
    return N % G == 0 and G <= 8


def igemm_ksplit(M, Cout, Ktot, target_blocks=256):
    """Split-K plan of the implicit-GEMM kernel for problems that would not fill the chip (the head's Linear
    layers): (ksplit, workspace floats).  ksplit == 1: plain kernel, no workspace.
    The engine only uses it under VFS_KSPLIT=1: measured on MI355X the exchange of fp32 partial tiles through
    device-coherent memory costs more than the 16-workgroup launches it replaces (R50 step 11.4 -> 11.85 ms)."""
    bc = 128 if Cout % 128 == 0 else 64
    tiles = ((M + 127) // 128) * ((Cout + bc - 1) // bc)
    nk = Ktot // 64
    if tiles >= 128 or tiles > 1024 or nk < 4:
        return 1, 0
    ks = max(1, min(nk // 2, (target_blocks + tiles - 1) // tiles))      # at least two K-steps per slice
    per = (nk + ks - 1) // ks
    ks = (nk + per - 1) // per
    if ks <= 1:
        return 1, 0
    return ks, 1024 + tiles * ks * 128 * bc


def wgrad_inl_floats(nsplit, Cout, Ktot):
    """workspace (floats) of vfs_conv_wgrad_inl: the generic kernel's tiles are 128 k-columns wide"""
    return nsplit * Cout * ((Ktot + 127) // 128 * 128)


def wgrad_splits(M, Cout, Ktot, target_blocks=192, halo_geom=None):
    """Split-K plan for the wgrad kernels: (nsplit, pix_per_split).  halo_geom = (N, H, W, Cin)
    selects the plan of the 3x3 halo kernel (workgroup = 64 cin x 64 cout x 9 taps, split over
    128-pixel spatial tiles)."""
    if halo_geom is not None:
        N, H, W, Cin = halo_geom
        ntiles = ((N + 1) // 2) if small_map(H, W) else N * ((H + 7) // 8) * ((W + 15) // 16)
        colblocks = (Cin // 64) * (Cout // 64)
        # every workgroup writes a 9x64x64 fp32 partial (147 KB): keep ~2 workgroups per CU so
        # the split-K traffic (blocks x 147 KB, written then re-read) stays well below the MFMA time
        tb = int(os.environ.get('VFS_WGRAD_TB', 192))   # whole-step A/B on MI355X (the kernels run beside the dgrad chain): round 2: 256 > 128 > 512; round 6 (leaner main chain): 192 > 256 > 384 (R18 6.84 -> 6.79 ms)
        nsplit = max(1, min(ntiles, (tb + colblocks - 1) // colblocks))
        tps = (ntiles + nsplit - 1) // nsplit
        nsplit = (ntiles + tps - 1) // tps
        return nsplit, tps * 128
    target_blocks = int(os.environ.get('VFS_WGRAD_TBG', target_blocks))      # round 6 re-tune: 192 (R50 7.95 -> 7.88 ms; 160 level, 128 / 256 / 384 slower)
    nkb = (Ktot + 127) // 128
    ncb = Cout // (128 if Cout % 128 == 0 else 64)
    # round 6 A/B knobs: a target of their own for the layers with few tiles (large maps: the operands dwarf the partials, more
    # workgroups per CU hide the load latency) and for the layers with many tiles (small maps: the partials rival the operands)
    if nkb * ncb <= 4 and 'VFS_WGRAD_TBG_SMALL' in os.environ:
        target_blocks = int(os.environ['VFS_WGRAD_TBG_SMALL'])
    if nkb * ncb >= 16 and 'VFS_WGRAD_TBG_DEEP' in os.environ:
        target_blocks = int(os.environ['VFS_WGRAD_TBG_DEEP'])
    # every split writes (and wgrad_reduce re-reads) a full Cout x Ktot fp32 partial: aim for ~2-4
    # workgroups per CU, at least 8 pixel steps (512 pixels) per split, and <= 48 MB of partials
    max_split = max(1, M // 512)
    nsplit = max(1, min(max_split, (target_blocks + nkb * ncb - 1) // (nkb * ncb)))
    nsplit = max(1, min(nsplit, (48 << 20) // (Cout * Ktot * 4)))
    pps = ((M + nsplit - 1) // nsplit + 63) // 64 * 64
    nsplit = (M + pps - 1) // pps
    return nsplit, pps

"""Host helpers shared by the engine and the tests: pack-table construction, wgrad split choice."""
import numpy as np
import torch

PACK_DTYPE = np.dtype([('w', '<u8'), ('wf', '<u8'), ('wd', '<u8'), ('start', '<i8'), ('Cout', '<i4'),
                       ('Cin', '<i4'), ('KH', '<i4'), ('KW', '<i4'), ('kind', '<i4'), ('pad0', '<i4')])
assert PACK_DTYPE.itemsize == 56


def build_pack_table(entries, device):
    """entries: list of (w_fp32[Cout,Cin,KH,KW] tensor, wf bf16 tensor, wd bf16 tensor|None, kind).
    Returns (table uint8 tensor on device, ntensors, total_elems)."""
    arr = np.zeros(len(entries), PACK_DTYPE)
    start = 0
    for i, (w, wf, wd, kind) in enumerate(entries):
        shp = list(w.shape) + [1, 1]
        arr[i] = (w.data_ptr(), wf.data_ptr(), 0 if wd is None else wd.data_ptr(), start,
                  shp[0], shp[1], shp[2], shp[3], kind, 0)
        start += w.numel()
    t = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
    return t, len(entries), start


def wgrad_splits(M, Cout, Ktot, target_blocks=1024):
    """Split-K plan for the wgrad kernel: (nsplit, pix_per_split)."""
    nkb = (Ktot + 127) // 128
    ncb = Cout // (128 if Cout % 128 == 0 else 64)
    max_split = max(1, (M + 63) // 64)
    nsplit = max(1, min(max_split, (target_blocks + nkb * ncb - 1) // (nkb * ncb)))
    pps = ((M + nsplit - 1) // nsplit + 63) // 64 * 64
    nsplit = (M + pps - 1) // pps
    return nsplit, pps

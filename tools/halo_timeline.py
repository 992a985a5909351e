#!/usr/bin/env python3
"""Per-workgroup phase timeline of the halo conv (experimental build with s_memtime stamps)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import VfsLib, set_lib, get_lib
set_lib(VfsLib(os.environ['VFS_HIP_LIB']))
lib = get_lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream().cuda_stream
for (N, H, W, C) in [(256, 64, 64, 64), (256, 32, 32, 128), (256, 8, 8, 512)]:
    M = N * H * W
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    wf = torch.randn(C, 3, 3, C, device=dev).to(torch.bfloat16)
    y = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
    stats = torch.empty((M + 127) // 128 * 2 * C, device=dev)
    BC = 128 if C % 128 == 0 else 64
    TP = 128
    nwg = (M // TP) * (C // BC)
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.conv_fwd(x, wf, y, dbg.view(torch.float32), stats, N, H, W, C, H, W, C, 3, 3, 1, 1, s)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 16)
    n = int((d[0] != 0).sum())
    d = d[:, :n].astype(np.float64)
    dur = np.diff(d, axis=1)
    print((N, H, W, C), 'WGs', nwg, 'stamps', n, 'kernel span cycles', d[:, -1].max() - d[:, 0].min())
    print('  mean phase cycles:', np.round(dur.mean(0)).astype(int).tolist())
    print('  p90  phase cycles:', np.round(np.percentile(dur, 90, axis=0)).astype(int).tolist())
    print('  mean WG lifetime', int((d[:, -1] - d[:, 0]).mean()), ' concurrency estimate', (d[:, -1] - d[:, 0]).sum() / (d[:, -1].max() - d[:, 0].min()) / 256)

#!/usr/bin/env python3
"""Per-workgroup phase timeline of the persistent c64 conv (experimental build with s_memtime stamps)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import VfsLib, set_lib, get_lib
set_lib(VfsLib(os.environ['VFS_HIP_LIB']))
lib = get_lib()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream().cuda_stream
N, H, W, C = 256, 64, 64, 64
M = N * H * W
x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
wf = torch.randn(C, 3, 3, C, device=dev).to(torch.bfloat16)
y = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
stats = torch.empty((M + 127) // 128 * 2 * C, device=dev)
dbg = torch.zeros(256 * 4 * 64, dtype=torch.int64, device=dev)
for it in range(3):
    dbg.zero_()
    lib.conv_fwd(x, wf, y, dbg.view(torch.float32), stats, N, H, W, C, H, W, C, 3, 3, 1, 1, s)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(256, 4, 64).astype(np.float64)
print('prologue cycles (mean):', (d[:, :, 1] - d[:, :, 0]).mean())
K = 8
per = d[:, :, 2:2 + K * 7].reshape(256, 4, 7, K)   # taps done | barrier A | staged | rows stored | row-reduced | stats barrier | epilogue done | B+store+C
start = np.concatenate([d[:, :, 1:2], per[:, :, :-1, K - 1]], axis=2)
names = ['taps', 'waitA', 'stage', 'gstore', 'dpp', 'statbar', 'statwr', 'B+st+C']
for w in range(4):
    prev = start[:, w]
    out = []
    for k in range(K):
        out.append(f'{names[k]} {(per[:, w, :, k] - prev).mean():6.0f}')
        prev = per[:, w, :, k]
    print(f'wave {w}: ' + '  '.join(out))
print('tile period', (per[:, 0, 1:, K - 1] - per[:, 0, :-1, K - 1]).mean())

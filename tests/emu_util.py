"""Helpers for running the HIP kernels on the CPU through the fiber emulator (tests only)."""
import functools
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


@functools.lru_cache(maxsize=1)
def emu_lib():
    from vfs_amd import build
    from vfs_amd._lib import VfsLib
    return VfsLib(build.build_emu())


def bf(x):
    return x.to(torch.bfloat16)


def nhwc(x):  # [N,C,H,W] fp32 -> NHWC bf16
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def nchw(x):  # NHWC bf16 -> [N,C,H,W] fp32
    return x.float().permute(0, 3, 1, 2).contiguous()


def rb(x):  # round to bf16, keep fp32
    return x.to(torch.bfloat16).float()


def relerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def pack_relu_mask(y):
    """the bit-packed ReLU mask of a [M][C] activation as the kernels lay it out (csrc/vfs_common.h mask8_index):
    bit i of a byte <-> channel c + i positive; bytes slab-major, uint8 [C/64][M][8] ([M][C/8] when C < 64)"""
    import numpy as np
    import torch
    y = y.float().reshape(-1, y.shape[-1])
    M, C = y.shape
    bits = np.packbits((y > 0).numpy().reshape(M, C // 8, 8), axis=2, bitorder='little').reshape(M, C // 8)
    if C >= 64:
        bits = bits.reshape(M, C // 64, 8).transpose(1, 0, 2)
    return torch.from_numpy(np.ascontiguousarray(bits).reshape(-1))

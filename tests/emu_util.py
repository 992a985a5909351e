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

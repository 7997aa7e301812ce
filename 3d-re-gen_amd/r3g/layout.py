"""Operand layouts of the attention kernel (csrc/kernels.h), for callers that build Q / K / V^T themselves
(r3g_op_attention: parity tests and micro-benchmarks; the model path produces them in the QKV projection's epilogue)."""
import torch


def vt_key_positions(n, device=None):
    """position of key k inside a V^T row (kernels.h vt_key_pos): within every aligned group of 16 keys the two middle
    4-key blocks are swapped, so that the 8 keys a lane feeds into one PV MFMA are one contiguous 16-byte chunk"""
    k = torch.arange(n, device=device)
    p = k & 15
    return (k & ~15) | (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1)


def make_vt(v, lk_pad):
    """v [..., Lk, 64] (any float dtype) -> V^T bf16 [..., 64, lk_pad] in the kernel's key order; padded keys are zero"""
    lk = v.shape[-2]
    if lk_pad % 16 or lk_pad < lk:
        raise ValueError("lk_pad must be a multiple of 16 and >= Lk")
    out = torch.zeros(v.shape[:-2] + (64, lk_pad), dtype=torch.bfloat16, device=v.device)
    out[..., vt_key_positions(lk, v.device)] = v.to(torch.bfloat16).transpose(-1, -2)
    return out


def read_vt(vt, lk):
    """inverse of make_vt: V^T in the kernel's key order -> v [..., Lk, 64]"""
    return vt[..., vt_key_positions(lk, vt.device)].transpose(-1, -2)

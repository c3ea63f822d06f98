"""Kernel-side weight layouts (built once from the reference-format state_dict tensors)."""
import torch


def pack_conv3x3(w):
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin] with k = (ky*3 + kx)*Cin + c (pm_conv2d_3x3)."""
    cout, cin = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()


def pack_conv_t3(w):
    """Conv3d [Cout, Cin, 3, 1, 1] -> [Cout, 3*Cin] with k = kt*Cin + c (pm_conv_temporal_k3)."""
    cout, cin = w.shape[:2]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3 * cin).contiguous()


def geglu_perm(n_out):
    """Row permutation for GEGLU.proj [2*n_out, K]: per 32 packed rows, 16 value rows then the 16
    matching gate rows (pm_gemm PM_ACT_GEGLU combines the two halves in registers)."""
    assert n_out % 16 == 0
    idx = torch.arange(n_out).reshape(n_out // 16, 1, 16)
    return torch.cat([idx, idx + n_out], dim=1).reshape(-1)


def pack_geglu(w, b):
    perm = geglu_perm(w.shape[0] // 2)
    return w[perm].contiguous(), b[perm].contiguous()

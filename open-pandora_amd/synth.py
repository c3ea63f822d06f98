"""Deterministic weight / input synthesiser (no checkpoint and no dataset is reachable offline).

Values come from a counter-based integer hash (exact integer arithmetic -> identical on every
device and torch version), mapped to a uniform distribution with a per-tensor standard deviation:
  * matrices / conv kernels: std = 1 / sqrt(fan_in)  (every tensor, INCLUDING the ones the
    reference zero-initialises - with those left at zero the U-Net output is exactly 0 and any
    parity check is vacuous);
  * norm scales: 1 + 0.1 u;  norm shifts / biases: 0.05 u;
  * every parameter is snapped to the bf16 grid (and is f16-exact), like a bf16-trained checkpoint.
"""
import zlib

import torch

_M32 = 0xFFFFFFFF


def _hash32(x):
    """lowbias32-style avalanche on int64 tensors holding values < 2^32 (wrapping multiplies keep
    the low 32 bits exact)."""
    x = (x ^ (x >> 16)) & _M32
    x = (x * 0x7FEB352D) & _M32
    x = (x ^ (x >> 15)) & _M32
    x = (x * 0x846CA68B) & _M32
    x = (x ^ (x >> 16)) & _M32
    return x


def uniform_pm1(n, seed, name, device="cpu"):
    """n values in [-1, 1), a pure function of (seed, name, index)."""
    base = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & _M32
    idx = torch.arange(n, dtype=torch.int64, device=device)
    h = _hash32(((idx * 0x9E3779B1) & _M32) ^ base)
    h = _hash32(h + ((idx >> 32) & _M32) + 0x85EBCA6B)
    u = (h >> 8).to(torch.float32) * (1.0 / (1 << 24))  # 24 exact bits -> [0, 1)
    return u * 2.0 - 1.0


def _checkpoint_grid(v):
    """Snap to values that are exact in bf16 AND f16 (bf16 mantissa, |v| >= 2^-14 or 0), as the weights
    of a bf16-trained checkpoint are: the 16-bit product and the f32 reference then consume bit-identical
    parameters and a parity run measures the arithmetic, not a weight re-quantisation."""
    v = v.to(torch.bfloat16).to(torch.float32)
    return torch.where(v.abs() < 2.0 ** -14, torch.zeros_like(v), v)


def synth_tensor(name, shape, seed, device="cpu"):
    n = 1
    for s in shape:
        n *= s
    u = uniform_pm1(n, seed, name, device).reshape(shape)
    if len(shape) >= 2:
        fan_in = n // shape[0]
        v = u * (3.0 ** 0.5) * (fan_in ** -0.5)
    elif name.endswith("alpha"):   # learnable image-attention scale (256 yaml): tanh(alpha) + 1 well away from 1
        v = 0.8 * u
    elif name.endswith("weight"):  # GroupNorm / LayerNorm scale
        v = 1.0 + 0.1 * u
    else:
        v = 0.05 * u
    return _checkpoint_grid(v)


def synth_state_dict(module_or_shapes, seed=20230211, device="cpu", dtype=torch.float32):
    """Seeded replacement for every tensor of a state_dict (same keys, shapes)."""
    if hasattr(module_or_shapes, "state_dict"):
        shapes = {k: tuple(v.shape) for k, v in module_or_shapes.state_dict().items()}
    else:
        shapes = dict(module_or_shapes)
    return {k: synth_tensor(k, s, seed, device).to(dtype) for k, s in shapes.items()}


def synth_inputs(h, w, frames=16, seed=123, device="cpu", context_tokens=77 + 16 * 16, context_dim=1024):
    """Synthetic sampler inputs of SURVEY §8(d): x_T ~ U-shaped unit variance, c_concat scaled by
    the AE scale factor 0.18215, cond / uncond cross-attention contexts."""
    r3 = 3.0 ** 0.5
    n = 4 * frames * h * w
    mk = lambda name, cnt: uniform_pm1(cnt, seed, name, device) * r3
    out = {
        "x_T": mk("x_T", n).reshape(1, 4, frames, h, w),
        "c_concat": (0.18215 * mk("c_concat", n)).reshape(1, 4, frames, h, w),
        "c_crossattn": mk("c_crossattn", context_tokens * context_dim).reshape(1, context_tokens, context_dim),
        "uc_crossattn": mk("uc_crossattn", context_tokens * context_dim).reshape(1, context_tokens, context_dim),
    }
    return out


def synth_noise(shape, seed, step, device="cpu"):
    """Per-step DDIM noise (unit variance), shared by both sides of a parity run."""
    n = 1
    for s in shape:
        n *= s
    # sum of 4 uniforms ~ close to Gaussian, unit variance
    acc = sum(uniform_pm1(n, seed, f"noise/{step}/{j}", device) for j in range(4))
    return (acc * (3.0 ** 0.5) / 2.0).reshape(shape)

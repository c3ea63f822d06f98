"""Frame-sharded multi-GPU execution of the denoising step (one process per GPU, torch.distributed:
backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Rank r owns frames [r*T/N, (r+1)*T/N) of every activation; weights and the text context are
replicated.  Everything in the U-Net is per-frame except three exchanges, which this
object provides to UNetModel:

  reduce_stats   GroupNorm over (T,H,W): all-reduce of 32 x {sum, sumsq} f32 partials (256 B, latency);
  exchange_halo  temporal 3-tap conv: one boundary frame to each neighbour (point-to-point, the clip
                 ends keep zero padding);
  gather_kv      temporal self-attention: all-gather of the K|V projections along the frame axis
                 (every query frame attends to all T frames at its pixel).

The single-GPU result of the same kernels is the oracle for this mode (the reference has no
counterpart): tests/test_frame_parallel_cpu.py checks equality with world_size 2 on gloo.
"""
import torch
import torch.distributed as dist


class FrameParallel:
    def __init__(self, total_frames, ops=None, group=None):
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert total_frames % self.world == 0, f"{total_frames} frames do not shard over {self.world} ranks"
        self.total_frames = total_frames
        self.local_frames = total_frames // self.world
        self.frame_offset = self.rank * self.local_frames
        self.backend = dist.get_backend(group)
        self.calls = {"reduce_stats": 0, "exchange_halo": 0, "gather_kv": 0}

    # ---- clip-level helpers (sampler boundary) --------------------------------------------------
    def shard_frames(self, x, dim=2):
        """(1, C, T, h, w) -> this rank's frames (contiguous copy)."""
        return x.narrow(dim, self.frame_offset, self.local_frames).contiguous()

    def gather_frames(self, x_local, dim=2):
        """inverse of shard_frames: every rank receives the whole clip."""
        parts = [torch.empty_like(x_local) for _ in range(self.world)]
        dist.all_gather(parts, x_local.contiguous(), group=self.group)
        return torch.cat(parts, dim=dim)

    # ---- the three in-forward exchanges ---------------------------------------------------------
    def reduce_stats(self, partial, local_count):
        """partial f32 [NI, groups, 2] local {sum, sumsq} -> (all-rank totals, total element count)."""
        self.calls["reduce_stats"] += 1
        tot = partial.contiguous().clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return tot, float(local_count) * self.world

    def exchange_halo(self, t, P):
        """t [F_local*P, C]: returns (frame before my first, frame after my last) or None at clip ends."""
        self.calls["exchange_halo"] += 1
        first = t[:P].contiguous()
        last = t[(self.local_frames - 1) * P:].contiguous()
        lo = torch.empty_like(first) if self.rank > 0 else None
        hi = torch.empty_like(first) if self.rank < self.world - 1 else None
        ops = []
        if self.rank > 0:
            ops += [dist.P2POp(dist.isend, first, self.rank - 1, self.group),
                    dist.P2POp(dist.irecv, lo, self.rank - 1, self.group)]
        if self.rank < self.world - 1:
            ops += [dist.P2POp(dist.isend, last, self.rank + 1, self.group),
                    dist.P2POp(dist.irecv, hi, self.rank + 1, self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return lo, hi

    def gather_kv(self, qkv, inner, P):
        """qkv [F_local, P, 3*inner] (q|k|v): returns k, v views [T, P, inner] over all frames."""
        self.calls["gather_kv"] += 1
        kv_local = qkv[..., inner:].contiguous()
        kv_all = torch.empty((self.total_frames,) + tuple(kv_local.shape[1:]), dtype=kv_local.dtype,
                             device=kv_local.device)
        if self.backend == "gloo":
            parts = list(kv_all.chunk(self.world, dim=0))
            dist.all_gather(parts, kv_local, group=self.group)
        else:
            dist.all_gather_into_tensor(kv_all, kv_local, group=self.group)
        return kv_all[..., :inner], kv_all[..., inner:]

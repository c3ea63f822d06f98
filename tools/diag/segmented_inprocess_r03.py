"""GPU, one rank on RCCL: the frame-sharded forward replayed as [HIP graph, exchange, HIP graph, ...]
(ddim._SegmentedForward) gives the eager frame-sharded result bit for bit and the unsharded result to rounding;
the exchanges (all-reduce, all-to-all) are real RCCL calls issued between the graphs.  (More than one rank
needs more than one GPU: the multi-rank equality of the same host code is tests/test_frame_parallel_cpu.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import golden_recipe as gr
from open_pandora_amd import synth
from open_pandora_amd.ddim import DDIMSampler, _SegmentedForward
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.frame_parallel import FrameParallel
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW

pytestmark = pytest.mark.gpu


def _build(ops, fp):
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    m.bind(ops, fp)
    return LatentVisualDiffusion(m)


def _sample(pm, S=3, eta=0.0):
    ins, cond, uc = gr.sampler_inputs(8, 8)
    dev = lambda d: {k: [v.cuda() for v in lst] for k, lst in d.items()}
    ns = gr.noises(ins["x_T"].shape, S)
    smp = DDIMSampler(pm)
    y, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=dev(cond), verbose=False,
                      unconditional_guidance_scale=4.0, unconditional_conditioning=dev(uc), eta=eta,
                      fs=torch.tensor([15]).cuda(), timestep_spacing="uniform_trailing", x_T=ins["x_T"].cuda(),
                      noise_fn=lambda i, shape: ns[i].cuda())
    return y.float().cpu(), smp


def test_segmented_graph_replay_of_frame_sharded_forward(hip_ops_factory, monkeypatch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        ops = hip_ops_factory(torch.float16)
        fp = FrameParallel(16)
        assert fp.backend == "nccl" and fp.world == 1
        pm = _build(ops, fp)
        seg, smp = _sample(pm)
        graphs = [g for g in smp._graphs.values() if isinstance(g, _SegmentedForward)]
        assert len(graphs) == 2 and not smp._seg_failed  # the cond and the uncond forward
        n_comm = sum(1 for st in graphs[0].steps if not isinstance(st, torch.cuda.CUDAGraph))
        # at one rank the temporal convs have no neighbour to exchange with: 17 all-reduces + 34 all-to-alls remain
        assert n_comm == 17 + 34 and len(graphs[0].steps) == 2 * n_comm + 1
        calls = dict(fp.calls)
        monkeypatch.setenv("PANDORA_SEGMENT_GRAPHS", "0")
        eager, smp2 = _sample(pm)
        assert not smp2._graphs
        assert torch.equal(seg, eager)
        # replays re-issue the recorded exchanges without walking the Python forward: the counters only see the
        # warm-up + recording passes of the first run (2 branches x 2 passes), the eager run every forward
        assert calls["reduce_stats"] == 17 * 2 * 2 and fp.calls["reduce_stats"] == calls["reduce_stats"] + 17 * 2 * 3
        # (the unsharded comparison run with the SAME Upsample form as the sharded forward - frame shards keep the gathered
        # conv, DESIGN.md section 6 - so that the two differ by the order of the statistics sums only)
        monkeypatch.setattr(ops, "upsample_presplit", False)
        plain, _ = _sample(_build(ops, None))
        assert ((seg - plain).norm() / plain.norm()).item() < 2e-3
    finally:
        dist.destroy_process_group()

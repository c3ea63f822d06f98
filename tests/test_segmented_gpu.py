"""GPU, one rank on RCCL: the frame-sharded forward replayed as [HIP graph, exchange, HIP graph, ...]
(ddim._SegmentedForward) gives the eager frame-sharded result bit for bit and the unsharded result to rounding;
the exchanges (all-reduce, all-to-all) are real RCCL calls issued between the graphs.  (More than one rank
needs more than one GPU: the multi-rank equality of the same host code is tests/test_frame_parallel_cpu.py.)

Every test that creates a process group runs in a SPAWNED CHILD: a native abort inside RCCL / the HIP
runtime is then a failed assertion that carries the child's stderr (faulthandler + C++ stack traces enabled there),
never the death of the pytest session."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(ops, fp):
    from oracle import golden_recipe as gr
    from open_pandora_amd import synth
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    from open_pandora_amd.unet import UNetModel
    from test_oracle_golden import RH_KW
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    m.bind(ops, fp)
    return LatentVisualDiffusion(m)


def _sample(pm, S=3, eta=0.0, T=16):
    from oracle import golden_recipe as gr
    from open_pandora_amd import synth
    from open_pandora_amd.ddim import DDIMSampler
    if T == 16:
        ins, cond, uc = gr.sampler_inputs(8, 8)
    else:  # a clip of T frames = one rank's share of a 16 / T-way frame split (bench.py --emulate-shard): 16 image tokens per frame
        ins = synth.synth_inputs(8, 8, T, seed=gr.INPUT_SEED, context_tokens=77 + 16 * T)
        cond = {"c_crossattn": [ins["c_crossattn"]], "c_concat": [ins["c_concat"]]}
        uc = {"c_crossattn": [ins["uc_crossattn"]], "c_concat": [ins["c_concat"]]}
    dev = lambda d: {k: [v.cuda() for v in lst] for k, lst in d.items()}
    ns = gr.noises(ins["x_T"].shape, S)
    smp = DDIMSampler(pm)
    y, _ = smp.sample(S=S, batch_size=1, shape=(4, T, 8, 8), conditioning=dev(cond), verbose=False,
                      unconditional_guidance_scale=4.0, unconditional_conditioning=dev(uc), eta=eta,
                      fs=torch.tensor([15]).cuda(), timestep_spacing="uniform_trailing", x_T=ins["x_T"].cuda(),
                      noise_fn=lambda i, shape: ns[i].cuda())
    return y.float().cpu(), smp


def _rccl_worker(rank, port, out, kv_gather, T=16):
    import faulthandler
    faulthandler.enable(all_threads=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCH_SHOW_CPP_STACKTRACES="1")
    import torch.distributed as dist
    from open_pandora_amd.ddim import _SegmentedForward
    from open_pandora_amd.frame_parallel import FrameParallel
    from open_pandora_amd.ops_hip import HipOps
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    res = {}
    try:
        ops = HipOps(torch.float16, "cuda:0")
        fp = FrameParallel(T, kv_gather=kv_gather)
        res["backend"], res["world"] = fp.backend, fp.world
        probe = torch.ones(4, device="cuda")
        dist.all_reduce(probe)
        res["rccl_ranks_seen"] = int(probe[0].item())  # (a real communicator answered)
        pm = _build(ops, fp)
        seg, smp = _sample(pm, T=T)
        graphs = [g for g in smp._graphs.values() if isinstance(g, _SegmentedForward)]
        res["n_graphs"], res["seg_failed"] = len(graphs), smp._seg_failed
        res["n_comm"] = sum(1 for st in graphs[0].steps if not isinstance(st, torch.cuda.CUDAGraph))
        res["n_steps"] = len(graphs[0].steps)
        res["calls_seg"] = dict(fp.calls)
        smp.close()  # graphs + their pool go BEFORE anything else touches the process group
        os.environ["PANDORA_SEGMENT_GRAPHS"] = "0"
        eager, smp2 = _sample(pm, T=T)
        res["eager_graphs"] = len(smp2._graphs)
        res["calls_eager"] = dict(fp.calls)
        plain, smp3 = _sample(_build(ops, None), T=T)
        smp3.close()
        res.update(seg=seg, eager=eager, plain=plain)
        torch.cuda.synchronize()
        torch.save(res, out)
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


# T = 2: the clip one rank of an 8-way frame split holds (BASELINE configs[3]: 16 frames over 8 GPUs) - the segments captured
# and replayed here are the 8-GPU ones (2-frame row counts through every GEMM / conv / GroupNorm, the exchanges with
# themselves over a real RCCL communicator), VERDICT r04 #5c
@pytest.mark.timeout(900)
@pytest.mark.parametrize("kv_gather,T", [(False, 16), (True, 16), (False, 2), (True, 2)],
                         ids=["reshard", "kv_gather", "reshard_2_frames_per_rank", "kv_gather_2_frames_per_rank"])
def test_segmented_graph_replay_of_frame_sharded_forward(tmp_path, kv_gather, T):
    if os.environ.get("PANDORA_HIPGRAPH", "1") == "0":
        pytest.skip("PANDORA_HIPGRAPH=0 (the eager fallback switch): there is no graph replay to test")
    out = str(tmp_path / "seg.pt")
    mp.spawn(_rccl_worker, args=(_free_port(), out, kv_gather, T), nprocs=1, join=True)
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["world"] == 1 and got["rccl_ranks_seen"] == 1
    assert got["n_graphs"] == 2 and not got["seg_failed"]  # the cond and the uncond forward
    # at one rank the temporal convs have no neighbour to exchange with: 17 all-reduces + 34 bulk exchanges remain
    # (reshard: 34 all-to-alls; kv_gather, the north-star's literal split: one K|V all-gather per temporal attention)
    assert got["n_comm"] == 17 + 34 and got["n_steps"] == 2 * got["n_comm"] + 1, got
    assert got["eager_graphs"] == 0
    assert torch.equal(got["seg"], got["eager"])
    # replays re-issue the recorded exchanges without walking the Python forward: the counters only see the
    # warm-up + recording passes of the first run (2 branches x 2 passes), the eager run every forward
    cs, ce = got["calls_seg"], got["calls_eager"]
    assert cs["reduce_stats"] == 17 * 2 * 2 and ce["reduce_stats"] == cs["reduce_stats"] + 17 * 2 * 3
    err = ((got["seg"] - got["plain"]).norm() / got["plain"].norm()).item()
    print(f"\n[parity] segmented replay over RCCL (1 rank, {'K|V all-gather' if kv_gather else 'frames<->pixels re-shard'}): "
          f"== eager sharded run bit for bit; vs unsharded {err:.2e}; {got['n_comm']} RCCL calls between {got['n_comm'] + 1} graphs")
    # (PANDORA_CFG_BATCH=1: the unsharded run is then ONE forward over both clips - other grids, other split-K plans - and
    # differs from the two sharded forwards at the bf16 level of the kernels, not bit for bit as the default two-stream form)
    assert err < (1e-2 if os.environ.get("PANDORA_CFG_BATCH", "0") == "1" else 2e-3)

"""Frame-sharded multi-GPU execution of the denoising step (one process per GPU, torch.distributed:
backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Rank r owns frames [r*T/N, (r+1)*T/N) of every activation; weights and the text context are
replicated.  Everything in the U-Net is per-frame except three exchanges, which this
object provides to UNetModel:

  exchange_stats_halo
                 one grouped point-to-point exchange per temporal-conv stage: the (T,H,W)-GroupNorm partial sums
                 (256 B to every rank of the group) travel with the raw boundary frames (to the two neighbours;
                 the clip ends keep zero padding) - 88 exchanges per forward where an all-reduce PLUS a halo
                 exchange per stage made 176;
  reduce_stats   the remaining (T,H,W) GroupNorms (one per TemporalTransformer): all-reduce of 256 B;
  exchange_halo  (the separate halo form, kept for callers that already hold global statistics);
  frames_to_pixels / pixels_to_frames
                 TemporalTransformer: one all-to-all in (after the GroupNorm) and one out (before
                 proj_out) re-shard frames <-> pixels, so its projections, both temporal
                 self-attentions and the feed-forward run locally on all T frames of a pixel slice
                 (moves 2C per token instead of the 2C(N-1) of an all-gather of K|V per attention;
                 gather_kv is kept as the all-gather form).

The single-GPU result of the same kernels is the oracle for this mode (the reference has no
counterpart): tests/test_frame_parallel_cpu.py checks equality with world sizes 2, 4 and 8 on gloo.

On the GPU (`ops` = HipOps) the two latency-class exchanges - exchange_stats_halo and reduce_stats, 105 of the 139 per
forward - go through `PeerMailbox` instead of torch.distributed: one kernel launch each, direct peer writes into
hipIpc-mapped mailboxes (csrc/peer.hip), so only the 34 bulk all-to-alls remain RCCL calls and everything between
two of them replays as one HIP graph (tests/test_peer_gpu.py: 2 and 4 processes sharing one MI355X).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import capi


class PeerUnavailable(RuntimeError):
    """The frame group cannot use peer mailboxes (raised on EVERY rank of the group, never on one alone)."""


class PeerMailbox:
    """hipIpc-mapped mailboxes of one frame group (csrc/peer.hip, include/pandora_mi355x.h `pm_peer_*`): the 256-byte
    GroupNorm partial sums to every rank and the boundary frames to the two neighbours travel as direct peer writes
    of ONE kernel launch per exchange - no RCCL call, no host wait, capturable inside a HIP graph.  Created lazily
    (and collectively) at the first exchange, sized by the largest halo frame seen; a larger one re-creates it."""

    NSTAT_MAX = 256

    def __init__(self, ops, group, rank, world):
        self.ops, self.lib, self.group, self.rank, self.world = ops, ops.lib, group, rank, world
        self.base = None
        self.halo_max = 0
        self.fine_grained = None
        self.exchanges = 0
        # bumped by every (re-)creation: captured HIP graphs hold the mailbox addresses BY VALUE inside their launch
        # parameters, so the sampler keys its graphs by this number and re-records after a re-creation
        self.generation = 0
        self.timeout_s = float(os.environ.get("PANDORA_PEER_TIMEOUT_S", "5"))

    def _create(self, halo_bytes):
        """Collective.  Either every rank of the group ends up with every mailbox mapped, or every rank raises
        PeerUnavailable (a rank that cannot allocate / export / map says so in the handle exchange instead of raising
        alone: its peers would otherwise wait for it in the collectives below)."""
        self.close()
        torch.cuda.synchronize(self.ops.device)
        self.halo_max = int(halo_bytes)
        nbytes = self.lib.pm_peer_mailbox_bytes(self.world, self.NSTAT_MAX, self.halo_max)
        base, fg = ctypes.c_void_p(), ctypes.c_int()
        handle = ctypes.create_string_buffer(64)
        rc = self.lib.pm_peer_create(nbytes, ctypes.byref(base), handle, ctypes.byref(fg))
        if rc == 0 and not fg.value and os.environ.get("PANDORA_PEER_ALLOW_COARSE", "0") != "1":
            # plain (coarse-grained) device memory: peer writes are not guaranteed to become visible inside a RUNNING
            # kernel there, and the protocol of csrc/peer.hip polls from inside one - a failed commissioning for the
            # whole group, reported through the handle exchange below like any other creation failure
            self.lib.pm_peer_destroy(base)
            rc = -2
        if rc == 0 and os.environ.get("PANDORA_PEER_INJECT_FAIL") == str(self.rank):  # fault injection (tests/test_peer_gpu.py)
            self.lib.pm_peer_destroy(base)
            rc = -1
        infos = [None] * self.world
        control_collective(lambda: dist.all_gather_object(infos, (int(rc), bytes(handle.raw)), group=self.group), self.group)
        if any(i[0] != 0 for i in infos):
            if rc == 0:
                self.lib.pm_peer_destroy(base)
            raise PeerUnavailable(f"pm_peer_create failed on rank(s) {[r for r, i in enumerate(infos) if i[0] != 0]}")
        self.base, self.fine_grained = base, bool(fg.value)
        self.generation += 1
        self.peers = (ctypes.c_void_p * self.world)()
        bad = 0
        for r in range(self.world):
            if r != self.rank:
                p = ctypes.c_void_p()
                if self.lib.pm_peer_open(infos[r][1], ctypes.byref(p)) != 0:
                    bad = 1
                    continue
                self.peers[r] = p
        flags = [None] * self.world
        # (also: every mailbox is mapped everywhere before the first write)
        control_collective(lambda: dist.all_gather_object(flags, bad, group=self.group), self.group)
        if any(flags):
            self._release()
            raise PeerUnavailable(f"pm_peer_open failed on rank(s) {[r for r, f in enumerate(flags) if f]}")

    def commission(self, halo_bytes=0):
        """Collective go / no-go at construction: map the mailboxes and run ONE exchange of known values with a short
        timeout.  -> True on every rank, or False on every rank (the group then keeps the torch.distributed form of
        the two exchanges).  A platform where peer writes / system-scope atomics between the group's devices do not
        work shows up here as a timed-out or wrong exchange instead of as a hang in the middle of a clip."""
        ok, why = 1, ""
        try:
            self._create(int(halo_bytes))  # (sized once for the largest halo frame the caller announces: no re-creation later)
        except PeerUnavailable as exc:  # (raised on every rank)
            return False, str(exc)
        keep, self.timeout_s = self.timeout_s, min(self.timeout_s, 2.0)
        try:
            mine = torch.full((8,), float(self.rank + 1), dtype=torch.float32, device=self.ops.device)
            tot, _, _ = self.exchange(mine)
            torch.cuda.synchronize(self.ops.device)
            want = float(self.world * (self.world + 1) // 2)
            self.check()
            if not bool((tot == want).all()):
                ok, why = 0, f"self-test exchange returned {tot.tolist()} instead of {want}"
        except capi.PandoraKernelError as exc:
            ok, why = 0, str(exc)
        finally:
            self.timeout_s = keep
        res = [None] * self.world
        control_collective(lambda: dist.all_gather_object(res, (ok, why), group=self.group), self.group)
        if all(r[0] for r in res):
            return True, ""
        self.close()
        return False, "; ".join(f"rank {r}: {w[1]}" for r, w in enumerate(res) if not w[0])

    def _release(self):
        for r in range(self.world):
            if r != self.rank and self.peers[r]:
                self.lib.pm_peer_close(self.peers[r])
                self.peers[r] = None
        self.lib.pm_peer_destroy(self.base)
        self.base = None

    def close(self):
        if self.base is None:
            return
        torch.cuda.synchronize(self.ops.device)
        control_collective(lambda: dist.barrier(group=self.group), self.group)  # nobody still writes into a mailbox that is about to go
        self._release()

    def check(self, collective=False):
        """Raise if any exchange since creation timed out (synchronous: once per clip, never per exchange).
        `collective=True` (the sampler, once per clip, on EVERY rank of the group): the error words are MAX-reduced over
        the group, so a time-out seen by one rank - a peer that was slow rather than dead - raises on all of them at the
        same point instead of leaving the others to hang in the next clip's collectives; the mailboxes are released
        group-wide first (the sticky error word and the fail-fast mode go with them) and the group continues on the
        torch.distributed form of the exchanges."""
        if self.base is None:
            return
        epoch, err = ctypes.c_int(), ctypes.c_int()
        capi.check(self.lib.pm_peer_status(self.base, ctypes.byref(epoch), ctypes.byref(err)), "pm_peer_status")
        bad = [self.rank] if err.value else []
        if collective:
            flags = [None] * self.world
            control_collective(lambda: dist.all_gather_object(flags, int(err.value != 0), group=self.group), self.group)
            bad = [r for r, f in enumerate(flags) if f]
            if bad:
                self.close()
        if bad:
            raise capi.PandoraKernelError(f"peer mailbox exchange timed out on rank(s) {bad} (a peer never arrived "
                                          f"within {self.timeout_s} s): the frame-sharded result is invalid")
        return epoch.value

    def exchange(self, stats, first=None, last=None):
        """stats f32 [n] -> totals (rank-order sum); first / last [P, C] frames -> (frame before my first, after my last)."""
        halo = 0 if first is None else first.numel() * first.element_size()
        if self.base is None or halo > self.halo_max:
            if torch.cuda.is_current_stream_capturing():
                raise capi.PandoraKernelError("peer mailbox must be created before graph capture (warm-up forward)")
            self._create(max(halo, self.halo_max))
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() <= self.NSTAT_MAX
        tot = torch.empty_like(stats)
        lo = hi = None
        if first is not None:
            assert first.is_contiguous() and last.is_contiguous() and first.shape == last.shape
            lo = torch.empty_like(first) if self.rank > 0 else None
            hi = torch.empty_like(first) if self.rank < self.world - 1 else None
        ptr = lambda t: None if t is None else t.data_ptr()
        rc = self.lib.pm_peer_exchange(self.base, self.peers, self.rank, self.world, ptr(stats), stats.numel(),
                                       ptr(first), ptr(last), halo, ptr(tot), ptr(lo), ptr(hi), self.NSTAT_MAX,
                                       self.halo_max, self.timeout_s, self.ops._stream())
        capi.check(rc, "pm_peer_exchange")
        self.exchanges += 1
        return tot, lo, hi


_COMM_STREAMS = {}


def _on_comm_stream(fn):
    """Run an RCCL call on the process's dedicated communication stream of the current device, joined to the caller's stream
    by events on both sides (why: FrameParallel._comm)."""
    cur = torch.cuda.current_stream()
    key = cur.device.index
    cs = _COMM_STREAMS.get(key)
    if cs is None:
        cs = _COMM_STREAMS[key] = torch.cuda.Stream(device=cur.device)
    cs.wait_stream(cur)
    with torch.cuda.stream(cs):
        fn()
    cur.wait_stream(cs)


def control_collective(fn, group=None):
    """A control-plane collective (object gathers, barriers, flags, the seed broadcast): under RCCL it goes through the
    dedicated communication stream like the data-path ones - the watchdog polls its end event too, and the stream the
    caller is on may start capturing right afterwards (FrameParallel._comm has the story)."""
    if torch.cuda.is_available() and torch.cuda.is_initialized() and dist.get_backend(group) == "nccl":
        _on_comm_stream(fn)
    else:
        fn()


def _host_staged_sync(t, group=None):
    """gloo (the CPU-test / single-GPU rehearsal backend) stages CUDA tensors through the host without
    ordering against the producing stream; RCCL ("nccl") is stream-ordered and needs nothing."""
    if t.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.current_stream(t.device).synchronize()


class FrameParallel:
    def __init__(self, total_frames, ops=None, group=None, kv_gather=False, halo_bytes=None):
        """`kv_gather`: the north-star's literal form of the temporal attention - every rank keeps its frames and
        all-gathers K|V over the frame axis (`gather_kv`) - instead of the frames <-> pixels re-shard.
        `halo_bytes`: the largest boundary frame (f32 [H*W, C]: 4*H*W*C bytes at level 0) the peer mailboxes will carry,
        so that they are sized ONCE at commissioning (default: PANDORA_PEER_HALO_BYTES, else grown - collectively - at
        the first larger exchange, which invalidates graphs recorded before: the sampler keys them by `generation`)."""
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.group = group
        self.kv_gather = kv_gather
        self.mailbox = None  # set below: PeerMailbox when `ops` drives a GPU (PANDORA_PEER_MAILBOX=0: the P2P form)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # P2POp peers are global ranks: map group-local neighbours to global ids
        self._global = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        assert total_frames % self.world == 0, f"{total_frames} frames do not shard over {self.world} ranks"
        self.total_frames = total_frames
        self.local_frames = total_frames // self.world
        self.frame_offset = self.rank * self.local_frames
        self.backend = dist.get_backend(group)
        self.calls = {"reduce_stats": 0, "exchange_halo": 0, "all_to_all": 0, "stats_halo": 0}
        # ddim._SegmentedForward while it records a forward: every in-forward exchange is handed to it as a
        # closure over buffers that were allocated BEFORE the call (so a replay re-issues the same RCCL calls on
        # the same addresses, between the HIP graphs that hold the kernels around them)
        self.recorder = None
        if self.world > 1 and ops is not None and hasattr(ops, "lib") and hasattr(ops.lib, "pm_peer_exchange") \
                and os.environ.get("PANDORA_PEER_MAILBOX", "1") != "0":
            mb = PeerMailbox(ops, group, self.rank, self.world)
            if halo_bytes is None:
                halo_bytes = int(os.environ.get("PANDORA_PEER_HALO_BYTES", "0"))
            ok, why = mb.commission(halo_bytes)  # collective: all ranks keep the mailbox or all fall back
            if ok:
                self.mailbox = mb
                self.calls["mailbox"] = 0
            else:
                import warnings
                warnings.warn("peer mailboxes are not available for this frame group (" + why + "): the GroupNorm / "
                              "halo exchanges go through torch.distributed point-to-point calls instead")

    def _comm(self, fn):
        """Issue one torch.distributed exchange - or hand it to the recording _SegmentedForward, which re-issues the same
        closure at every replay.

        RCCL ("nccl"): the call is issued on a DEDICATED stream that is joined to the caller's stream by events on both
        sides, never on the caller's stream itself.  ProcessGroupNCCL's watchdog thread polls the end event of every
        collective, and on this HIP runtime hipEventQuery fails (hipErrorCapturedEvent) for an event whose recording
        stream is capturing AT THE TIME OF THE QUERY - even if the record happened before the capture began
        (tools/diag/event_query_probe.py).  The segmented replay captures on the very stream its warm-up forward ran on:
        a watchdog poll that fell into that window raised inside the watchdog thread, and ProcessGroupNCCL then
        terminated the process (the r03 abort of tests/test_segmented_gpu.py: SIGABRT in torch.cat /
        destroy_process_group, 1 run in 4).  The dedicated stream never captures, so its events are always queryable.

        gloo stages device tensors through the host without ordering against the stream that produced them: there the
        closure itself waits for the stream first (inside the closure, so that replays do it too)."""
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            inner = fn
            if self.backend == "nccl":
                def fn():  # (buffers stay owned by the caller's stream: the comm stream is joined back before any reuse)
                    _on_comm_stream(inner)
            else:
                def fn():
                    torch.cuda.current_stream().synchronize()
                    inner()
        if self.recorder is None:
            fn()
        else:
            self.recorder.comm(fn)

    @staticmethod
    def _p2p(ops):
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    # ---- clip-level helpers (sampler boundary) --------------------------------------------------
    def shard_frames(self, x, dim=2):
        """(1, C, T, h, w) -> this rank's frames (contiguous copy)."""
        return x.narrow(dim, self.frame_offset, self.local_frames).contiguous()

    def gather_frames(self, x_local, dim=2):
        """inverse of shard_frames: every rank receives the whole clip."""
        parts = [torch.empty_like(x_local) for _ in range(self.world)]
        src = x_local.contiguous()
        rec, self.recorder = self.recorder, None  # (a sampler-level collective: never part of a recorded forward)
        try:
            self._comm(lambda: dist.all_gather(parts, src, group=self.group))
        finally:
            self.recorder = rec
        return torch.cat(parts, dim=dim)

    def all_reduce_sum(self, t):
        """sum over the frame group (sampler-level reductions, e.g. the std of rescale_noise_cfg)"""
        t = t.contiguous().clone()
        rec, self.recorder = self.recorder, None
        try:
            self._comm(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
        finally:
            self.recorder = rec
        return t

    def _mailbox_exchange(self, stats, first=None, last=None):
        """-> (totals, lo, hi) through the peer mailboxes, or None: no mailbox (then the caller's torch.distributed
        form runs).  A re-creation for a larger halo frame that fails does so on every rank (PeerUnavailable), so the
        whole group drops to the torch.distributed form at the same exchange."""
        if self.mailbox is None:
            return None
        try:
            got = self.mailbox.exchange(stats, first, last)
        except PeerUnavailable as exc:
            import warnings
            warnings.warn(f"peer mailboxes dropped for this frame group ({exc}): torch.distributed point-to-point from here on")
            self.mailbox = None
            return None
        self.calls["mailbox"] += 1
        return got

    # ---- the three in-forward exchanges ---------------------------------------------------------
    def reduce_stats(self, partial, local_count):
        """partial f32 [NI, groups, 2] local {sum, sumsq} -> (all-rank totals, total element count)."""
        self.calls["reduce_stats"] += 1
        got = self._mailbox_exchange(partial.contiguous().view(-1))
        if got is not None:
            return got[0].view_as(partial), float(local_count) * self.world
        tot = partial.contiguous().clone()
        self._comm(lambda: dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group))
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
            ops += [dist.P2POp(dist.isend, first, self._global(self.rank - 1), self.group),
                    dist.P2POp(dist.irecv, lo, self._global(self.rank - 1), self.group)]
        if self.rank < self.world - 1:
            ops += [dist.P2POp(dist.isend, last, self._global(self.rank + 1), self.group),
                    dist.P2POp(dist.irecv, hi, self._global(self.rank + 1), self.group)]
        if ops:
            self._comm(lambda: self._p2p(ops))
        return lo, hi

    def exchange_stats_halo(self, x, P, partial, local_count):
        """ONE grouped point-to-point exchange per temporal-conv stage instead of an all-reduce plus a halo
        exchange: every rank sends its (T,H,W)-GroupNorm partial sums (256 B) to every other rank of the frame
        group and its RAW boundary frames of x [F_local*P, C] to the two neighbours - raw, because the normalised
        halo needs the global statistics that travel in the same message; the receiver normalises the (at most
        two) halo frames itself with the totals it has just formed.  Totals are summed in rank order on every
        rank (bitwise identical everywhere).  -> (totals [1, groups, 2], total count, raw frame before my first
        or None, raw frame after my last or None)."""
        self.calls["stats_halo"] += 1
        part = partial.contiguous()
        first = x[:P].contiguous()
        last = x[(self.local_frames - 1) * P:].contiguous()
        got = self._mailbox_exchange(part.view(-1), first, last)  # one kernel launch: peer writes + arrival counters,
        if got is not None:                                        # the same rank-order totals
            return got[0].view_as(part), float(local_count) * self.world, got[1], got[2]
        lo = torch.empty_like(first) if self.rank > 0 else None
        hi = torch.empty_like(first) if self.rank < self.world - 1 else None
        parts = [part if r == self.rank else torch.empty_like(part) for r in range(self.world)]
        ops = []
        for r in range(self.world):
            if r != self.rank:
                ops += [dist.P2POp(dist.isend, part, self._global(r), self.group),
                        dist.P2POp(dist.irecv, parts[r], self._global(r), self.group)]
        if self.rank > 0:
            ops += [dist.P2POp(dist.isend, first, self._global(self.rank - 1), self.group),
                    dist.P2POp(dist.irecv, lo, self._global(self.rank - 1), self.group)]
        if self.rank < self.world - 1:
            ops += [dist.P2POp(dist.isend, last, self._global(self.rank + 1), self.group),
                    dist.P2POp(dist.irecv, hi, self._global(self.rank + 1), self.group)]
        if ops:
            self._comm(lambda: self._p2p(ops))
        tot = parts[0].clone()
        for r in range(1, self.world):
            tot += parts[r]
        return tot, float(local_count) * self.world, lo, hi

    def _a2a(self, src):
        dst = torch.empty_like(src)
        self._comm(lambda: dist.all_to_all_single(dst, src, group=self.group))
        return dst

    def frames_to_pixels(self, t, P):
        """[F_local*P, C] (my frames, all pixels) -> [F_total*(P/N), C] (all frames, my pixel slice).
        One all-to-all moves each element once; the whole TemporalTransformer (projections, both
        temporal self-attentions, feed-forward: all per-pixel) then runs without any exchange."""
        self.calls["all_to_all"] += 1
        N, Fl = self.world, self.local_frames
        assert P % N == 0, f"{P} pixels do not shard over {N} ranks"
        C = t.shape[1]
        src = t.view(Fl, N, P // N, C).permute(1, 0, 2, 3).contiguous()  # chunk j -> rank j
        return self._a2a(src).view(N * Fl * (P // N), C)               # [source rank = frame block, ...]

    def pixels_to_frames(self, t, P):
        """inverse of frames_to_pixels."""
        self.calls["all_to_all"] += 1
        N, Fl = self.world, self.local_frames
        C = t.shape[1]
        dst = self._a2a(t.contiguous()).view(N, Fl, P // N, C)          # [source rank = pixel slice, ...]
        return dst.permute(1, 0, 2, 3).reshape(Fl * P, C)

    def gather_kv(self, qkv, inner, P):
        """qkv [F_local, P, 3*inner] (q|k|v): returns k, v views [T, P, inner] over all frames."""
        self.calls["gather_kv"] = self.calls.get("gather_kv", 0) + 1
        kv_local = qkv[..., inner:].contiguous()
        kv_all = torch.empty((self.total_frames,) + tuple(kv_local.shape[1:]), dtype=kv_local.dtype,
                             device=kv_local.device)
        if self.backend == "gloo":
            parts = list(kv_all.chunk(self.world, dim=0))
            self._comm(lambda: dist.all_gather(parts, kv_local, group=self.group))
        else:
            self._comm(lambda: dist.all_gather_into_tensor(kv_all, kv_local, group=self.group))
        return kv_all[..., :inner], kv_all[..., inner:]


class CFGParallel:
    """Classifier-free-guidance pair parallelism: the cond and the uncond U-Net forward of a DDIM step
    (ddim.py:233-234, run back to back by the reference) go to two different ranks; ONE exchange of the
    model output per step (2.4 MB at 16x72x128) replaces ~230 in-forward collectives that the same two
    ranks would need as frame shards.  Each partner then applies the identical update to its copy of
    the latent (shared noise seed), so no second exchange is needed."""

    def __init__(self, partner, branch):
        self.partner, self.branch = partner, branch  # branch 0 = conditional, 1 = unconditional
        self.calls = 0

    def exchange(self, e_mine):
        """-> (e_cond, e_uncond)"""
        self.calls += 1
        e_mine = e_mine.contiguous()
        e_other = torch.empty_like(e_mine)
        _host_staged_sync(e_mine)
        ops = [dist.P2POp(dist.isend, e_mine, self.partner), dist.P2POp(dist.irecv, e_other, self.partner)]

        def go():
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if e_mine.is_cuda and dist.get_backend() == "nccl":
            _on_comm_stream(go)  # (RCCL calls never run on a stream that may capture: FrameParallel._comm)
        else:
            go()
        return (e_mine, e_other) if self.branch == 0 else (e_other, e_mine)


def make_hybrid(total_frames, use_cfg=True, kv_gather=False, ops=None, halo_bytes=None):
    """Decompose the world for one clip: with CFG on and an even world size, ranks [0, N/2) take the
    conditional branch and [N/2, N) the unconditional one (CFGParallel pairs r <-> r + N/2); inside a
    branch the N/2 ranks shard the frames (FrameParallel on a sub-group).  Returns (fp, cfgp); either
    may be None.  Every rank must call this (dist.new_group is collective)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return None, None
    if not use_cfg or world % 2:  # (use_cfg=False + kv_gather=True at world 8 = the north-star's split: 2 frames per GPU)
        return FrameParallel(total_frames, ops=ops, kv_gather=kv_gather, halo_bytes=halo_bytes), None
    half = world // 2
    groups = [dist.new_group(list(range(b * half, (b + 1) * half))) for b in (0, 1)]
    branch = rank // half
    cfgp = CFGParallel(partner=(rank + half) % world, branch=branch)
    fp = FrameParallel(total_frames, ops=ops, group=groups[branch], kv_gather=kv_gather, halo_bytes=halo_bytes) if half > 1 else None
    return fp, cfgp

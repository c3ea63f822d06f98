"""HipOps: the op table of the denoising path, bound to the gfx950 C-ABI kernels.

Every method takes/returns torch tensors that live on the GPU (PyTorch is only the allocator and
the stream owner) and enqueues exactly one or two kernels from libpandora_mi355x.so on torch's
current stream.  Activations are channels-last token matrices [frames*H*W, C] in f16 or bf16.
There is deliberately no CPU / eager fallback here.
"""
import math
import os

import torch

from . import capi

_DT = {torch.float16: capi.PM_F16, torch.bfloat16: capi.PM_BF16}
_F32 = capi.PM_F32


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# parity="selective": the sites whose [hi | lo] operands buy the error, from the leave-one-out table of tools/parity_sites.py
# (profiles/r05/parity_sites_320x512_leave_one_out.txt: full-width 10-step CFG-4 FRAMES against the real reference with each
# (kind, level) class taken out of the full parity configuration, and the step time it costs).  The error the [hi | lo]
# operands remove sits at the shallow levels - every transformer's GroupNorm -> proj_in (all of it for 0.5 ms), the f32
# stream into Downsample / Upsample / stem convs, the level-0/1 GroupNorm -> 3x3 conv, the level-0 temporal sites; the
# LayerNorm -> q|k|v / q / GEGLU projections of the SPATIAL blocks and everything at the 1280-channel levels buy nothing
# measurable and keep their one-rounding operands (and the fused LayerNorm + projection kernel).
SELECTIVE_PARITY_SITES = frozenset(
    [("gnp", lv) for lv in range(4)] + [("split", lv) for lv in range(3)] + [("gn3", 0), ("gn3", 1), ("gnt", 0), ("gnt", 2),
                                                                            ("lnt1", 0), ("lnt2", 0), ("lnt3", 0)])


class HipOps:
    name = "hip"
    supports_graphs = True  # every op only enqueues kernels on the current stream: capturable
    conv_t3_clips = True    # conv_t3(clip_frames=...): independent clips batched along the frame axis in one launch
    # pm_attention's single-segment kernel works in the base-2 domain on q * (64^-1/2 * log2 e).  A caller that
    # owns the q projection folds this factor into its weights (UNetModel.prepare) and calls
    # attention(..., prescaled=True); otherwise the kernel scales (and re-rounds) the q fragments itself.
    q_prescale = 64 ** -0.5 * 1.4426950408889634

    def __init__(self, dtype=torch.bfloat16, device="cuda", workspace_mb=256, fp8_attention=False, fp8_min_tokens=2048,
                 parity=False, diag=False):
        if dtype not in _DT:
            raise ValueError(f"HipOps supports float16/bfloat16 activations, got {dtype}")
        # diag=True: this op table runs on the -DPM_DIAG build of the library (kernel-variant overrides, tuning
        # switches: include/pandora_mi355x_diag.h) - measurement code and variant tests only
        self.lib = capi.load_diag() if diag else capi.load()
        # BASELINE configs[4]: the spatial self-attention of the U-Net on pm_attention_fp8 (opt-in: ~2-3e-2 per call)
        self.fp8_attention = bool(fp8_attention)
        self.fp8_min_tokens = int(fp8_min_tokens)  # shorter sequences stay on pm_attention (the packing pass costs more than it saves)
        if os.environ.get("PANDORA_Q_PRESCALE", "1") == "0":  # (numerics experiments: scale inside the kernel)
            self.q_prescale = None
        # LayerNorm + projection pairs of the shallow levels as ONE kernel (pm_ln_gemm); "0": the two-kernel pair (A/B)
        self.fused_ln = os.environ.get("PANDORA_FUSED_LN", "1") != "0"
        self.ln_pair_stream = os.environ.get("PANDORA_LN_PAIR_STREAM", "1") != "0"
        # f32 operands (the residual stream into skip 1x1 convs, Downsample / Upsample convs) are rounded by one
        # pm_split16 pass and run on the DMA-staged 16-bit kernels; "0": the register-staged f32 loaders (A/B)
        self.presplit = os.environ.get("PANDORA_PRESPLIT", "1") != "0"
        self.upsample_presplit = os.environ.get("PANDORA_UPSAMPLE_PRESPLIT", "1") != "0"  # (A/B: 0 = the gathered general mode)
        # The parity configuration (VERDICT r02 #4a): every GroupNorm / LayerNorm output - the A operand of the convs and
        # projections, whose one 16-bit rounding is 57 % of the end-to-end error^2 (tests/test_error_budget_gpu.py) - is
        # written as [hi | lo] (PM_OUT_HILO) and consumed over 2C channels with the weights walked twice (PM_FLAG_W_WRAP
        # for dense GEMMs, repeated packed weights for the 3x3 / temporal convs); the f32 stream entering Downsample /
        # Upsample convs likewise.  Twice the MFMA work on those ops: a numerics mode, not a performance mode.
        # parity=True: every site.  parity="selective" / a collection of (kind, level) sites (UNetModel._site names the site of
        # the next call in `self.site`): only those - the rest keep the one-rounding operands AND the fused LayerNorm +
        # projection kernel; calls outside the U-Net (first stage, Resampler: `site` None) take [hi | lo] whenever parity is on.
        self.site = None
        self.parity_sites = None
        if isinstance(parity, str):
            if parity != "selective":
                raise ValueError(f"parity={parity!r}: True, False, 'selective' or a collection of (kind, level) sites")
            parity = SELECTIVE_PARITY_SITES
        if isinstance(parity, int) and not isinstance(parity, bool):
            parity = bool(parity)
        if not isinstance(parity, bool) and parity is not None:
            self.parity_sites = frozenset((str(k), int(lv)) for k, lv in parity)
            parity = True
        self.parity = bool(parity)
        if self.parity and self.parity_sites is None:
            self.fused_ln = False  # (the panel kernel keeps its normalised panel as 16 bit in LDS)
        self._dup = {}  # parity: packed conv weights with every tap's channel block repeated, keyed by (data_ptr, taps)
        self.dtype = dtype
        self.dt = _DT[dtype]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.PandoraKernelError("HipOps needs a ROCm device (cuda:N)")
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        # split-K scratch for the GEMM family (deep levels: few output tiles, long K); one buffer per
        # op table is enough because every launch on the stream is ordered after the previous reduce
        self.ws_bytes = int(workspace_mb) << 20
        # ... per STREAM: the sampler runs the cond and the uncond forward of a step on two streams, and two
        # split-K calls in flight at once must not share their slabs
        self._workspaces = {}
        # r06: GroupNorm totals as int64 fixed-point limbs [NI, 32, 4, 8] that the producing kernel ADDS to with integer atomics
        # (PM_FLAG_STATS_I64 / PM_TOTALS_I64): no finalize launch between a conv and its GroupNorm (191 of ~1050 launches per
        # forward).  The buffers come from a per-stream arena that `begin_forward` zeroes with ONE memset; outside a forward (or
        # past its end) a call takes a fresh zeroed tensor.  "0": the column-sum + finalize form (A/B runs)
        self.stats_i64 = os.environ.get("PANDORA_STATS_I64", "1") != "0"
        # "2" (A/B that lost, diagnostics build only - the shipped MFMA kernels do not carry the path): also from their epilogues
        self.stats_i64_all = (os.environ.get("PANDORA_STATS_I64", "1") == "2" and os.environ.get("PANDORA_DIAG_LIB") == "1"
                              and "PANDORA_LIB" not in os.environ)
        self.stats_nsum = self.stats_i64  # groupnorm_apply sums consecutive totals entries itself (per-frame -> clip sums)
        self._arena = {}  # stream -> [int64 tensor, next free element]
        self.arena_bytes = 16 << 20

    # -- helpers ---------------------------------------------------------------------------------
    def _hilo(self):
        """Is the norm / conversion output of the CURRENT site carried as [hi | lo]?  (see `parity` in __init__)"""
        if not self.parity:
            return False
        return self.parity_sites is None or self.site is None or self.site in self.parity_sites

    def _stream(self):
        # raw handle of torch's current stream on this device; the C getter avoids building a Stream object
        # per launch (a quarter of the host-side cost of an eagerly launched forward, tools/archive/host_profile.py)
        if self._raw_stream is not None:
            return self._raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def workspace(self):
        """split-K scratch of the CURRENT stream (allocated on first use; inside a graph capture it comes from
        the graph's pool, so warm every stream up before capturing)"""
        if not self.ws_bytes:
            return None
        key = self._stream()
        ws = self._workspaces.get(key)
        if ws is None:
            ws = self._workspaces[key] = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        return ws

    def _rows(self, t, f32_ok=False):
        """(row stride in elements) of a 2-D, last-dim-contiguous view."""
        assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
        assert t.dtype == self.dtype or (f32_ok and t.dtype == torch.float32), t.dtype
        return t.stride(0)

    def _in_dt(self, t):
        return _F32 if t.dtype == torch.float32 else self.dt

    def _gemm_io(self, a, residual, out, M, n_out, stream):
        """flags + output allocation shared by the GEMM family.  `stream=True`: the result belongs to
        the f32 residual stream (f32 output and f32 residual); an f32 `a` is rounded while staging;
        an f32 residual with `stream=False` is added in f32 and the sum stored as 16 bit."""
        flags = (capi.PM_FLAG_A_F32 if a.dtype == torch.float32 else 0) | (capi.PM_FLAG_OUT_F32 if stream else 0)
        odt = torch.float32 if stream else self.dtype
        if out is None:
            out = self.empty(M, n_out, dtype=odt)
        assert out.dtype == odt and out.stride(1) == 1
        if residual is not None:
            assert residual.stride(1) == 1 and residual.dtype in (odt, torch.float32), (residual.dtype, odt)
            if residual.dtype == torch.float32 and not stream:
                flags |= capi.PM_FLAG_RES_F32
        return flags, out

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    # fused GroupNorm statistics: `stats=(NI, groups)` on a GEMM-family op returns (out, totals) where
    # totals [NI, groups, 2] are the {sum, sumsq} of the stored output per instance (NI consecutive row
    # blocks) and group; taken from the epilogue's column sums when every instance is a whole number of
    # 64-row blocks, else by the ordinary statistics pass over the output.
    def begin_forward(self):
        """Called by UNetModel at the top of a forward (on the forward's stream): zero the stream's totals arena, ONE memset."""
        if not self.stats_i64:
            return
        key = self._stream()
        ar = self._arena.get(key)
        if ar is None:
            ar = self._arena[key] = [torch.empty(self.arena_bytes // 8, dtype=torch.int64, device=self.device), 0]
        ar[0].zero_()
        ar[1] = 0

    def _alloc_totals(self, NI, groups=32):
        """zeroed int64 [NI, groups, 4, 8] (every value in a 64-byte sector of its own: csrc/gemm_common.hpp GS_STRIDE): the next
        slice of the stream's arena (each slice is handed out once per zeroing)"""
        n = NI * groups * 4 * 8
        ar = self._arena.get(self._stream())
        if ar is not None and ar[1] + n <= ar[0].numel():
            t = ar[0][ar[1]:ar[1] + n].view(NI, groups, 4, 8)
            ar[1] += n
            return t
        return torch.zeros(NI, groups, 4, 8, dtype=torch.int64, device=self.device)

    @staticmethod
    def totals_f32(tot):
        """int64 limbs [N, groups, 4, 8] -> f32 {sum, sumsq} [N, groups, 2] (frame-sharded exchanges, tests); f32 passes through"""
        if tot.dtype != torch.int64:
            return tot
        t = tot[..., 0].double()
        return torch.stack([t[..., 0] * 2.0 ** -12 + t[..., 1] * 2.0 ** -44, t[..., 2] * 2.0 ** -12 + t[..., 3] * 2.0 ** -44],
                           dim=-1).float()

    def _stats_begin(self, M, n_out, stats, K=0):
        """-> (statistics buffer, rows per block / flags) or None when the epilogue cannot deliver the statistics."""
        if stats is None:
            return None
        NI, groups = stats[0], stats[1]
        rows = self.lib.pm_gemm_colstats_rows(M, n_out, K, 0, self.ws_bytes)  # 64 unsplit, 16 from the split-K reduce
        if M % NI or (M // NI) % rows or M % rows or n_out % groups:
            return None
        # int64 totals only where the statistics come out of the split-K REDUCE pass (rows == 16: the deep levels).  In the MFMA
        # kernels' own epilogue the group reduction + atomics cost more than the finalize launch they save (A/B on one box, step
        # +1.2 % / +3 % with them, profiles/r06/stats_i64_ab.txt): unsplit calls keep column sums + pm_groupnorm_finalize_colstats
        if self.stats_i64 and groups == 32 and NI <= 255 and (rows == 16 or self.stats_i64_all):
            return self._alloc_totals(NI, groups), -(capi.PM_FLAG_STATS_I64 | (NI << 8))  # (negative: the flag word)
        return torch.empty((M // rows) * n_out * 2, dtype=torch.float32, device=self.device), rows

    @staticmethod
    def _stats_flags(col):
        return -col[1] if col is not None and col[1] < 0 else 0

    def _stats_end(self, out, col, stats):
        if stats is None:
            return out
        NI, groups = stats[0], stats[1]
        if col is None:
            # (NI, groups, "lazy"): the caller only wants statistics that come for free from the epilogue
            return (out, None) if len(stats) > 2 else (out, self.groupnorm_stats(out, NI, groups))
        if col[1] < 0:
            return out, col[0]  # int64 totals: the epilogue has added to them, nothing to finalize
        buf, rows = col
        M, n_out = out.shape
        tot = torch.empty(NI, groups, 2, dtype=torch.float32, device=self.device)
        rc = self.lib.pm_groupnorm_finalize_colstats(_ptr(buf), _ptr(tot), M // rows, n_out, NI, groups, self._stream())
        capi.check(rc, "pm_groupnorm_finalize_colstats")
        return out, tot

    # -- GEMM family -----------------------------------------------------------------------------
    def gemm(self, a, w, bias=None, residual=None, act="none", out=None, stream=False, stats=None, col_scale=None,
             split_a=False):
        """out[M, N] = epi(a[M, K] @ w[N, K]^T); GEGLU halves N (weights pre-interleaved).
        col_scale f32 [N] (instead of a bias): column n is multiplied by col_scale[n] in f32 before the store.
        split_a (f32 `a`, stream output): two passes, a = hi + lo in 16 bit each (PM_FLAG_A_LO)."""
        wrap = 0
        if a.shape[1] % 64:
            # pm_gemm walks whole 64-wide K-tiles (every Linear of the U-Net has K % 64 == 0); narrower models (the
            # reduced-width first-stage encoder's 1x1 shortcut) get both operands zero-padded along K, ONCE, before
            # any of the passes below: same product (the low pass of split_a used to see the unpadded K)
            pad = 64 - a.shape[1] % 64
            a = torch.nn.functional.pad(a, (0, pad))
            w = torch.nn.functional.pad(w, (0, pad)).contiguous()
        if a.dtype == torch.float32 and self.presplit:
            # the f32 stream as an operand of the DMA-staged 16-bit kernels: one pass of pm_split16 in front
            # (split_a: [hi | lo] against W walked twice - the two-pass PM_FLAG_A_LO product in one launch)
            both = split_a and stream and act == "none"
            a = self.split16(a, with_lo=both)
            wrap = capi.PM_FLAG_W_WRAP if both else 0
        elif split_a and a.dtype == torch.float32 and stream and act == "none":
            y = self.gemm(a, w, bias, residual, out=out, stream=True)
            return self._gemm_lo(a, w, y, stats)
        if not wrap and a.dtype == self.dtype and a.shape[1] == 2 * w.shape[1] and w.shape[1] % 64 == 0:
            wrap = capi.PM_FLAG_W_WRAP  # a [hi | lo] operand (PM_OUT_HILO norm output): W walked twice
        M, K = a.shape
        N = w.shape[0]
        assert w.shape[1] * (2 if wrap else 1) == K and w.is_contiguous() and w.dtype == self.dtype
        n_out = N // 2 if act == "geglu" else N
        flags, out = self._gemm_io(a, residual, out, M, n_out, stream)
        flags |= wrap
        if col_scale is not None:
            assert bias is None and col_scale.dtype == torch.float32 and col_scale.numel() == N
            bias, flags = col_scale, flags | capi.PM_FLAG_BIAS_IS_SCALE
        col = self._stats_begin(M, n_out, stats, K)
        flags |= self._stats_flags(col)
        rc = self.lib.pm_gemm(_ptr(a), self._rows(a, True), _ptr(w), w.shape[1], _ptr(bias),
                              _ptr(residual), residual.stride(0) if residual is not None else 0,
                              _ptr(out), out.stride(0), M, N, K, capi.ACT_CODES[act], flags, self.dt,
                              _ptr(self.workspace), self.ws_bytes, _ptr(col[0] if col else None), self._stream())
        capi.check(rc, f"pm_gemm M={M} N={N} K={K}")
        return self._stats_end(out, col, stats)

    def split16(self, x, with_lo=False):
        """f32 [M, K] -> 16-bit [M, K] (rounded) or [M, 2K] = [hi | lo] with lo = round(x - hi)  (pm_split16)."""
        M, K = x.shape
        assert x.dtype == torch.float32 and x.stride(1) == 1
        y = self.empty(M, 2 * K if with_lo else K)
        rc = self.lib.pm_split16(_ptr(x), x.stride(0), _ptr(y), y.stride(0), M, K, int(with_lo), self.dt, self._stream())
        capi.check(rc, f"pm_split16 M={M} K={K}")
        return y

    def split16_upsample2x(self, x, F, H, W, with_lo=False):
        """f32 [F*H*W, K] -> 16-bit [F*2H*2W, K or 2K]: pm_split16 through a nearest x2 upsample (pm_split16_upsample2x)."""
        M, K = x.shape
        assert x.dtype == torch.float32 and x.stride(1) == 1 and M == F * H * W
        y = self.empty(4 * M, 2 * K if with_lo else K)
        rc = self.lib.pm_split16_upsample2x(_ptr(x), x.stride(0), _ptr(y), y.stride(0), F, H, W, K, int(with_lo), self.dt,
                                            self._stream())
        capi.check(rc, f"pm_split16_upsample2x F={F} H={H} W={W} K={K}")
        return y

    def _gemm_lo(self, a, w, y, stats):
        """y += (a - round16(a)) @ w^T, in place (y f32); fused statistics, if wanted, come from this final pass."""
        M, K = a.shape
        N = w.shape[0]
        col = self._stats_begin(M, N, stats, K)
        flags = capi.PM_FLAG_A_F32 | capi.PM_FLAG_OUT_F32 | capi.PM_FLAG_A_LO | self._stats_flags(col)
        rc = self.lib.pm_gemm(_ptr(a), self._rows(a, True), _ptr(w), K, 0, _ptr(y), y.stride(0), _ptr(y), y.stride(0),
                              M, N, K, capi.PM_ACT_NONE, flags, self.dt, _ptr(self.workspace), self.ws_bytes,
                              _ptr(col[0] if col else None), self._stream())
        capi.check(rc, f"pm_gemm (low part) M={M} N={N} K={K}")
        return self._stats_end(y, col, stats)

    def conv3x3(self, x, wp, bias, F, H, W, stride=1, upsample=False, residual=None, out=None, stream=False,
                pad_lo=1, stats=None, presplit_upsample=True):
        """x [F*H*W, Cin] -> [F*Ho*Wo, Cout]; wp packed [Cout, 9*Cin]."""
        if x.dtype == torch.float32 and self.presplit and x.shape[1] % 8 == 0:
            if upsample and stride == 1 and pad_lo == 1 and self.upsample_presplit and presplit_upsample and x.shape[1] % 64 == 0:
                # Upsample.conv (openaimodel3d.py:96-108): the nearest x2 interpolation is written out by the conversion pass
                # (4x the 16-bit bytes, once) and the conv runs in the FAST 3x3 mode on the DMA-staged ring kernels instead of
                # gathering the upsampled pixels per lane in the general mode
                x = self.split16_upsample2x(x, F, H, W, with_lo=self._hilo())
                H, W, upsample = 2 * H, 2 * W, False
            else:
                x = self.split16(x, with_lo=self._hilo())  # the f32 stream (Downsample / Upsample): rounded once here, then the DMA loaders
            # (pm_split16 moves 8-element chunks: other widths keep the register-staged f32 loader)
        if x.shape[1] * 9 == 2 * wp.shape[1]:
            wp = self._repeat_taps(wp, 9)  # [hi | lo] input: the same weights for both halves of every tap
        cin = x.shape[1]
        cout = wp.shape[0]
        assert x.shape[0] == F * H * W and wp.shape[1] == 9 * cin and wp.is_contiguous()
        hv, wv = (2 * H, 2 * W) if upsample else (H, W)
        ho, wo = (hv + pad_lo - 2) // stride + 1, (wv + pad_lo - 2) // stride + 1
        flags, out = self._gemm_io(x, residual, out, F * ho * wo, cout, stream)
        col = self._stats_begin(F * ho * wo, cout, stats, 9 * cin)
        flags |= self._stats_flags(col)
        rc = self.lib.pm_conv2d_3x3(_ptr(x), self._rows(x, True), _ptr(wp), _ptr(bias), _ptr(residual),
                                    residual.stride(0) if residual is not None else 0, _ptr(out),
                                    out.stride(0), F, H, W, cin, cout, stride, int(upsample), int(pad_lo),
                                    _ptr(self.zero_page), flags, self.dt, _ptr(self.workspace),
                                    self.ws_bytes, _ptr(col[0] if col else None), self._stream())
        capi.check(rc, f"pm_conv2d_3x3 F={F} H={H} W={W} Cin={cin} Cout={cout}")
        return self._stats_end(out, col, stats)

    def conv_t3(self, x, wp, bias, F, P, residual=None, halo_lo=None, halo_hi=None, out=None, stream=False,
                stats=None, clip_frames=None):
        """3-tap conv over frames: x [F*P, Cin] -> [F*P, Cout]; wp packed [Cout, 3*Cin].
        clip_frames: F / clip_frames independent clips batched along the frames (zero padding at both ends of each)."""
        if x.dtype == torch.float32 and self.presplit and halo_lo is None and halo_hi is None and x.shape[1] % 8 == 0:
            x = self.split16(x)
        if x.shape[1] * 3 == 2 * wp.shape[1]:
            wp = self._repeat_taps(wp, 3)
        cin = x.shape[1]
        cout = wp.shape[0]
        assert x.shape[0] == F * P and wp.shape[1] == 3 * cin and wp.is_contiguous()
        for h in (halo_lo, halo_hi):
            assert h is None or (h.shape == (P, cin) and self._rows(h) == self._rows(x))
        flags, out = self._gemm_io(x, residual, out, F * P, cout, stream)
        col = self._stats_begin(F * P, cout, stats, 3 * cin)
        flags |= self._stats_flags(col)
        rc = self.lib.pm_conv_temporal_k3_clips(_ptr(x), self._rows(x, True), _ptr(halo_lo), _ptr(halo_hi),
                                                _ptr(wp), _ptr(bias), _ptr(residual),
                                                residual.stride(0) if residual is not None else 0,
                                                _ptr(out), out.stride(0), F, clip_frames or F, P, cin, cout,
                                                _ptr(self.zero_page), flags, self.dt, _ptr(self.workspace),
                                                self.ws_bytes, _ptr(col[0] if col else None), self._stream())
        capi.check(rc, f"pm_conv_temporal_k3 F={F} P={P} Cin={cin} Cout={cout}")
        return self._stats_end(out, col, stats)

    def _repeat_taps(self, wp, taps):
        """packed conv weights [Cout, taps*Cin] -> [Cout, taps*2*Cin] with every tap's channel block repeated (cached)."""
        key = (wp.data_ptr(), taps, tuple(wp.shape))
        hit = self._dup.get(key)
        if hit is None or hit[0] is not wp:
            cout = wp.shape[0]
            w3 = wp.view(cout, taps, -1)
            hit = self._dup[key] = (wp, torch.cat([w3, w3], dim=2).reshape(cout, -1).contiguous())
        return hit[1]

    def gemv(self, w, x, bias=None, silu_in=False, act="none"):
        """f32 y[N] = act(w[N, K] @ (silu?)(x[K]) + bias)."""
        N, K = w.shape
        assert x.dtype == torch.float32 and x.numel() == K and w.is_contiguous()
        y = torch.empty(N, dtype=torch.float32, device=self.device)
        rc = self.lib.pm_gemv_f32(_ptr(w), K, _ptr(x), _ptr(bias), _ptr(y), N, K, int(silu_in),
                                  capi.ACT_CODES[act], self.dt, self._stream())
        capi.check(rc, f"pm_gemv_f32 N={N} K={K}")
        return y

    def timestep_embedding(self, t, freqs):
        """f32 [n, 2 * half] = [cos(t f) | sin(t f)] for n timesteps (int64 or f32 on the device) and the f32 frequency table
        (utils_diffusion.py:8-28): one launch instead of torch's convert / mul / cos / sin / cat."""
        assert t.dim() == 1 and t.dtype in (torch.int64, torch.float32) and t.is_contiguous()
        assert freqs.dtype == torch.float32 and freqs.is_contiguous()
        n, half = t.numel(), freqs.numel()
        y = torch.empty(n, 2 * half, dtype=torch.float32, device=self.device)
        rc = self.lib.pm_timestep_embedding(_ptr(t), int(t.dtype == torch.int64), _ptr(freqs), _ptr(y), n, half, self._stream())
        capi.check(rc, f"pm_timestep_embedding n={n} half={half}")
        return y

    # -- normalisation ---------------------------------------------------------------------------
    def groupnorm_stats(self, x, NI, groups=32):
        """{sum, sumsq} per (instance, group): f32 [NI, groups, 2] (deterministic two-level sum)."""
        M, C = x.shape
        P = M // NI
        assert P * NI == M
        if self.stats_i64 and groups == 32:
            tot = self._alloc_totals(NI, groups)
            rc = self.lib.pm_groupnorm_stats(_ptr(x), self._rows(x, True), 0, _ptr(tot), NI, P, C, groups,
                                             self._in_dt(x) | capi.PM_TOTALS_I64, self._stream())
            capi.check(rc, f"pm_groupnorm_stats (int64 totals) NI={NI} P={P} C={C}")
            return tot
        nch = self.lib.pm_groupnorm_nchunks(P, C)
        part = torch.empty(NI, nch, groups, 2, dtype=torch.float32, device=self.device)
        tot = torch.empty(NI, groups, 2, dtype=torch.float32, device=self.device)
        rc = self.lib.pm_groupnorm_stats(_ptr(x), self._rows(x, True), _ptr(part), _ptr(tot), NI, P, C,
                                         groups, self._in_dt(x), self._stream())
        capi.check(rc, f"pm_groupnorm_stats NI={NI} P={P} C={C}")
        return tot

    def groupnorm_apply(self, x, totals, gamma, beta, eps, NI, silu, count=None, groups=32, out=None):
        M, C = x.shape
        P = M // NI
        if count is None:
            count = float(P * (C // groups))
        odt = self.dt
        if self._hilo() and out is None:
            out, odt = self.empty(M, 2 * C), self.dt | capi.PM_OUT_HILO  # [hi | lo]
        if out is None:
            out = self.empty(M, C)
        if totals.dtype == torch.int64:  # limbs [NI * nsum, groups, 4]: the kernel sums nsum consecutive entries per instance
            nsum = totals.shape[0] // NI
            assert totals.shape == (NI * nsum, groups, 4, 8) and totals.is_contiguous() and 1 <= nsum <= 255
            odt |= capi.PM_TOTALS_I64 | (nsum << 16)
        else:  # f32 {sum, sumsq}: [NI, groups, 2], or [NI * n, groups, 2] = n entries per instance that the kernel sums (in entry order)
            nsum = totals.shape[0] // NI
            assert totals.shape == (NI * nsum, groups, 2) and totals.is_contiguous() and 1 <= nsum <= 255
            if nsum > 1:
                odt |= nsum << 16
        rc = self.lib.pm_groupnorm_apply(_ptr(x), self._rows(x, True), _ptr(totals), _ptr(gamma),
                                         _ptr(beta), _ptr(out), self._rows(out), NI, P, C, groups,
                                         float(count), float(eps), int(silu), self._in_dt(x), odt,
                                         self._stream())
        capi.check(rc, f"pm_groupnorm_apply NI={NI} P={P} C={C}")
        return out

    def groupnorm(self, x, gamma, beta, eps, NI, silu, groups=32, stats_reduce=None, out=None, totals=None):
        """GroupNorm over NI instances of [P, C]; stats_reduce(totals [NI, groups, 2], local_count)
        (frame-sharded mode) returns the all-rank totals and the total element count per group.
        `totals`: statistics already produced by the op that wrote x (fused epilogue)."""
        tot = totals if totals is not None else self.groupnorm_stats(x, NI, groups)
        count = None
        if stats_reduce is not None:
            tot, count = stats_reduce(self.totals_f32(tot), (x.shape[0] // NI) * (x.shape[1] // groups))
            tot = tot.contiguous()
        return self.groupnorm_apply(x, tot, gamma, beta, eps, NI, silu, count, groups, out)

    def layernorm(self, x, gamma, beta, eps=1e-5, out=None):
        M, C = x.shape
        odt = self.dt
        if self._hilo() and out is None:
            out, odt = self.empty(M, 2 * C), self.dt | capi.PM_OUT_HILO  # [hi | lo]
        if out is None:
            out = self.empty(M, C)
        rc = self.lib.pm_layernorm(_ptr(x), self._rows(x, True), _ptr(gamma), _ptr(beta), _ptr(out),
                                   self._rows(out), M, C, float(eps), self._in_dt(x), odt,
                                   self._stream())
        capi.check(rc, f"pm_layernorm M={M} C={C}")
        return out

    # -- attention -------------------------------------------------------------------------------
    def ln_gemm(self, x, gamma, beta, w, bias=None, act="none", col_scale=None, eps=1e-5):
        """gemm(layernorm(x), w, ...) for the f32 stream x: ONE kernel (pm_ln_gemm) where the shape is served,
        the pm_layernorm + pm_gemm pair otherwise - same results either way."""
        M, K = x.shape
        N = w.shape[0]
        code = capi.ACT_CODES[act]
        # r06: where the projection itself would run on gemm_wide_stream (GEGLU / plain 16-bit flavours on whole 256 x 256 tiles: 25-
        # 33 % ahead of the other GEMM kernels at K = 320), the pm_layernorm + pm_gemm pair beats the panel kernel
        # (profiles/r06/wide_stream_probe.txt); PANDORA_LN_PAIR_STREAM=0 keeps the panel kernel (A/B)
        stream_pair = (self.ln_pair_stream and col_scale is None and code in (capi.PM_ACT_NONE, capi.PM_ACT_GEGLU)
                       and self.lib.pm_gemm_kernel_choice(M, N, K, code, 0, self.ws_bytes) == 5)
        if (self.fused_ln and not stream_pair and not self._hilo() and x.dtype == torch.float32
                and code in (capi.PM_ACT_NONE, capi.PM_ACT_GEGLU) and self.lib.pm_ln_gemm_supported(M, N, K, code)):
            assert w.shape[1] == K and w.is_contiguous() and w.dtype == self.dtype and x.stride(1) == 1
            flags = 0
            if col_scale is not None:
                assert bias is None and col_scale.dtype == torch.float32 and col_scale.numel() == N
                bias, flags = col_scale, capi.PM_FLAG_BIAS_IS_SCALE
            out = self.empty(M, N // 2 if act == "geglu" else N)
            rc = self.lib.pm_ln_gemm(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), eps, _ptr(w), K, _ptr(bias),
                                     _ptr(out), out.stride(0), M, N, K, code, flags, self.dt, self._stream())
            capi.check(rc, f"pm_ln_gemm M={M} N={N} K={K}")
            return out
        return self.gemm(self.layernorm(x, gamma, beta, eps), w, bias, act=act, col_scale=col_scale)

    def attention(self, q, k1, v1, heads, k2=None, v2=None, w2=1.0, out=None, prescaled=False):
        """q [B, Nq, heads*64] view; kX/vX [B or 1, NkX, heads*64] views (last dim contiguous).
        out = attn(q,k1,v1) + w2*attn(q,k2,v2), each softmax-normalised on its own.
        prescaled: q already carries `q_prescale` (then softmax(q k^T 64^-1/2) == softmax_2(q' k^T))."""
        B, Nq, C = q.shape
        assert C == heads * 64 and q.stride(2) == 1 and q.dtype == self.dtype

        def kv(k, v):
            assert k.shape == v.shape and k.stride() == v.stride() and k.stride(2) == 1
            assert k.shape[0] in (1, B) and k.shape[2] == C
            return (0 if k.shape[0] == 1 and B > 1 else k.stride(0)), k.stride(1), k.shape[1]

        bs1, rs1, n1 = kv(k1, v1)
        bs2 = rs2 = n2 = 0
        if k2 is not None:
            bs2, rs2, n2 = kv(k2, v2)
        if out is None:
            out = self.empty(B, Nq, C)
        scale = math.log(2.0) if prescaled else 64 ** -0.5  # ln 2 * log2 e == 1: no scaling left in the kernel
        rc = self.lib.pm_attention(_ptr(q), q.stride(0), q.stride(1), _ptr(k1), _ptr(v1), bs1, rs1,
                                   n1, _ptr(k2), _ptr(v2), bs2, rs2, n2, float(w2), _ptr(out),
                                   out.stride(0), out.stride(1), B, heads, Nq, scale, self.dt,
                                   self._stream())
        capi.check(rc, f"pm_attention B={B} heads={heads} Nq={Nq} Nk={n1}+{n2}")
        return out

    def attention_fp8(self, q, k, v, heads, out=None, prescaled=False):
        """Single-segment attention with fp8 (e4m3) operands on the block-scaled MFMA (pm_attention_fp8): q [B, Nq, C],
        k / v [B, Nk, C] views of identical strides.  ~2-3e-2 relative error: BASELINE configs[4], opt-in."""
        B, Nq, C = q.shape
        Nk = k.shape[1]
        assert C == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and k.shape == v.shape and k.stride() == v.stride()
        assert k.shape[0] == B and q.dtype == self.dtype
        if out is None:
            out = self.empty(B, Nq, C)
        need = self.lib.pm_attention_fp8_workspace_bytes(B, heads, Nq, Nk)
        key = ("fp8", self._stream())
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < need:
            ws = self._workspaces[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        scale = math.log(2.0) if prescaled else 64 ** -0.5
        rc = self.lib.pm_attention_fp8(_ptr(q), q.stride(0), q.stride(1), _ptr(k), _ptr(v), k.stride(0), k.stride(1), Nk,
                                       _ptr(out), out.stride(0), out.stride(1), B, heads, Nq, scale, self.dt,
                                       _ptr(ws), ws.numel(), self._stream())
        capi.check(rc, f"pm_attention_fp8 B={B} heads={heads} Nq={Nq} Nk={Nk}")
        return out

    def attention_generic(self, q, k, v, heads, out=None):
        """Any head dim <= 128 (pm_attention_generic): q [B, Nq, heads*D], k / v [B, Nk, heads*D] views, Nk <= 1024."""
        B, Nq, C = q.shape
        D = C // heads
        assert D * heads == C and q.stride(2) == 1 and k.stride(2) == 1 and k.shape == v.shape and k.stride() == v.stride()
        if out is None:
            out = self.empty(B, Nq, C)
        rc = self.lib.pm_attention_generic(_ptr(q), q.stride(0), q.stride(1), _ptr(k), _ptr(v), k.stride(0), k.stride(1),
                                           k.shape[1], _ptr(out), out.stride(0), out.stride(1), B, heads, Nq, D,
                                           float(D) ** -0.5, self.dt, self._stream())
        capi.check(rc, f"pm_attention_generic B={B} heads={heads} Nq={Nq} D={D}")
        return out

    def attention_temporal(self, q, k, v, heads, out=None):
        """q [Fq, P, heads*64], k/v [Fk, P, heads*64] views: attention over frames at each pixel."""
        Fq, P, C = q.shape
        Fk = k.shape[0]
        assert C == heads * 64 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride() == k.stride()
        assert q.stride(0) == P * q.stride(1) and k.stride(0) == P * k.stride(1)
        if out is None:
            out = self.empty(Fq, P, C)
        assert out.stride(0) == P * out.stride(1)
        rc = self.lib.pm_attention_temporal(_ptr(q), q.stride(1), _ptr(k), _ptr(v), k.stride(1),
                                            _ptr(out), out.stride(1), Fq, Fk, P, heads, 64 ** -0.5,
                                            self.dt, self._stream())
        capi.check(rc, f"pm_attention_temporal Fq={Fq} Fk={Fk} P={P} heads={heads}")
        return out

    # -- path boundary ---------------------------------------------------------------------------
    def ddim_update(self, x, e_c, e_u, noise, cfg, sqrt_ac, sqrt_1mac, rescale, sqrt_a_prev,
                    dir_coef, sigma, want_x0=True):
        assert x.dtype == torch.float32 and x.is_contiguous() and e_c.is_contiguous()
        x_prev = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        rc = self.lib.pm_ddim_update(_ptr(x), _ptr(e_c), _ptr(e_u), _ptr(noise), _ptr(x_prev),
                                     _ptr(x0), x.numel(), float(cfg), float(sqrt_ac),
                                     float(sqrt_1mac), float(rescale), float(sqrt_a_prev),
                                     float(dir_coef), float(sigma), self._in_dt(e_c), self._stream())
        capi.check(rc, "pm_ddim_update")
        return x_prev, x0

    def pack_input(self, x, cond):
        """x f32 [C1, F, P], cond f32 [C2, F, P] -> [F*P, C1+C2] activations."""
        C1, F, P = x.shape
        C2 = 0 if cond is None else cond.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous()
        y = self.empty(F * P, C1 + C2)
        rc = self.lib.pm_pack_input(_ptr(x), _ptr(cond), _ptr(y), C1, C2, F, P, self.dt, self._stream())
        capi.check(rc, "pm_pack_input")
        return y

    def latent_affine(self, x, W, b, inv_scale, cpad=8):
        """x f32 [C, F, P] -> [F*P, cpad]: W . (x * inv_scale) + b, channels-last, zero-padded channels."""
        C, F, P = x.shape
        assert x.dtype == torch.float32 and x.is_contiguous() and W.shape == (C, C)
        y = self.empty(F * P, cpad)
        rc = self.lib.pm_latent_affine(_ptr(x), _ptr(W), _ptr(b), _ptr(y), C, cpad, F, P, float(inv_scale),
                                       self.dt, self._stream())
        capi.check(rc, "pm_latent_affine")
        return y

    def softmax_rows(self, x, scale):
        """f32 scores [M, N] -> softmax(scale * x) per row, in the activation dtype."""
        M, N = x.shape
        assert x.dtype == torch.float32 and x.stride(1) == 1
        y = self.empty(M, N)
        rc = self.lib.pm_softmax_rows(_ptr(x), x.stride(0), _ptr(y), y.stride(0), M, N, float(scale), self.dt,
                                      self._stream())
        capi.check(rc, f"pm_softmax_rows M={M} N={N}")
        return y

    def unpack_output(self, y, F, P):
        """[F*P, C] -> [C, F, P] (same dtype as y: 16-bit or f32)."""
        C = y.shape[1]
        assert y.is_contiguous()
        out = self.empty(C, F, P, dtype=y.dtype)
        rc = self.lib.pm_unpack_output(_ptr(y), _ptr(out), C, F, P, self._in_dt(y), self._stream())
        capi.check(rc, "pm_unpack_output")
        return out


def memory_efficient_attention(q, k, v, attn_bias=None, op=None, scale=None, ops=None):
    """The reference's kernel seam (attention.py:64-67 rebinds CrossAttention.forward to efficient_forward, which
    calls `xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None, op=None)` on contiguous
    `(b * heads, n, 64)` tensors, attention.py:166-175) served by pm_attention: same argument list, same layout,
    same result `(b * heads, n_q, 64)`.  Each (batch, head) pair is one batch element of a 1-head call."""
    if attn_bias is not None:
        raise NotImplementedError("attn_bias: the reference passes None (attention.py:175)")
    if q.dim() != 3 or q.shape[-1] != 64 or k.shape != v.shape or k.shape[0] != q.shape[0] or k.shape[-1] != 64:
        raise ValueError(f"expected (b*heads, n, 64) tensors, got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if scale is not None and abs(scale - 64 ** -0.5) > 1e-9:
        raise NotImplementedError("pm_attention applies the softmax scale 64^-1/2 of attention.py:24")
    ops = ops or HipOps(q.dtype, q.device)
    return ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), heads=1)

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


class PackedWeights:
    """Mixin of the modules that keep kernel-side packed copies of their parameters (`self._packed`): the copies
    must die whenever the parameters change - through `load_state_dict` on the module OR on any parent (the
    reference's loaders call it on the whole LatentVisualDiffusion, scripts/evaluation/inference.py:27-52), through
    `.to()` / `.half()` / `.cuda()`, or through in-place edits (an optimizer step).  `_pack_epoch` counts the
    invalidations so that holders of derived state (the sampler's captured HIP graphs, which keep raw pointers to
    the packed tensors) can notice."""

    def _init_packed(self):
        self._packed, self._packed_fp, self._pack_epoch = None, None, 0
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._packed, self._packed_fp, self._fp_sample = None, None, None
        self._pack_epoch += 1

    def _apply(self, fn, *args, **kwargs):  # .to() / .half() / .float() / .cuda() of this module or a parent
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def _fingerprint(self):
        """(version counter, address) of a fixed sample of the parameters: an in-place edit bumps the version, a
        re-assignment changes the address.  The sample (~40 tensors) is listed once per pack: walking all ~1500
        parameters per forward cost about a millisecond of host time; a re-assigned parameter object
        is caught by the address of the slot it is read back from."""
        sample = getattr(self, "_fp_sample", None)
        if sample is None:
            named = list(self.named_parameters())
            step = max(1, len(named) // 40)
            mods = dict(self.named_modules())
            # + every 0-dim parameter (CrossAttention.alpha): prepare() folds those into HOST scalars, nothing else would notice
            picked = named[::step] + [(n, p) for i, (n, p) in enumerate(named) if p.dim() == 0 and i % step]
            sample = self._fp_sample = (len(named), [(mods[n.rpartition(".")[0]], n.rpartition(".")[2]) for n, _ in picked])
        out = []
        for mod, leaf in sample[1]:
            p = getattr(mod, leaf)
            out.append((p._version, p.data_ptr()))
        return (sample[0],) + tuple(out)

    def packed(self):
        """the current packed weight set, rebuilt (prepare()) if the parameters changed since it was made.  Holders of
        derived state (a captured HIP graph) call this before every reuse: a replay does not walk forward(), so an
        in-place weight edit would otherwise go unnoticed by the graph."""
        if self._packed is None or self._packed_fp != self._fingerprint():
            if self._packed is not None:
                self.invalidate_packed()
            self._fp_sample = None
            self.prepare()
            self._packed_fp = self._fingerprint()
        return self._packed

"""GPU parity of every C-ABI kernel against the per-op oracle (oracle/ops_torch.py, f32 CPU math in
the reference's own NCHW / einsum formulation).  Tolerances: norm-relative error <= 1e-3 for f16 I/O
(the north-star bound), <= 4e-3 for bf16 I/O (bf16 output rounding alone is 2^-9 = 2e-3)."""
import pytest
import torch

from oracle.ops_torch import TorchOps
from open_pandora_amd import packing

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1e-3, torch.bfloat16: 4e-3}
DTYPES = [torch.float16, torch.bfloat16]
REF = TorchOps()


def _f32(t):
    # (GroupNorm totals of the HIP op table are int64 fixed-point limbs [N, groups, 4] since r06: HipOps.totals_f32)
    from open_pandora_amd.ops_hip import HipOps
    return HipOps.totals_f32(t) if t.dtype == torch.int64 else t


def rel_err(a, b):
    a, b = _f32(a).float().cpu(), _f32(b).float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def rnd(*shape, dtype, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def dev(*ts):
    return [None if t is None else t.cuda() for t in ts]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (128, 128, 64), (1, 1280, 320), (77, 640, 1024),
                                   (1000, 4, 128), (513, 2560, 640)])
def test_gemm_bias_residual(hip_ops_factory, dtype, M, N, K):
    ops = hip_ops_factory(dtype)
    a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    bias = rnd(N, dtype=torch.float32, seed=3)
    res = rnd(M, N, dtype=dtype, seed=4)
    for use_b, use_r, act in [(False, False, "none"), (True, True, "none"), (True, False, "silu")]:
        want = REF.gemm(a, w, bias if use_b else None, res if use_r else None, act)
        da, dw, db, dr = dev(a, w, bias if use_b else None, res if use_r else None)
        got = ops.gemm(da, dw, db, dr, act)
        assert rel_err(got, want) <= TOL[dtype], (use_b, use_r, act)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 960, 320), (640, 3840, 1280)])  # (unsplit; split-K: the scale is in the reduce)
def test_gemm_column_scale(hip_ops_factory, dtype, M, N, K):
    """PM_FLAG_BIAS_IS_SCALE: out[:, n] = (a @ w^T)[:, n] * s[n], scaled in f32 before the one rounding of the
    store (the softmax scale on the q third of a fused q|k|v projection)."""
    ops = hip_ops_factory(dtype)
    a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    s = torch.cat([torch.full((N // 3,), 0.18033688), torch.ones(N - N // 3)])
    want = (a.float() @ w.float().t()) * s
    got = ops.gemm(a.cuda(), w.cuda(), col_scale=s.cuda())
    assert rel_err(got, want) <= TOL[dtype]
    # exactness: on the scaled third the result is the correctly rounded product, not a re-rounded 16-bit value
    exact = want.to(dtype).float()
    assert (got.float().cpu()[:, :N // 3] != exact[:, :N // 3]).float().mean().item() < 0.02


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 320, 640), (640, 1280, 2560)])  # (the second splits over K)
def test_gemm_split_f32_operand(hip_ops_factory, dtype, M, N, K):
    """split_a: an f32 A operand carried as hi + lo 16-bit parts over two passes (PM_FLAG_A_LO) - the result
    is the f32 product to ~1e-5 instead of the 16-bit operand's 3e-4 / 2e-3; fused statistics of the final sum."""
    ops = hip_ops_factory(dtype)
    a = rnd(M, K, dtype=torch.float32, scale=2.0, seed=1)
    w, bias = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2), rnd(N, dtype=torch.float32, seed=3)
    want = a @ w.float().t() + bias
    plain = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True)
    got, tot = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True, split_a=True, stats=(1, 32))
    e_plain, e_split = rel_err(plain, want), rel_err(got, want)
    assert got.dtype == torch.float32 and e_split <= (2e-5 if dtype == torch.float16 else 1e-4) and e_split < 0.1 * e_plain
    ref_tot = TorchOps._with_stats(want, (1, 32))[1]
    assert rel_err(tot, ref_tot) <= 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("presplit", [True, False])
def test_f32_operand_shape_edges(hip_ops_factory, monkeypatch, dtype, presplit):
    """K % 64 != 0 with an f32 A operand (split_a's low pass used to see the unpadded K -> PM_E_SHAPE): both
    operands are zero-padded once, in front of every pass, with and without the pm_split16 pre-pass."""
    ops = hip_ops_factory(dtype)
    monkeypatch.setattr(ops, "presplit", presplit)
    M, N, K = 300, 64, 96
    a = rnd(M, K, dtype=torch.float32, scale=2.0, seed=1)
    w, bias = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2), rnd(N, dtype=torch.float32, seed=3)
    want = a @ w.float().t() + bias
    got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True, split_a=True)
    assert rel_err(got, want) <= (2e-5 if dtype == torch.float16 else 1e-4)
    assert rel_err(ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True), want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 320, 640), (2560, 1280, 2560), (4099, 320, 960)])
def test_split16_and_wrapped_weights(hip_ops_factory, monkeypatch, dtype, M, N, K):
    """pm_split16 is bit-exact ([hi | lo] = the roundings PM_FLAG_A_F32 / PM_FLAG_A_LO apply while staging), and
    the one-pass [hi | lo] x W-walked-twice product (PM_FLAG_W_WRAP) equals the two-pass register-staged one."""
    ops = hip_ops_factory(dtype)
    a = rnd(M, K, dtype=torch.float32, scale=2.0, seed=1)
    hl = ops.split16(a.cuda(), with_lo=True).cpu()
    hi = a.to(dtype)
    assert torch.equal(hl[:, :K], hi) and torch.equal(hl[:, K:], (a - hi.float()).to(dtype))
    assert torch.equal(ops.split16(a.cuda()).cpu(), hi)
    w, bias = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2), rnd(N, dtype=torch.float32, seed=3)
    res = rnd(M, N, dtype=torch.float32, seed=4)
    want = a @ w.float().t() + bias + res
    assert ops.presplit
    one, tot = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), residual=res.cuda(), stream=True, split_a=True, stats=(1, 32))
    monkeypatch.setattr(ops, "presplit", False)
    two = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), residual=res.cuda(), stream=True, split_a=True)
    tol = 2e-5 if dtype == torch.float16 else 1e-4
    assert rel_err(one, want) <= tol and rel_err(two, want) <= tol and rel_err(one, two) <= 1e-6
    assert rel_err(tot, TorchOps._with_stats(want, (1, 32))[1]) <= 1e-4
    # plain f32 operand (rounded once): the pre-rounded DMA path equals the register-staged one bit for bit
    monkeypatch.setattr(ops, "presplit", True)
    p1 = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True)
    monkeypatch.setattr(ops, "presplit", False)
    p0 = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), stream=True)
    assert rel_err(p1, p0) <= 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integers(hip_ops_factory, dtype):
    """Small-integer operands make every product and sum exact: any fragment-layout mistake
    (row/col swap, k permutation) shows up as a mismatch, not as rounding."""
    ops = hip_ops_factory(dtype)
    g = torch.Generator().manual_seed(7)
    M, N, K = 261, 200, 192
    a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
    w = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
    w[:, ::2] *= 0  # asymmetric sparsity pattern
    want = a.float() @ w.float().t()
    got = ops.gemm(a.cuda(), w.cuda())
    assert torch.equal(got.float().cpu(), want.to(dtype).float())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_strided_views(hip_ops_factory, dtype):
    """A, residual and the output may be column slices of wider row-major buffers (fused qkv,
    zero-copy skip concat)."""
    ops = hip_ops_factory(dtype)
    M, N, K = 200, 320, 320
    abuf = rnd(M, 3 * K, dtype=dtype, seed=1)
    w = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    rbuf = rnd(M, 2 * N, dtype=dtype, seed=3)
    obuf = torch.zeros(M, N + 64, dtype=dtype).cuda()
    a, r = abuf[:, K:2 * K], rbuf[:, N:]
    want = REF.gemm(a, w, None, r)
    got = ops.gemm(abuf.cuda()[:, K:2 * K], w.cuda(), None, rbuf.cuda()[:, N:], out=obuf[:, 64:])
    assert rel_err(got, want) <= TOL[dtype]
    assert obuf[:, :64].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(260, 320), (64, 1280), (129, 512)])
def test_gemm_geglu(hip_ops_factory, dtype, M, C):
    ops = hip_ops_factory(dtype)
    w = rnd(8 * C, C, dtype=dtype, scale=C ** -0.5, seed=1)
    b = rnd(8 * C, dtype=torch.float32, seed=2)
    a = rnd(M, C, dtype=dtype, seed=3)
    # reference formulation: proj -> chunk -> x * gelu(gate)   (attention.py:415-422)
    y = a.float() @ w.float().t() + b
    xv, gate = y.chunk(2, dim=-1)
    want = xv * torch.nn.functional.gelu(gate)
    wp, bp = packing.pack_geglu(w, b)
    assert rel_err(REF.gemm(a, wp, bp, act="geglu"), want) < 1e-6  # the oracle's own packing
    got = ops.gemm(a.cuda(), wp.cuda(), bp.cuda(), act="geglu")
    assert got.shape == (M, 4 * C)
    assert rel_err(got, want) <= TOL[dtype]


def test_erf_approximant(hip_ops_factory):
    """The epilogues' erf / GELU (csrc/common.hpp, r06: Q(t) = 2^P(t), one v_exp_f32 and 8 full-rate ops) element by element
    against f64 erf (max abs <= 1e-5): a dense grid, the tails, huge and tiny arguments, NaN propagation."""
    from open_pandora_amd import capi
    ops = hip_ops_factory(torch.bfloat16, diag=True)
    x = torch.cat([torch.linspace(-12, 12, 2_000_001, dtype=torch.float64).float(),
                   torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e-8, -1e-8, 30.0, -30.0, 1e4, -1e4, 3e38, -3e38])]).cuda()
    y = torch.empty_like(x)
    for mode, name in ((0, "erf"), (1, "gelu")):
        capi.check(ops.lib.pm_debug_erf(x.data_ptr(), y.data_ptr(), x.numel(), mode, ops._stream()), "pm_debug_erf")
        torch.cuda.synchronize()
        xd = x.double().cpu()
        want = torch.erf(xd) if mode == 0 else 0.5 * xd * (1.0 + torch.erf(xd * 2 ** -0.5))
        err = (y.double().cpu() - want).abs() / want.abs().clamp(min=1.0)  # absolute below 1, relative above (f32 rounding)
        print(f"[parity] {name} approximant: max err {err.max():.3e} (bound 1e-5)")
        assert torch.isfinite(y).all() and err.max() <= 1e-5, (name, float(err.max()))
    nan = torch.tensor([float("nan")], device="cuda")
    out = torch.empty_like(nan)
    capi.check(ops.lib.pm_debug_erf(nan.data_ptr(), out.data_ptr(), 1, 1, ops._stream()), "pm_debug_erf")
    assert torch.isnan(out).all()  # a NaN stays a NaN through the GELU epilogue


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("F,H,W,Cin,Cout,stride,ups", [
    (2, 9, 7, 8, 64, 1, False),      # stem-like: K = 72 (K tail inside a tile)
    (3, 10, 16, 64, 72, 1, False),
    (2, 10, 16, 128, 128, 2, False),  # Downsample
    (2, 9, 7, 64, 64, 2, False),      # odd sizes, stride 2
    (2, 5, 8, 64, 128, 1, True),      # Upsample (nearest x2 fused into the gather)
    (1, 12, 12, 320, 4, 1, False),    # out conv: Cout = 4
])
def test_conv3x3(hip_ops_factory, dtype, F, H, W, Cin, Cout, stride, ups):
    ops = hip_ops_factory(dtype)
    x = rnd(F * H * W, Cin, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=2)
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    # independent reference: plain NCHW conv on the ORIGINAL weight layout
    xi = x.float().reshape(F, H, W, Cin).permute(0, 3, 1, 2)
    if ups:
        xi = torch.nn.functional.interpolate(xi, scale_factor=2, mode="nearest")
    yo = torch.nn.functional.conv2d(xi, w.float(), bias, stride=stride, padding=1)
    want = yo.permute(0, 2, 3, 1).reshape(-1, Cout)
    res = rnd(want.shape[0], Cout, dtype=dtype, seed=4)
    wp = packing.pack_conv3x3(w)
    assert rel_err(REF.conv3x3(x, wp, bias, F, H, W, stride, ups), want) < 1e-6
    got = ops.conv3x3(x.cuda(), wp.cuda(), bias.cuda(), F, H, W, stride, ups, residual=res.cuda())
    assert rel_err(got, want + res.float()) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("F,H,W,Cin,Cout", [(2, 5, 8, 64, 128), (3, 9, 7, 128, 64)])
def test_conv3x3_upsample_of_the_f32_stream(hip_ops_factory, monkeypatch, dtype, F, H, W, Cin, Cout):
    """Upsample.conv on the f32 stream (openaimodel3d.py:96-108): pm_split16_upsample2x writes the nearest x2 interpolation
    in 16 bit and the conv runs in the fast 3x3 mode; equal to the gathered general mode (PANDORA_UPSAMPLE_PRESPLIT=0) up to
    the summation order, both within the operand-rounding tolerance of the reference formulation."""
    ops = hip_ops_factory(dtype)
    x = rnd(F * H * W, Cin, dtype=torch.float32, scale=2.0, seed=1)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=2)
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    xi = torch.nn.functional.interpolate(x.to(dtype).float().reshape(F, H, W, Cin).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    want = torch.nn.functional.conv2d(xi, w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    wp = packing.pack_conv3x3(w)
    up = ops.split16_upsample2x(x.cuda(), F, H, W)
    assert torch.equal(up.float().cpu().reshape(F, 2 * H, 2 * W, Cin).permute(0, 3, 1, 2), xi)  # bit-exact rows
    both = ops.split16_upsample2x(x.cuda(), F, H, W, with_lo=True).float().cpu()  # [hi | lo]: hi + lo carries x at ~2x the mantissa
    x_up = torch.nn.functional.interpolate(x.reshape(F, H, W, Cin).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    x_up = x_up.permute(0, 2, 3, 1).reshape(-1, Cin)
    assert torch.equal(both[:, :Cin], up.float().cpu())
    assert rel_err(both[:, :Cin] + both[:, Cin:], x_up) < (2e-6 if dtype == torch.float16 else 2e-5)
    got = ops.conv3x3(x.cuda(), wp.cuda(), bias.cuda(), F, H, W, 1, True, stream=True)
    monkeypatch.setattr(ops, "upsample_presplit", False)
    gathered = ops.conv3x3(x.cuda(), wp.cuda(), bias.cuda(), F, H, W, 1, True, stream=True)
    assert got.shape == (F * 4 * H * W, Cout) and got.dtype == torch.float32
    assert rel_err(got, want) <= 2e-5 and rel_err(gathered, want) <= 2e-5 and rel_err(got, gathered) <= 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("F,P,C,halo", [(16, 40, 64, False), (4, 77, 320, False), (2, 50, 128, True)])
def test_conv_temporal(hip_ops_factory, dtype, F, P, C, halo):
    ops = hip_ops_factory(dtype)
    x = rnd(F * P, C, dtype=dtype, seed=1)
    w = rnd(C, C, 3, 1, 1, dtype=dtype, scale=(3 * C) ** -0.5, seed=2)
    bias = rnd(C, dtype=torch.float32, seed=3)
    lo = rnd(P, C, dtype=dtype, seed=5) if halo else None
    hi = rnd(P, C, dtype=dtype, seed=6) if halo else None
    # independent reference: Conv3d on (b c t h w) with the ORIGINAL weight layout
    xe = x.float().reshape(F, P, C)
    if halo:
        xe = torch.cat([lo.float()[None], xe, hi.float()[None]], 0)
    xc = xe.permute(2, 0, 1)[None, :, :, :, None]
    yo = torch.nn.functional.conv3d(xc, w.float(), bias, padding=(0 if halo else 1, 0, 0))
    want = yo[0, :, :, :, 0].permute(1, 2, 0).reshape(F * P, C)
    res = rnd(F * P, C, dtype=dtype, seed=4)
    wp = packing.pack_conv_t3(w)
    assert rel_err(REF.conv_t3(x, wp, bias, F, P, halo_lo=lo, halo_hi=hi), want) < 1e-6
    dl, dh = dev(lo, hi)
    got = ops.conv_t3(x.cuda(), wp.cuda(), bias.cuda(), F, P, residual=res.cuda(), halo_lo=dl, halo_hi=dh)
    assert rel_err(got, want + res.float()) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_wide_256x256_asm_loop(hip_ops_factory, dtype):
    """gemm_wide_kernel (csrc/gemm_wide.hip, r06: 256x256 tile, four waves of 128x128, generated assembly main loop with the
    accumulators fixed in a[0:255]), FORCED on every legal call through the diagnostics build (pm_debug_gemm_wide(2)) and checked
    against the other kernels of pm_gemm on the same inputs: exact-integer products (ragged last row / column tiles, K loops of
    2, 3 and 21 steps = both exits of the unrolled loop), every epilogue flavour incl. the lean 16-bit one (bias, per-column
    scale, GEGLU), f32 stream + f32 residual + fused GroupNorm statistics; and the library's own rule picks it for a long-K
    shape (choice of the shipped library: pm_gemm at K = 4096 on 4096 x 4096)."""
    ops = hip_ops_factory(dtype, diag=True)
    g = torch.Generator().manual_seed(5)
    try:
        for M, N, K in [(777, 1000, 128), (512, 512, 192), (1300, 520, 1344)]:
            a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
            w = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
            w[:, 1::3] *= 0
            want = (a.float() @ w.float().t()).to(dtype).float()
            ops.lib.pm_debug_gemm_wide(2)
            assert torch.equal(ops.gemm(a.cuda(), w.cuda()).float().cpu(), want), (M, N, K)
        M, N, K = 1024, 768, 1280
        a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
        bias, res = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=dtype, seed=4)
        scale = rnd(N, dtype=torch.float32, seed=8).abs() + 0.5
        res32 = rnd(M, N, dtype=torch.float32, seed=5)
        da, dw, db, dr = dev(a, w, bias, res)
        wgp, bgp = packing.pack_geglu(w, bias)
        calls = {"bias": lambda: ops.gemm(da, dw, db), "bias + res16": lambda: ops.gemm(da, dw, db, dr),
                 "column scale": lambda: ops.gemm(da, dw, col_scale=scale.cuda()),
                 "geglu": lambda: ops.gemm(da, wgp.cuda(), bgp.cuda(), act="geglu"),
                 "silu": lambda: ops.gemm(da, dw, db, act="silu"),
                 "stream + res32 + stats": lambda: ops.gemm(da, dw, db, res32.cuda(), stream=True, stats=(4, 32))}
        for name, fn in calls.items():
            outs = []
            for mode in (0, 2):
                ops.lib.pm_debug_gemm_wide(mode)
                y = fn()
                outs.append(y if isinstance(y, tuple) else (y, None))
            (y0, t0), (y2, t2) = outs
            assert rel_err(y2.float(), y0.float().cpu()) <= 2e-6 + (TOL[dtype] if y0.dtype != torch.float32 else 0), name
            if t0 is not None:
                assert rel_err(t2, t0.cpu()) <= 1e-5, name
        assert rel_err(ops.gemm(da, dw, db), REF.gemm(a, w, bias, None, "none")) <= TOL[dtype]
    finally:
        ops.lib.pm_debug_gemm_wide(1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_wide_stream_one_statement_kernel(hip_ops_factory, dtype):
    """gemm_wide_stream_kernel (csrc/gemm_wide_stream.hip, r06: the 256x256 tile as ONE generated assembly statement per
    workgroup - K stream continuous across tiles, bias as the accumulators' initial value, conversion + stores in assembly),
    forced through the diagnostics build on whole-tile shapes: exact-integer products must equal the other kernels BIT FOR BIT
    (K loops of 4, 5, 7 and 20 steps: both parities of the two-buffer loop, tiles per workgroup 1 .. 4: the cross-tile stream with
    odd K-step counts flips the buffer parity between tiles), + bias and GEGLU against the f32 reference, and the library's own
    rule sends the model's GEGLU projections there (pm_gemm_kernel_choice == 5)."""
    ops = hip_ops_factory(dtype, diag=True)
    g = torch.Generator().manual_seed(9)
    try:
        for M, N, K in [(256, 256, 256), (512, 768, 320), (8192, 8192, 448), (1024, 1024, 1280)]:
            a = torch.randint(-3, 4, (M, K), generator=g).to(dtype).cuda()
            w = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
            w[:, 1::3] *= 0
            w = w.cuda()
            outs = []
            for mode in (0, 2):
                ops.lib.pm_debug_gemm_wide(0)
                ops.lib.pm_debug_gemm_wstream(mode)
                outs.append(ops.gemm(a, w).float())
            assert torch.equal(outs[0], outs[1]), (M, N, K)
            assert torch.equal(outs[1][:64, :64].cpu(), (a[:64].float().cpu() @ w[:64].float().cpu().t()).to(dtype).float())
        M, N, K = 2048, 1536, 640
        a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
        bias = rnd(N, dtype=torch.float32, seed=3)
        da, dw, db = dev(a, w, bias)
        wgp, bgp = packing.pack_geglu(w, bias)
        ops.lib.pm_debug_gemm_wstream(2)
        assert rel_err(ops.gemm(da, dw, db), REF.gemm(a, w, bias, None, "none")) <= TOL[dtype]
        assert rel_err(ops.gemm(da, wgp.cuda(), bgp.cuda(), act="geglu"), _geglu_ref(a, w, bias)) <= TOL[dtype]
        ops.lib.pm_debug_gemm_wstream(1)
        ops.lib.pm_debug_gemm_wide(1)
        for shape in ((10240, 5120, 640), (2560, 10240, 1280), (40960, 2560, 320), (36864, 5120, 640)):
            assert ops.lib.pm_gemm_kernel_choice(*shape, 2, 0, ops.ws_bytes) == 5, shape  # GEGLU projections of both resolutions
        assert ops.lib.pm_gemm_kernel_choice(2560, 1280, 5120, 0, 0, ops.ws_bytes) != 5      # 50 tiles: split-K stays
    finally:
        ops.lib.pm_debug_gemm_wstream(1)
        ops.lib.pm_debug_gemm_wide(1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_ringw_256x128_tiles(hip_ops_factory, dtype):
    """gemm_ringw_kernel (the ring kernel on 256x128 tiles, csrc/gemm.hip), selected by the library's own rule
    (pm_gemm_kernel_choice == 3: unsplit, K >= 512, whole rounds of 256-row tiles): exact-integer product with a ragged last
    row tile, bias / GEGLU, f32 residual + fused GroupNorm statistics, and the temporal-conv loader incl. halo frames."""
    ops = hip_ops_factory(dtype)
    M, N, K = 49000, 128, 512  # 192 row tiles of 256 (the last one ragged: 104 rows) x 1 column tile
    assert ops.lib.pm_gemm_kernel_choice(M, N, K, 0, 0, ops.ws_bytes) == 3
    g = torch.Generator().manual_seed(11)
    a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
    w = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
    w[:, 1::3] *= 0
    want = a.float() @ w.float().t()
    assert torch.equal(ops.gemm(a.cuda(), w.cuda()).float().cpu(), want.to(dtype).float())
    # random operands: bias + 16-bit residual; f32 stream output + f32 residual + statistics; GEGLU
    a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    bias, res = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=dtype, seed=4)
    da, dw, db, dr = dev(a, w, bias, res)
    assert rel_err(ops.gemm(da, dw, db, dr), REF.gemm(a, w, bias, res, "none")) <= TOL[dtype]
    res32 = rnd(M, N, dtype=torch.float32, seed=5)
    got, tot = ops.gemm(da, dw, db, res32.cuda(), stream=True, stats=(1, 32))
    want32 = a.float() @ w.float().t() + bias + res32
    assert got.dtype == torch.float32 and rel_err(got, want32) <= 1e-4 + TOL[dtype] * 0.5
    assert rel_err(tot, TorchOps._with_stats(want32, (1, 32))[1]) <= 2e-3
    NG = 512  # (value | gate rows: 4 column tiles -> 768 tiles of 256x128 = three whole rounds on 256 CUs)
    wg, bg = rnd(NG, K, dtype=dtype, scale=K ** -0.5, seed=6), rnd(NG, dtype=torch.float32, seed=7)
    assert ops.lib.pm_gemm_kernel_choice(M, NG, K, 2, 0, ops.ws_bytes) == 3
    wgp, bgp = packing.pack_geglu(wg, bg)
    assert rel_err(ops.gemm(da, wgp.cuda(), bgp.cuda(), act="geglu"), _geglu_ref(a, wg, bg)) <= TOL[dtype]
    # temporal conv through the same kernel (K = 3 C = 768; 16 frames x 3072 pixels = 192 row tiles), with halo frames
    F, P, C = 16, 3072, 256
    x = rnd(F * P, C, dtype=dtype, seed=8)
    wt = rnd(C, C, 3, 1, 1, dtype=dtype, scale=(3 * C) ** -0.5, seed=9)
    bt = rnd(C, dtype=torch.float32, seed=10)
    lo, hi = rnd(P, C, dtype=dtype, seed=12), rnd(P, C, dtype=dtype, seed=13)
    wp = packing.pack_conv_t3(wt)
    for halo in (False, True):
        want_t = REF.conv_t3(x, wp, bt, F, P, halo_lo=lo if halo else None, halo_hi=hi if halo else None)
        got_t = ops.conv_t3(x.cuda(), wp.cuda(), bt.cuda(), F, P, halo_lo=lo.cuda() if halo else None,
                            halo_hi=hi.cuda() if halo else None)
        assert rel_err(got_t, want_t) <= TOL[dtype], halo


def _geglu_ref(a, w, b):
    y = a.float() @ w.float().t() + b
    n = y.shape[1] // 2
    return y[:, :n] * torch.nn.functional.gelu(y[:, n:])  # proj -> chunk -> x * gelu(gate)  (attention.py:415-422)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("NI,P,C,silu,eps", [(16, 150, 320, True, 1e-5), (1, 2000, 320, True, 1e-5),
                                             (4, 700, 64, False, 1e-6), (2, 33, 1920, True, 1e-5),
                                             (1, 16 * 40, 2560, False, 1e-6)])
def test_groupnorm(hip_ops_factory, dtype, NI, P, C, silu, eps):
    ops = hip_ops_factory(dtype)
    x = (rnd(NI * P, C, dtype=torch.float32, seed=1) * 2 + 0.7).to(dtype)
    gamma = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=2)
    beta = 0.3 * rnd(C, dtype=torch.float32, seed=3)
    want = REF.groupnorm(x, gamma, beta, eps, NI, silu)
    got = ops.groupnorm(x.cuda(), gamma.cuda(), beta.cuda(), eps, NI, silu)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_external_stats(hip_ops_factory, dtype):
    """Frame-sharded mode: two shards' partial sums combined == statistics over the whole clip."""
    ops = hip_ops_factory(dtype)
    P, C = 600, 320
    x = (rnd(2 * P, C, dtype=torch.float32, seed=1) + 0.5).to(dtype)
    gamma = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=2)
    beta = 0.3 * rnd(C, dtype=torch.float32, seed=3)
    want = REF.groupnorm(x, gamma, beta, 1e-5, 1, True)
    xs = [x[:P].cuda(), x[P:].cuda()]
    parts = [ops.groupnorm_stats(t, 1) for t in xs]
    total = (parts[0] + parts[1]).contiguous()
    count = 2 * P * (C // 32)
    got = torch.cat([ops.groupnorm_apply(t, total, gamma.cuda(), beta.cuda(), 1e-5, 1, True, count)
                     for t in xs], 0)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_apply_sums_f32_entries(hip_ops_factory, dtype):
    """pm_groupnorm_apply with n f32 entries per instance (bits 16..23 of out_dtype, no PM_TOTALS_I64): per-frame sums -> the
    (T,H,W) statistics of each clip without a reduce launch; equal to the pre-summed totals up to the summation order."""
    ops = hip_ops_factory(dtype)
    B, T, P, C = 2, 4, 96, 320
    x = (rnd(B * T * P, C, dtype=torch.float32, seed=1) + 0.5).to(dtype).cuda()
    gamma = (1 + 0.2 * rnd(C, dtype=torch.float32, seed=2)).cuda()
    beta = (0.3 * rnd(C, dtype=torch.float32, seed=3)).cuda()
    per_frame = _f32(ops.groupnorm_stats(x, B * T)).contiguous()  # [B T, 32, 2]
    want = REF.groupnorm(x.cpu(), gamma.cpu(), beta.cpu(), 1e-5, B, True)
    count = float(T * P * (C // 32))
    got = ops.groupnorm_apply(x, per_frame, gamma, beta, 1e-5, B, True, count)
    assert rel_err(got, want) <= TOL[dtype]
    pre = per_frame.view(B, T, 32, 2).sum(1).contiguous()
    ref = ops.groupnorm_apply(x, pre, gamma, beta, 1e-5, B, True, count)
    assert rel_err(got, ref.cpu()) <= 2e-3 * TOL[dtype] + 1e-6


def test_timestep_embedding(hip_ops_factory):
    """pm_timestep_embedding == the reference's cos / sin of t x (bf16-quantised) frequencies (utils_diffusion.py:8-28)."""
    from open_pandora_amd.unet import timestep_embedding, _FREQS
    ops = hip_ops_factory(torch.bfloat16)
    for t in (torch.tensor([999, 0, 417], dtype=torch.int64), torch.tensor([24.0, 3.0], dtype=torch.float32)):
        want = timestep_embedding(t, 320)
        freqs = _FREQS[(320, 10000, "cpu")].cuda()
        got = ops.timestep_embedding(t.cuda(), freqs)
        assert got.shape == want.shape
        assert (got.cpu() - want).abs().max().item() <= 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(333, 320), (50, 1280), (7, 512), (100, 640)])
def test_layernorm(hip_ops_factory, dtype, M, C):
    ops = hip_ops_factory(dtype)
    x = (rnd(M, C, dtype=torch.float32, seed=1) * 1.5 - 0.4).to(dtype)
    gamma = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=2)
    beta = 0.3 * rnd(C, dtype=torch.float32, seed=3)
    want = REF.layernorm(x, gamma, beta)
    got = ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda())
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,mode,nsplit", [
    (4200, 320, 960, "scale", 0), (4200, 320, 960, "plain", 3), (4133, 320, 320, "bias", 0),
    (4200, 320, 2560, "geglu", 0), (4200, 320, 2560, "geglu", 5), (4099, 320, 2560, "geglu", 3),
    (4100, 320, 640, "plain", 1), (12288, 320, 960, "scale", 2), (40960, 320, 960, "scale", 0)])
def test_ln_gemm_fused(hip_ops_factory, monkeypatch, dtype, M, K, N, mode, nsplit):
    """pm_ln_gemm (LayerNorm + projection in one panel kernel: norm1/2/3 -> to_q|k|v / to_q / GEGLU ff.net[0],
    attention.py:242-246) against the oracle's LayerNorm -> Linear, and against the two-kernel HIP pair; ragged
    last panels (M % 128 != 0), every column split the host may choose, all three epilogues."""
    ops = hip_ops_factory(dtype)
    assert ops.fused_ln and ops.lib.pm_ln_gemm_supported(M, N, K, 2 if mode == "geglu" else 0)
    if nsplit:
        monkeypatch.setenv("PANDORA_LNGEMM_NSPLIT", str(nsplit))
    x = rnd(M, K, dtype=torch.float32, scale=1.5, seed=1) - 0.4 + 2.0 * rnd(M, 1, dtype=torch.float32, seed=5)
    gamma = 1 + 0.2 * rnd(K, dtype=torch.float32, seed=2)
    beta = 0.3 * rnd(K, dtype=torch.float32, seed=3)
    w = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=4)
    bias = rnd(N, dtype=torch.float32, seed=6) if mode in ("bias", "geglu") else None
    scale = torch.cat([torch.full((N // 3,), 0.18033688), torch.ones(N - N // 3)]) if mode == "scale" else None
    act = "geglu" if mode == "geglu" else "none"
    want = REF.ln_gemm(x, gamma, beta, w, bias, act=act, col_scale=scale)
    dx, dg, db, dw, dbias, dscale = dev(x, gamma, beta, w, bias, scale)
    got = ops.ln_gemm(dx, dg, db, dw, dbias, act=act, col_scale=dscale)
    assert got.shape == want.shape and got.dtype == dtype
    assert rel_err(got, want) <= TOL[dtype]
    pair = ops.gemm(ops.layernorm(dx, dg, db), dw, dbias, act=act, col_scale=dscale)
    # same operands, same accumulation order: only a rare 1-ulp difference of a LayerNorm output (sum order) remains
    assert rel_err(got, pair) <= 0.25 * TOL[dtype]
    assert not ops.lib.pm_ln_gemm_supported(300, N, K, 0) and not ops.lib.pm_ln_gemm_supported(M, N, 640, 0)
    small = ops.ln_gemm(dx[:300], dg, db, dw, dbias, act=act, col_scale=dscale)  # unserved shape: the pair runs
    assert rel_err(small, want[:300]) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,N", [(2, 2, 200), (1, 5, 640), (3, 1, 40), (2, 5, 2560)])
def test_attention_self_fused_qkv(hip_ops_factory, dtype, B, heads, N):
    """Self-attention reading q, k, v as column slices of one fused [B, N, 3C] projection."""
    ops = hip_ops_factory(dtype)
    C = heads * 64
    qkv = rnd(B, N, 3 * C, dtype=dtype, scale=1.3, seed=1)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    want = REF.attention(q, k, v, heads)
    d = qkv.cuda()
    got = ops.attention(d[..., :C], d[..., C:2 * C], d[..., 2 * C:], heads)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_long_sequences_16x16x32_form(hip_ops_factory, dtype, prescaled):
    """Sequences of >= 4096 tokens run on attn_self16_kernel (csrc/attn16.hip, v_mfma_f32_16x16x32: two query rows per lane,
    the P^T operand built from the S accumulators of two 16-key blocks, V^T through paired transposed reads 16 keys apart)
    in the SHIPPED library: ragged length (masked last tile, query tile not full), a raise of the stale maximum in a fast
    tile (twice for one row) and in the ragged last tile, a dominant key in the first tile, both scaling modes."""
    ops = hip_ops_factory(dtype)
    B, heads, N = 1, 2, 4200  # 65 whole key tiles + 40 keys; 32 whole query tiles + 104 rows
    C = heads * 64
    q = rnd(B, N, C, dtype=torch.float32, scale=0.3 if prescaled else 1.0, seed=1)
    k = rnd(B, N, C, dtype=torch.float32, seed=2)
    v = rnd(B, N, C, dtype=torch.float32, seed=3)
    amp = (lambda a: 2.0 * a) if prescaled else (lambda a: a)  # (similar base-2 score magnitudes in both modes)
    k[0, 70] = amp(3 * q[0, 5])
    k[0, 1300] = amp(6 * q[0, 5])
    k[0, 4199] = amp(5 * q[0, 4180])
    k[0, 10] = amp(8 * q[0, 300])
    k[0, 4170] = amp(4 * q[0, 4190])
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    want = _attn_ref_base2(q, k, v, heads) if prescaled else REF.attention(q, k, v, heads)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, prescaled=prescaled)
    assert rel_err(got, want) <= TOL[dtype]
    for i in (5, 4180, 300, 4190, 6, 4199):
        assert rel_err(got[0, i], want[0, i]) <= 2 * TOL[dtype], i


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_peaked_scores(hip_ops_factory, dtype):
    """Force the running max to jump late: one key in the LAST tile dominates a few query rows."""
    ops = hip_ops_factory(dtype)
    B, heads, N = 1, 2, 300
    C = heads * 64
    q = rnd(B, N, C, dtype=torch.float32, seed=1)
    k = rnd(B, N, C, dtype=torch.float32, seed=2)
    v = rnd(B, N, C, dtype=torch.float32, seed=3)
    k[0, 290] = 6 * q[0, 17]
    k[0, 3] = 5 * q[0, 200]
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    want = REF.attention(q, k, v, heads)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,N", [(16, 5, 160), (4, 20, 40)])
def test_attention_cross_text_image(hip_ops_factory, dtype, B, heads, N):
    """Text keys (77, shared by all frames: batch stride 0) + per-frame image keys (16)."""
    ops = hip_ops_factory(dtype)
    C = heads * 64
    q = rnd(B, N, C, dtype=dtype, seed=1)
    kt, vt = rnd(1, 77, C, dtype=dtype, seed=2), rnd(1, 77, C, dtype=dtype, seed=3)
    ki, vi = rnd(B, 16, C, dtype=dtype, seed=4), rnd(B, 16, C, dtype=dtype, seed=5)
    want = REF.attention(q, kt, vt, heads, ki, vi, 1.0)
    dq, dkt, dvt, dki, dvi = dev(q, kt, vt, ki, vi)
    got = ops.attention(dq, dkt, dvt, heads, dki, dvi, 1.0)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Fq,Fk,P,heads", [(16, 16, 37, 5), (2, 16, 50, 8), (16, 16, 640, 10)])
def test_attention_temporal(hip_ops_factory, dtype, Fq, Fk, P, heads):
    ops = hip_ops_factory(dtype)
    C = heads * 64
    qkv = rnd(Fk, P, 3 * C, dtype=dtype, scale=1.2, seed=1)
    q, k, v = qkv[:Fq, :, :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    want = REF.attention_temporal(q, k, v, heads)
    d = qkv.cuda()
    got = ops.attention_temporal(d[:Fq, :, :C], d[..., C:2 * C], d[..., 2 * C:], heads)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemv(hip_ops_factory, dtype):
    ops = hip_ops_factory(dtype)
    w = rnd(1000, 1280, dtype=dtype, scale=1280 ** -0.5, seed=1)
    x, b = rnd(1280, dtype=torch.float32, seed=2), rnd(1000, dtype=torch.float32, seed=3)
    for silu_in, act in [(False, "none"), (True, "none"), (False, "silu")]:
        want = REF.gemv(w, x, b, silu_in, act)
        got = ops.gemv(w.cuda(), x.cuda(), b.cuda(), silu_in, act)
        assert rel_err(got, want) <= 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_ddim_update_and_layout(hip_ops_factory, dtype):
    ops = hip_ops_factory(dtype)
    C, F, P = 4, 16, 60
    x = rnd(C, F, P, dtype=torch.float32, seed=1)
    cond = rnd(C, F, P, dtype=torch.float32, seed=2)
    ec, eu = rnd(C, F, P, dtype=dtype, seed=3), rnd(C, F, P, dtype=dtype, seed=4)
    noise = rnd(C, F, P, dtype=torch.float32, seed=5)
    args = (4.0, 0.83, 0.55, 0.97, 0.9, 0.31, 0.2)
    want_p, want_0 = REF.ddim_update(x, ec, eu, noise, *args)
    got_p, got_0 = ops.ddim_update(x.cuda(), ec.cuda(), eu.cuda(), noise.cuda(), *args)
    assert rel_err(got_p, want_p) <= 1e-6 and rel_err(got_0, want_0) <= 1e-6
    packed = ops.pack_input(x.cuda(), cond.cuda())
    assert torch.equal(packed.cpu(), REF.pack_input(x, cond).to(dtype))
    y = rnd(F * P, C, dtype=dtype, seed=6)
    assert torch.equal(ops.unpack_output(y.cuda(), F, P).cpu(), REF.unpack_output(y, F, P))


def test_bad_arguments_raise(hip_ops_factory):
    from open_pandora_amd import capi
    ops = hip_ops_factory(torch.float16)
    a = rnd(16, 72, dtype=torch.float16, seed=1).cuda()  # K = 72 is not a multiple of 64
    w = rnd(8, 72, dtype=torch.float16, seed=2).cuda()
    out = torch.empty(16, 8, dtype=torch.float16, device="cuda")
    # the C-ABI refuses a K that is not whole 64-wide tiles ...
    rc = ops.lib.pm_gemm(a.data_ptr(), 72, w.data_ptr(), 72, None, None, 0, out.data_ptr(), 8, 16, 8, 72, 0, 0,
                         capi.PM_F16, None, 0, None, torch.cuda.current_stream().cuda_stream)
    assert rc == -2
    with pytest.raises(capi.PandoraKernelError):
        capi.check(rc, "pm_gemm K=72")
    with pytest.raises(capi.PandoraKernelError):  # (and HipOps surfaces any refusal as an exception)
        ops.layernorm(torch.zeros(4, 12, dtype=torch.float16).cuda(), torch.ones(12).cuda(), torch.zeros(12).cuda())
    # ... the op table serves it by zero-padding both operands along K (reduced-width first-stage encoder)
    assert rel_err(ops.gemm(a, w), a.float().cpu() @ w.float().cpu().t()) <= TOL[torch.float16]


# ---- f32 residual stream variants (PM_FLAG_A_F32 / PM_FLAG_OUT_F32, f32 norm inputs) -------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_f32_stream(hip_ops_factory, dtype):
    """A operand from the f32 stream (rounded to 16 bit while staging), f32 residual and output."""
    ops = hip_ops_factory(dtype)
    M, N, K = 300, 320, 640
    a32 = rnd(M, K, dtype=torch.float32, seed=1)
    w = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    bias = rnd(N, dtype=torch.float32, seed=3)
    res32 = rnd(M, N, dtype=torch.float32, scale=3.0, seed=4)
    want = REF.gemm(a32.to(dtype), w, bias, res32)  # operand rounding is part of the contract
    got = ops.gemm(a32.cuda(), w.cuda(), bias.cuda(), res32.cuda(), stream=True)
    assert got.dtype == torch.float32
    assert rel_err(got, want) <= 2e-5
    got16 = ops.gemm(a32.cuda(), w.cuda(), bias.cuda())  # f32 A, 16-bit output
    assert got16.dtype == dtype and rel_err(got16, REF.gemm(a32.to(dtype), w, bias)) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_f32_stream(hip_ops_factory, dtype):
    ops = hip_ops_factory(dtype)
    F, H, W, Cin, Cout = 2, 10, 16, 64, 128
    x32 = rnd(F * H * W, Cin, dtype=torch.float32, seed=1)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=2)
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    wp = packing.pack_conv3x3(w)
    for stride, ups in ((2, False), (1, True), (1, False)):
        want = REF.conv3x3(x32.to(dtype), wp, bias, F, H, W, stride, ups)
        res32 = rnd(want.shape[0], Cout, dtype=torch.float32, seed=4)
        got = ops.conv3x3(x32.cuda(), wp.cuda(), bias.cuda(), F, H, W, stride, ups, residual=res32.cuda(), stream=True)
        assert got.dtype == torch.float32 and rel_err(got, want + res32) <= 2e-5
    # temporal conv: 16-bit operand, f32 residual + output (TemporalConvBlock identity + x)
    P, C = 50, 128
    xt = rnd(4 * P, C, dtype=dtype, seed=5)
    wt = packing.pack_conv_t3(rnd(C, C, 3, 1, 1, dtype=dtype, scale=(3 * C) ** -0.5, seed=6))
    r32 = rnd(4 * P, C, dtype=torch.float32, seed=7)
    want = REF.conv_t3(xt, wt, bias, 4, P, residual=r32)
    got = ops.conv_t3(xt.cuda(), wt.cuda(), bias.cuda(), 4, P, residual=r32.cuda(), stream=True)
    assert got.dtype == torch.float32 and rel_err(got, want) <= 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms_f32_input(hip_ops_factory, dtype):
    ops = hip_ops_factory(dtype)
    C = 320
    gamma = 1 + 0.2 * rnd(C, dtype=torch.float32, seed=2)
    beta = 0.3 * rnd(C, dtype=torch.float32, seed=3)
    x = rnd(16 * 150, C, dtype=torch.float32, seed=1) * 2 + 0.7
    for NI, silu in ((16, True), (1, False)):
        want = REF.groupnorm(x, gamma, beta, 1e-5, NI, silu)
        got = ops.groupnorm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, NI, silu)
        assert got.dtype == dtype and rel_err(got, want) <= TOL[dtype]
    want = REF.layernorm(x[:333], gamma, beta)
    got = ops.layernorm(x[:333].cuda(), gamma.cuda(), beta.cuda())
    assert got.dtype == dtype and rel_err(got, want) <= TOL[dtype]


def test_ddim_update_f32_model_output(hip_ops_factory):
    ops = hip_ops_factory(torch.float16)
    C, F, P = 4, 16, 60
    x = rnd(C, F, P, dtype=torch.float32, seed=1)
    ec, eu = rnd(C, F, P, dtype=torch.float32, seed=3), rnd(C, F, P, dtype=torch.float32, seed=4)
    args = (4.0, 0.83, 0.55, 0.97, 0.9, 0.31, 0.0)
    want_p, _ = REF.ddim_update(x, ec, eu, None, *args)
    got_p, _ = ops.ddim_update(x.cuda(), ec.cuda(), eu.cuda(), None, *args)
    assert rel_err(got_p, want_p) <= 1e-6
    y = rnd(F * P, C, dtype=torch.float32, seed=6)
    assert torch.equal(ops.unpack_output(y.cuda(), F, P).cpu(), REF.unpack_output(y, F, P))


# ---- split-K (few output tiles, long K: the deep U-Net levels) ----------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_splitk_matches_unsplit_and_oracle(hip_ops_factory, dtype):
    from open_pandora_amd import capi
    from open_pandora_amd.ops_hip import HipOps
    ops = hip_ops_factory(dtype)
    nows = HipOps(dtype, "cuda:0", workspace_mb=0)  # NULL workspace => the same call runs unsplit
    lib = capi.load()
    # dense: M=640, N=1280, K=5120 (ff2 at the deepest level)
    M, N, K = 640, 1280, 5120
    assert lib.pm_gemm_workspace_bytes(M, N, K, 0) > 0
    a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    bias, res32 = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=torch.float32, seed=4)
    want = REF.gemm(a, w, bias, res32)
    got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), res32.cuda(), stream=True)
    ref2 = nows.gemm(a.cuda(), w.cuda(), bias.cuda(), res32.cuda(), stream=True)
    assert rel_err(got, want) <= 2e-5 and rel_err(got, ref2) <= 2e-6
    got16 = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), act="silu")
    assert rel_err(got16, REF.gemm(a, w, bias, act="silu")) <= TOL[dtype]
    # conv3x3 at the deepest level: 16 x 5 x 8 pixels, 1280 -> 1280 channels (K = 11520)
    F, H, W, C = 16, 5, 8, 1280
    assert lib.pm_gemm_workspace_bytes(F * H * W, C, 9 * C, 0) > 0
    x = rnd(F * H * W, C, dtype=dtype, seed=5)
    wp = rnd(C, 9 * C, dtype=dtype, scale=(9 * C) ** -0.5, seed=6)
    b2 = rnd(C, dtype=torch.float32, seed=7)
    want = REF.conv3x3(x, wp, b2, F, H, W)
    got = ops.conv3x3(x.cuda(), wp.cuda(), b2.cuda(), F, H, W)
    assert rel_err(got, want) <= TOL[dtype]
    assert rel_err(got, nows.conv3x3(x.cuda(), wp.cuda(), b2.cuda(), F, H, W)) <= TOL[dtype]
    # temporal conv, same level
    wt = rnd(C, 3 * C, dtype=dtype, scale=(3 * C) ** -0.5, seed=8)
    want = REF.conv_t3(x, wt, b2, F, H * W)
    got = ops.conv_t3(x.cuda(), wt.cuda(), b2.cuda(), F, H * W)
    assert rel_err(got, want) <= TOL[dtype]


# ---- full-size M: the persistent grid (several work items per workgroup, cross-tile prefetch) ----------
@pytest.mark.parametrize("dtype", DTYPES)
def test_large_m_gemm_conv_tconv(hip_ops_factory, dtype):
    """Full-size M (40960 rows).  Also the shapes that take the optional 256x128 3-stage tile
    are the grids of more than 2 x 256 tiles on which a workgroup walks several work items."""
    ops = hip_ops_factory(dtype)
    # dense: M = 40960 (+ ragged tail), N = 320 (2.5 column tiles), K = 320 and 1280, every epilogue
    M = 40960 + 77
    for K, N, act in ((320, 320, "none"), (1280, 320, "silu"), (320, 2560, "geglu")):
        a = rnd(M, K, dtype=dtype, seed=1)
        w = rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
        bias = rnd(N, dtype=torch.float32, seed=3)
        res = None if act == "geglu" else rnd(M, N, dtype=torch.float32, seed=4)
        want = REF.gemm(a, w, bias, res, act)
        got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), None if res is None else res.cuda(), act,
                       stream=res is not None)
        tol = 2e-5 if res is not None and act == "none" else TOL[dtype]
        assert rel_err(got, want) <= max(tol, 2e-5), (K, N, act, rel_err(got, want))
    # exact-integer check through the big tile (fragment/stage bookkeeping)
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-2, 3, (33000, 192), generator=g).to(dtype)
    w = torch.randint(-2, 3, (256, 192), generator=g).to(dtype)
    assert torch.equal(ops.gemm(a.cuda(), w.cuda()).float().cpu(), (a.float() @ w.float().t()).to(dtype).float())
    # conv3x3 16 x 40 x 64, 64 -> 256 channels (padding, frame borders inside 256-row tiles)
    F, H, W, Cin, Cout = 16, 40, 64, 64, 256
    x = rnd(F * H * W, Cin, dtype=dtype, seed=5)
    wp = rnd(Cout, 9 * Cin, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=6)
    b2 = rnd(Cout, dtype=torch.float32, seed=7)
    assert rel_err(ops.conv3x3(x.cuda(), wp.cuda(), b2.cuda(), F, H, W), REF.conv3x3(x, wp, b2, F, H, W)) <= TOL[dtype]
    want = REF.conv3x3(x, wp, b2, F, H, W, stride=2)
    assert want.shape[0] == 16 * 20 * 32
    # temporal conv 16 frames x 2560 pixels, 320 channels
    xt = rnd(16 * 2560, 320, dtype=dtype, seed=8)
    wt = rnd(320, 3 * 320, dtype=dtype, scale=960 ** -0.5, seed=9)
    b3 = rnd(320, dtype=torch.float32, seed=10)
    assert rel_err(ops.conv_t3(xt.cuda(), wt.cuda(), b3.cuda(), 16, 2560), REF.conv_t3(xt, wt, b3, 16, 2560)) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_self_64_rows_per_wave(hip_ops_factory, dtype):
    """Large grids take the 256-query-row workgroup (two 32-row blocks per wave); ragged N and a
    peaked row in the last tile."""
    ops = hip_ops_factory(dtype)
    B, heads, N = 16, 16, 2100  # 9 x 256 = 2304 workgroups
    C = heads * 64
    qkv = rnd(B, N, 3 * C, dtype=torch.float32, scale=1.2, seed=1)
    qkv[3, 2090, C:2 * C] = 5 * qkv[3, 77, :C]  # key 2090 dominates query 77 of batch 3 (all heads)
    qkv[5, 1000, C:2 * C] = 4 * qkv[5, 300, :C]  # a raise of the stale maximum in a fast (unmasked) tile
    qkv = qkv.to(dtype)
    want = REF.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    d = qkv.cuda()
    got = ops.attention(d[..., :C], d[..., C:2 * C], d[..., 2 * C:], heads)
    assert rel_err(got, want) <= TOL[dtype]


def _attn_ref_base2(q, k, v, heads):
    """softmax_2(q k^T) v per head in f64 (what pm_attention computes for scale = ln 2)."""
    B, N, C = q.shape
    sp = lambda t: t.double().reshape(t.shape[0], t.shape[1], heads, 64).permute(0, 2, 1, 3)
    s = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * 0.6931471805599453
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), sp(v)).permute(0, 2, 1, 3).reshape(B, N, C).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [0, 1, 3, 5, 16])
def test_attention_stale_max_paths(hip_ops_factory, dtype, variant):
    """attn_self_kernel keeps a STALE running maximum that is only raised when a tile outgrows it by 2^6: force
    every branch (cdna guide rule 26) - a raise in a fast (unmasked, not first) tile, twice for the same row; a
    raise in the last fast tile; rows whose scores stay far BELOW the first tile's maximum; a ragged masked
    last tile; rows that never raise - in all kernel variants (32 / 64 rows per wave, pipelined P.V; 16 = the 16x16x32-MFMA
    form, csrc/attn16.hip; 0 = the shipped library's own choice, 1 / 16 force the two production forms)."""
    # (variant 0 = the shipped library; the other variants exist only in the diagnostics build, include/pandora_mi355x_diag.h)
    ops = hip_ops_factory(dtype) if variant == 0 else hip_ops_factory(dtype, diag=True)
    B, heads = 2, 3
    C = heads * 64
    try:
        for N in (384, 450):  # 6 whole tiles; 7 full + 1 ragged
            q = rnd(B, N, C, dtype=torch.float32, seed=1)
            k = rnd(B, N, C, dtype=torch.float32, seed=2)
            v = rnd(B, N, C, dtype=torch.float32, seed=3)
            k[0, 70] = 3 * q[0, 5]      # query 5: raise in tile 1 ...
            k[0, 200] = 6 * q[0, 5]     # ... and again in tile 3 (both fast tiles)
            k[1, 383] = 5 * q[1, 100]   # last key of tile 5
            k[1, 10] = 8 * q[1, 300]    # dominant key in the FIRST tile: every later tile is far below m
            k[0, N - 1] = 4 * q[0, 77]  # last key of the clip (masked tile when N = 450)
            q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
            want = REF.attention(q, k, v, heads)
            if variant:
                ops.lib.pm_debug_attn_variant(variant)
            got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads)
            assert rel_err(got, want) <= TOL[dtype], (N, variant)
            for (b, i) in ((0, 5), (1, 100), (1, 300), (0, 77), (0, 6)):  # the forced rows one by one
                assert rel_err(got[b, i], want[b, i]) <= 2 * TOL[dtype], (N, variant, b, i)
    finally:
        if variant:
            ops.lib.pm_debug_attn_variant(0)


# e4m3 operands (3 mantissa bits) for q, k, v AND the probabilities; an exact-arithmetic emulation of those four
# roundings gives 5.4e-2 on N(0,1) data (q.k 4.0e-2, P 2.4e-2, V 2.7e-2 alone): stated separately (configs[4])
FP8_TOL = 8e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,N", [(2, 5, 640), (3, 2, 333), (1, 10, 64), (16, 5, 2100)])
def test_attention_fp8(hip_ops_factory, dtype, B, heads, N):
    """pm_attention_fp8 (block-scaled MFMA, K = 64): layout exactness is covered by the structure of the error -
    a wrong key permutation or operand pairing gives O(1) error, fp8 rounding a few 1e-2; ragged N, a peaked row in
    a fast tile and one in the ragged last tile."""
    ops = hip_ops_factory(dtype)
    C = heads * 64
    qkv = rnd(B, N, 3 * C, dtype=torch.float32, scale=1.0, seed=1)
    if N > 200:
        qkv[0, 150, C:2 * C] = 3 * qkv[0, 7, :C]      # raise of the stale maximum in a fast tile
        qkv[B - 1, N - 1, C:2 * C] = 3 * qkv[B - 1, 20, :C]  # ... in the last (possibly masked) tile
    qkv = qkv.to(dtype)
    want = REF.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    d = qkv.cuda()
    got = ops.attention_fp8(d[..., :C], d[..., C:2 * C], d[..., 2 * C:], heads)
    err = rel_err(got, want)
    print(f"\n[fp8] B={B} heads={heads} N={N} {dtype}: rel err {err:.2e}")
    assert err <= FP8_TOL
    # per-row check: no row may be grossly wrong (a layout error on a subset of lanes hides in a norm)
    rowerr = (got.float().cpu() - want).norm(dim=-1) / want.norm(dim=-1).clamp_min(1e-6)
    assert rowerr.max().item() < 0.7, rowerr.max().item()  # (an operand-order error decorrelates a row: ~1.4)


def test_attention_fp8_exact_on_fp8_representable_inputs(hip_ops_factory):
    """With q, k, v on the e4m3 grid and a softmax that is exactly one-hot (one dominant key per row), the only
    rounding left is the output's: any operand-order mistake shows as a wrong row, not as noise."""
    ops = hip_ops_factory(torch.float16)
    g = torch.Generator().manual_seed(3)
    B, heads, N = 2, 3, 192
    C = heads * 64
    grid = torch.tensor([-2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0])
    v = grid[torch.randint(0, 9, (B, N, C), generator=g)]
    k = torch.zeros(B, N, C)
    q = torch.zeros(B, N, C)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B * heads)]).reshape(B, heads, N)
    for b in range(B):
        for h in range(heads):
            # query i points at key perm[i] through a one-hot code of 8 bits over 64 dims (scores 0 or 256)
            bits = ((torch.arange(N)[:, None] >> torch.arange(8)[None]) & 1).float()          # [N, 8]
            code = torch.cat([bits, 1 - bits], 1).repeat(1, 4) * 4.0                           # [N, 64], 32 ones x 4
            k[b, :, h * 64:(h + 1) * 64] = code
            q[b, :, h * 64:(h + 1) * 64] = code[perm[b, h]] * 4.0
    want = torch.stack([torch.cat([v[b, perm[b, h], h * 64:(h + 1) * 64] for h in range(heads)], -1) for b in range(B)])
    got = ops.attention_fp8(q.half().cuda(), k.half().cuda(), v.half().cuda(), heads)
    assert rel_err(got, want) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,D,Nq,Nk", [(2, 16, 80, 257, 257), (1, 3, 128, 5, 1000), (3, 2, 8, 70, 64)])
def test_attention_generic_head_dims(hip_ops_factory, dtype, B, heads, D, Nq, Nk):
    """pm_attention_generic: the ViT-H/14 tower's 16 x 80 heads over 257 tokens, and the edges of its domain."""
    ops = hip_ops_factory(dtype)
    C = heads * D
    q, k, v = rnd(B, Nq, C, dtype=dtype, seed=1), rnd(B, Nk, C, dtype=dtype, scale=1.3, seed=2), rnd(B, Nk, C, dtype=dtype, seed=3)
    k[0, Nk - 1, :D] = 3 * q[0, 2, :D]  # a peaked row
    want = REF.attention_generic(q, k, v, heads)
    got = ops.attention_generic(q.cuda(), k.cuda(), v.cuda(), heads)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_prescaled_q(hip_ops_factory, dtype):
    """scale = ln 2 (scale * log2 e == 1): q is taken as already multiplied by 64^-1/2 log2 e, the kernel
    applies no scaling of its own (the U-Net folds the factor into the to_q weights)."""
    ops = hip_ops_factory(dtype)
    B, heads, N = 3, 5, 333
    C = heads * 64
    q = rnd(B, N, C, dtype=dtype, scale=0.3, seed=1)
    k, v = rnd(B, N, C, dtype=dtype, scale=1.2, seed=2), rnd(B, N, C, dtype=dtype, seed=3)
    want = _attn_ref_base2(q, k, v, heads)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, prescaled=True)
    assert rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_f32_residual_16bit_out(hip_ops_factory, dtype):
    """PM_FLAG_RES_F32: f32 residual added in f32, 16-bit store (also through the split-K reduce)."""
    ops = hip_ops_factory(dtype)
    for M, N, K in ((300, 320, 1280), (640, 1280, 5120)):
        a, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
        bias, res32 = rnd(N, dtype=torch.float32, seed=3), rnd(M, N, dtype=torch.float32, scale=3.0, seed=4)
        got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), res32.cuda())
        assert got.dtype == dtype and rel_err(got, REF.gemm(a, w, bias, res32)) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_rows_and_latent_affine(hip_ops_factory, dtype):
    ops = hip_ops_factory(dtype)
    s = rnd(300, 2560, dtype=torch.float32, scale=30.0, seed=1)
    s[17, 2000] = 400.0  # one dominant entry
    got = ops.softmax_rows(s.cuda(), 512 ** -0.5)
    assert rel_err(got, REF.softmax_rows(s, 512 ** -0.5)) <= TOL[dtype]
    x = rnd(4, 3, 70, dtype=torch.float32, seed=2)
    W, b = rnd(4, 4, dtype=torch.float32, seed=3), rnd(4, dtype=torch.float32, seed=4)
    got = ops.latent_affine(x.cuda(), W.cuda(), b.cuda(), 1 / 0.18215)
    assert got.shape == (210, 8) and rel_err(got, REF.latent_affine(x, W, b, 1 / 0.18215)) <= TOL[dtype]
    assert got[:, 4:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_asymmetric_pad_stride2(hip_ops_factory, dtype):
    """pad_lo = 0: the (0,1,0,1) zero padding of the first-stage encoder's Downsample."""
    ops = hip_ops_factory(dtype)
    for F, H, W, Cin, Cout in ((2, 8, 6, 64, 64), (1, 16, 24, 128, 128), (2, 8, 8, 8, 32)):
        x = rnd(F * H * W, Cin, dtype=dtype, seed=1)
        w = rnd(Cout, Cin, 3, 3, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=2)
        bias = rnd(Cout, dtype=torch.float32, seed=3)
        xi = torch.nn.functional.pad(x.float().reshape(F, H, W, Cin).permute(0, 3, 1, 2), (0, 1, 0, 1))
        want = torch.nn.functional.conv2d(xi, w.float(), bias, stride=2).permute(0, 2, 3, 1).reshape(-1, Cout)
        got = ops.conv3x3(x.cuda(), packing.pack_conv3x3(w).cuda(), bias.cuda(), F, H, W, stride=2, pad_lo=0)
        assert got.shape == want.shape and rel_err(got, want) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_groupnorm_statistics(hip_ops_factory, dtype):
    """`stats=(NI, groups)`: the epilogue's column sums -> GroupNorm totals equal a statistics pass over
    the stored output (per-frame and (T,H,W); conv, temporal conv, GEMM; ragged instance -> fallback)."""
    ops = hip_ops_factory(dtype)
    F, H, W, Cin, Cout = 4, 16, 16, 64, 320            # 256 rows per frame = 2 row tiles
    x = rnd(F * H * W, Cin, dtype=dtype, seed=1)
    wp = rnd(Cout, 9 * Cin, dtype=dtype, scale=(9 * Cin) ** -0.5, seed=2)
    b = rnd(Cout, dtype=torch.float32, seed=3)
    res = rnd(F * H * W, Cout, dtype=torch.float32, seed=4)
    for NI in (F, 1):
        out, tot = ops.conv3x3(x.cuda(), wp.cuda(), b.cuda(), F, H, W, residual=res.cuda(), stream=True, stats=(NI, 32))
        want_out, want_tot = REF.conv3x3(x, wp, b, F, H, W, residual=res, stats=(NI, 32))
        assert rel_err(out, want_out) <= 2e-5 and _f32(tot).shape == (NI, 32, 2)
        assert rel_err(tot, want_tot) <= 1e-4
        assert rel_err(tot, ops.groupnorm_stats(out, NI)) <= 2e-6   # == the unfused statistics pass
        gamma = 1 + 0.2 * rnd(Cout, dtype=torch.float32, seed=5)
        beta = 0.3 * rnd(Cout, dtype=torch.float32, seed=6)
        y = ops.groupnorm(out, gamma.cuda(), beta.cuda(), 1e-5, NI, True, totals=tot)
        assert rel_err(y, REF.groupnorm(want_out, gamma, beta, 1e-5, NI, True)) <= TOL[dtype]
    xt = rnd(4 * 128, 128, dtype=dtype, seed=7)
    wt = rnd(128, 3 * 128, dtype=dtype, scale=384 ** -0.5, seed=8)
    out, tot = ops.conv_t3(xt.cuda(), wt.cuda(), None, 4, 128, stream=True, stats=(1, 32))
    assert rel_err(tot, ops.groupnorm_stats(out, 1)) <= 2e-6
    out, tot = ops.gemm(xt.cuda(), wt[:, :128].contiguous().cuda(), stats=(4, 32))  # 16-bit output:
    assert rel_err(tot, ops.groupnorm_stats(out, 4)) <= 1e-3  # totals are of the f32 values before rounding
    # 100 rows per instance: not a whole number of 128-row tiles -> ordinary statistics pass
    xr = rnd(3 * 100, 64, dtype=dtype, seed=9)
    out, tot = ops.gemm(xr.cuda(), rnd(64, 64, dtype=dtype, seed=10).cuda(), stats=(3, 32))
    assert _f32(tot).shape == (3, 32, 2) and rel_err(tot, ops.groupnorm_stats(out, 3)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_stats_from_the_splitk_reduce(hip_ops_factory, dtype):
    """Shapes that split over K emit the GroupNorm column sums from the reduce pass (16-row blocks): the deep
    levels' convs.  Totals must equal a statistics pass over the stored output; per-frame works when a frame is a
    whole number of 16-row blocks (160 pixels at level 2)."""
    ops = hip_ops_factory(dtype)
    F, H, W, C = 16, 10, 16, 256  # 2560 rows, 160 per frame; K = 2304: few tiles, long K -> split
    assert ops.lib.pm_gemm_colstats_rows(F * H * W, C, 9 * C, 0, ops.ws_bytes) == 16
    x = rnd(F * H * W, C, dtype=dtype, seed=1)
    wp = rnd(C, 9 * C, dtype=dtype, scale=(9 * C) ** -0.5, seed=2)
    b = rnd(C, dtype=torch.float32, seed=3)
    res = rnd(F * H * W, C, dtype=torch.float32, seed=4)
    for NI in (1, F):
        out, tot = ops.conv3x3(x.cuda(), wp.cuda(), b.cuda(), F, H, W, residual=res.cuda(), stream=True, stats=(NI, 32))
        want_out, want_tot = REF.conv3x3(x, wp, b, F, H, W, residual=res, stats=(NI, 32))
        assert rel_err(out, want_out) <= TOL[dtype]
        assert _f32(tot).shape == (NI, 32, 2) and rel_err(tot, ops.groupnorm_stats(out, NI)) <= 2e-6
    xt = rnd(16 * 40, 1280, dtype=dtype, seed=5)  # temporal conv at the deepest level: 640 rows
    wt = rnd(256, 3 * 1280, dtype=dtype, scale=3840 ** -0.5, seed=6)
    out, tot = ops.conv_t3(xt.cuda(), wt.cuda(), None, 16, 40, stream=True, stats=(1, 32))
    assert rel_err(tot, ops.groupnorm_stats(out, 1)) <= 2e-6
    out, tot = ops.gemm(xt.cuda(), wt[:, :1280].contiguous().cuda(), stats=(1, 32))  # 16-bit output
    assert rel_err(tot, ops.groupnorm_stats(out, 1)) <= 1e-3

"""Generator of gemm_wide_stream_kernel's per-tile assembly (csrc/gemm_wide_stream.hip -> gemm_wide_stream.inc): the 256 x 256 /
128 x 128-wave-tile kernel of tools/gen_wide_loop.py with the K STREAM CONTINUOUS ACROSS TILES and the epilogue in assembly.

Why (profiles/r06/wide_kernel_stamps.txt): gemm_wide pays ~5 700 cycles of prologue (two exposed load round trips behind the
previous tile's store drain - vmcnt retires in order) and 12 000-17 000 cycles of epilogue per tile, ~2 370 per K-step: at K = 640
the fixed costs are 43 % of a tile.  Here the last three K-steps of a tile request the NEXT tile's first K-steps (the staging
pipeline never drains: no prologue but the workgroup's first), the epilogue's stores are YOUNGER than those requests in the vmcnt
queue (the next tile's first waits count past them: the store drain runs under the next K loop), the bias is the accumulators'
INITIAL value (loaded under the epilogue for the next tile) and the epilogue itself is straight-line assembly on the registers of
fragment set 1, free at a tile boundary.  Flavours: 16-bit output, no residual, no statistics, bias or none ("lin"), GEGLU;
whole 256 x 256 tiles only (M % 256 == 0, N % 256 == 0), K >= 256.

    python tools/gen_wide_stream.py > open-pandora_amd/csrc/gemm_wide_stream.inc

Registers (all clobbered): v48-v55 A row offsets, v56-v63 W row offsets (per tile: lane base + tile offset + j x stride),
v64-v127 / v128-v191 fragment sets 0 / 1 (the epilogue uses set 1: v128-v159 conversion temporaries, v160-v191 the next tile's
bias), v192-v255 staging set P, a0-a255 accumulators, s90-s99 scalars.  v0-v47 stay the compiler's.  One K-step = the schedule
"B" of gen_wide_loop.py.  K-step s of a tile: first half stages W(s+1) and requests W(s+2), second half stages A(s+2) and requests
A(s+3); steps nk-3 / nk-2 / nk-1 switch the A (second half of nk-3) and W (first half of nk-2) offsets to the next tile."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from gen_wide_loop import ABUF, WBUF, FRAG_ORDER, NQ, Emit, acc, c_string, frag_a, frag_w, preg, read_frag, vr, write_piece  # noqa: E402

AO, BO = 48, 56          # v48.. A row offsets, v56.. W row offsets
T0, BIAS = 128, 160      # epilogue temporaries / next tile's bias (fragment set 1)
TA, TW, TC, TB, TAN, TWN, TCN, TBN, TIDX, PAR, STMP2 = ("s%d" % r for r in range(70, 81))
KW, KA, NCNT, STMP, SROW = "s90", "s91", "s92", "s93", "s94"
GC = ["s82", "s84", "s86", "s88", "s96"]  # gelu_q_exp2's coefficients c4..c0, each the low half of an aligned pair (VOP3P source;
#                                          op_sel_hi = 0 broadcasts it); c5 sits in a VGPR pair: VOP3 takes no literal on gfx9


class E2(Emit):
    def vm_store(self, text):
        self.ins(text)
        self.vm.append(-1)


def load_piece(e, q):
    if q < 8:
        e.vm_load(f"buffer_load_dwordx4 {vr(preg(q))}, v{AO + q}, %[adesc], {KA} offen", q)
    else:
        e.vm_load(f"buffer_load_dwordx4 {vr(preg(q))}, v{BO + q - 8}, %[wdesc], {KW} offen", q)


def set_a_offsets(e, t):
    """v48.. = lane base + tile offset %[t] + j * stride (8 x v_add_u32, the scalar walk in STMP)"""
    e.ins(f"s_mov_b32 {STMP}, {t}")
    for j in range(8):
        e.ins(f"v_add_u32 v{AO + j}, %[aob], {STMP}")
        if j < 7:
            e.ins(f"s_add_u32 {STMP}, {STMP}, %[sa]")


def set_w_offsets(e, t):
    """v56.. : LDS rows 32 j + 8 w + r8 hold column cperm(row): period 64, so j even / odd have their own lane base"""
    e.ins(f"s_mov_b32 {STMP}, {t}")
    for j in range(8):
        e.ins(f"v_add_u32 v{BO + j}, %[bob{j & 1}], {STMP}")
        if j & 1 and j < 7:
            e.ins(f"s_add_u32 {STMP}, {STMP}, %[sw]")


def step_body(e, buf, mfma, kind):
    """kind: 'M' steady, 'T3' / 'T2' / 'T1' the last three K-steps of a tile (offset switches), schedule B of gen_wide_loop"""
    seq = ["r"] * 16 + ["w", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "l"]
    fill = {}
    for half in (0, 1):
        nr = nw = nl = 0
        for pidx in range(32):
            slot = 64 * half + 2 * pidx + 1
            k = seq[pidx]
            if k == "r":
                kk, idx = FRAG_ORDER[nr]
                fill.setdefault(slot, []).append(("rd", 1 - half, kk, idx, buf if half == 0 else buf ^ 1))
                nr += 1
            elif k == "w":
                if nw == 0:
                    fill.setdefault(slot, []).append(("covervm", (8 if half == 0 else 0) + 7))
                fill.setdefault(slot, []).append(("wr", (8 if half == 0 else 0) + nw, buf ^ 1 if half == 0 else buf))
                nw += 1
            else:
                fill.setdefault(slot, []).append(("ld", (8 if half == 0 else 0) + nl))
                nl += 1
    fill.setdefault(0, []).append(("cover", 0))
    fill.setdefault(62, []).append(("bar",))
    # K offsets: the W requests of the first half advance KW behind them, the A requests of the second half KA
    fill.setdefault(64, []).append(("advw",))
    fill.setdefault(127, []).append(("adva",))
    if kind == "T3":   # second half requests A(next tile, K-step 0)
        fill.setdefault(64, []).append(("swa",))
    if kind == "T2":   # first half requests W(next tile, K-step 0)
        fill.setdefault(0, []).insert(0, ("sww",))
    for m in range(128):
        for f in fill.get(m, []):
            if f[0] == "rd":
                read_frag(e, f[1], f[2], f[3], f[4])
            elif f[0] == "wr":
                e.need_loaded(f[1])
                write_piece(e, f[1], f[2])
            elif f[0] == "ld":
                load_piece(e, f[1])
            elif f[0] == "bar":
                e.drain_lds()
                e.ins("s_barrier")
            elif f[0] == "advw":
                e.ins(f"s_add_u32 {KW}, {KW}, 128")
            elif f[0] == "adva":
                pass  # (emitted behind the step's last MFMA, below)
            elif f[0] == "cover":
                e.need_written(range(64, 128))
            elif f[0] == "covervm":
                e.need_loaded(f[1])
            elif f[0] == "swa":
                set_a_offsets(e, TAN)
                e.ins(f"s_mov_b32 {KA}, 0")
            elif f[0] == "sww":
                set_w_offsets(e, TWN)
                e.ins(f"s_mov_b32 {KW}, 0")
        s, idx = divmod(m, 64)
        i, j = divmod(idx, 8)
        e.need_written(list(range(frag_w(s, j), frag_w(s, j) + 4)) + list(range(frag_a(s, i), frag_a(s, i) + 4)))
        c = acc(i, j)
        e.ins(f"{mfma} a[{c}:{c + 3}], {vr(frag_w(s, j))}, {vr(frag_a(s, i))}, a[{c}:{c + 3}]")
    e.ins(f"s_add_u32 {KA}, {KA}, 128")


def bias_loads(e, tb, geglu):
    """the 32 bias floats of a lane for tile offset %[tb] -> v160..v191 (8 x dwordx4; a NULL bias = a descriptor of 0 records: zeros).
    lin   : piece column half PJ, block pair jp: 8 adjacent floats at column PJ*64 + jp*32 + 8 fq  -> regs 160 + (PJ*2 + jp)*8
    geglu : piece column half PJ, block j: 4 floats at column PJ*64 + j*16 + 4 fq                   -> regs 160 + (PJ*4 + j)*4"""
    for k in range(8):
        if geglu:
            col = (k >> 2) * 64 + (k & 3) * 16
        else:
            col = (k >> 2) * 64 + ((k >> 1) & 1) * 32 + (k & 1) * 4
        e.vm_load(f"buffer_load_dwordx4 {vr(BIAS + 4 * k)}, %[bvo], %[bdesc], {tb} offen offset:{col * 4}", 100 + k)


def bias_reg(j8, r, geglu):
    """register holding the bias of accumulator block column j8 (0..7), element r"""
    pj, j = j8 >> 2, j8 & 3
    if geglu:
        return BIAS + (pj * 4 + j) * 4 + r
    return BIAS + (pj * 2 + (j >> 1)) * 8 + (j & 1) * 4 + r


def acc_init_rows(e, geglu, rbs):
    """accumulators of row blocks `rbs` <- the NEXT tile's bias (v160..v191)"""
    for k in range(8):
        e.need_loaded(100 + k)
    for i8 in rbs:
        for j8 in range(8):
            for r in range(4):
                e.ins(f"v_accvgpr_write_b32 a{acc(i8, j8) + r}, v{bias_reg(j8, r, geglu)}")


def acc_init(e, geglu):
    acc_init_rows(e, geglu, range(8))
    e.ins("s_nop 4")


GELU_C = ["0xba05bb0c", "0x3bf0996a", "0xbd56e399", "0xbeeb2c8f", "0xbf93565f", "0xbf800009"]  # gelu_q_exp2 c5..c0 (common.hpp), f32 bits
TT, QQ, C5 = 144, 148, 152  # GEGLU epilogue scratch: |g| (4), polynomial (4), the pair (c5, c5); conversion groups rotate over v128..v143


def epilogue(e, geglu, cvt):
    """accumulators -> 16-bit output.  lin: 8 adjacent columns per (row block, block pair): 4 cvt_pk + one 16-byte store;
    geglu: (value) * gelu(gate), 4 columns per pair: packed-f32 arithmetic (two elements per instruction: no MFMA runs beside
    it here), 2 cvt_pk + one 8-byte store.  A row block's accumulators are re-initialised (the next tile's bias) right behind
    its last read, between the stores.  Temporaries rotate over groups of 8 (a store reads its data registers late)."""
    e.ins("s_nop 7")
    e.ins("s_nop 7")
    bias_loads(e, TBN, geglu)
    e.ins(f"s_mov_b32 {SROW}, {TC}")
    if geglu:
        e.ins(f"v_mov_b32 v{C5}, {GELU_C[0]}")
        e.ins(f"v_mov_b32 v{C5 + 1}, {GELU_C[0]}")
    ngrp = 2 if geglu else 3
    grp = 0
    for rb in range(8):            # row block of the wave's 128 rows
        for pj in range(2):
            for jp in range(2):
                t = T0 + 8 * (grp % ngrp)
                grp += 1
                j0 = pj * 4 + 2 * jp
                for r in range(4):
                    e.ins(f"v_accvgpr_read_b32 v{t + r}, a{acc(rb, j0) + r}")
                    e.ins(f"v_accvgpr_read_b32 v{t + 4 + r}, a{acc(rb, j0 + 1) + r}")
                if not geglu:
                    if cvt == "bf16":
                        for k in range(4):
                            e.ins(f"v_cvt_pk_bf16_f32 v{t + k}, v{t + 2 * k}, v{t + 2 * k + 1}")
                    else:
                        for k in range(8):
                            e.ins(f"v_cvt_f16_f32 v{t + k}, v{t + k}")
                        for k in range(4):
                            e.ins(f"v_pack_b32_f16 v{t + k}, v{t + 2 * k}, v{t + 2 * k + 1}")
                    e.vm_store(f"buffer_store_dwordx4 {vr(t)}, %[cvo], %[cdesc], {SROW} offen offset:{(pj * 64 + jp * 32) * 2}")
                else:
                    # value v[t..t+3], gate g = v[t+4..t+7]; T = |g|; gelu(g) = 0.5 (g + T) - T 2^P(T)  (common.hpp gelu_erf_f)
                    g = t + 4
                    for r in range(4):
                        e.ins(f"v_and_b32 v{TT + r}, 0x7fffffff, v{g + r}")
                    for h in (0, 2):
                        e.ins(f"v_pk_fma_f32 {vr(QQ + h, 2)}, {vr(TT + h, 2)}, {vr(C5, 2)}, s[{GC[0][1:]}:{int(GC[0][1:]) + 1}] op_sel_hi:[1,1,0]")
                    for cst in GC[1:]:
                        for h in (0, 2):
                            e.ins(f"v_pk_fma_f32 {vr(QQ + h, 2)}, {vr(QQ + h, 2)}, {vr(TT + h, 2)}, s[{cst[1:]}:{int(cst[1:]) + 1}] op_sel_hi:[1,1,0]")
                    for r in range(4):
                        e.ins(f"v_exp_f32 v{QQ + r}, v{QQ + r}")
                    for h in (0, 2):
                        e.ins(f"v_pk_add_f32 {vr(g + h, 2)}, {vr(g + h, 2)}, {vr(TT + h, 2)}")       # g + |g|
                    for h in (0, 2):
                        e.ins(f"v_pk_mul_f32 {vr(QQ + h, 2)}, {vr(QQ + h, 2)}, {vr(TT + h, 2)}")     # |g| Q(|g|)
                    for h in (0, 2):
                        e.ins(f"v_pk_fma_f32 {vr(g + h, 2)}, {vr(g + h, 2)}, 0.5, {vr(QQ + h, 2)} op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")
                    for h in (0, 2):
                        e.ins(f"v_pk_mul_f32 {vr(t + h, 2)}, {vr(t + h, 2)}, {vr(g + h, 2)}")
                    if cvt == "bf16":
                        e.ins(f"v_cvt_pk_bf16_f32 v{t}, v{t}, v{t + 1}")
                        e.ins(f"v_cvt_pk_bf16_f32 v{t + 1}, v{t + 2}, v{t + 3}")
                    else:
                        for k in range(4):
                            e.ins(f"v_cvt_f16_f32 v{t + k}, v{t + k}")
                        e.ins(f"v_pack_b32_f16 v{t}, v{t}, v{t + 1}")
                        e.ins(f"v_pack_b32_f16 v{t + 1}, v{t + 2}, v{t + 3}")
                    e.vm_store(f"buffer_store_dwordx2 {vr(t, 2)}, %[cvo], %[cdesc], {SROW} offen offset:{(pj * 32 + jp * 16) * 2}")
        acc_init_rows(e, geglu, [rb])
        if rb < 7:
            e.ins(f"s_add_u32 {SROW}, {SROW}, %[rs]")
    e.ins("s_nop 4")


def next_scalars(e):
    """{ta, tw, tc, tb} of tile min(TIDX + 1, ntiles - 1) from this wave's table in LDS -> TAN.. (v128 / v[132:135] as scratch:
    called at a tile boundary, fragment set 1 is free).  One LDS round trip per tile."""
    e.ins(f"s_add_u32 {STMP2}, {TIDX}, 1")
    e.ins(f"s_sub_u32 {STMP}, %[ntl], 1")
    e.ins(f"s_min_u32 {STMP2}, {STMP2}, {STMP}")
    e.ins(f"s_lshl_b32 {STMP2}, {STMP2}, 4")
    e.ins(f"s_add_u32 {STMP2}, {STMP2}, %[tbl]")
    e.ins(f"v_mov_b32 v128, {STMP2}")
    e.lds_op("ds_read_b128 v[132:135], v128", writes=range(132, 136))
    e.drain_lds()
    for k, dst in enumerate((TAN, TWN, TCN, TBN)):
        e.ins(f"v_readfirstlane_b32 {dst}, v{132 + k}")


def tile_start(e):
    next_scalars(e)
    e.ins(f"s_mov_b32 {KW}, 256")
    e.ins(f"s_mov_b32 {KA}, 384")
    e.ins(f"s_sub_u32 {NCNT}, %[nk], 3")


def prologue(e, geglu):
    """the workgroup's first tile: offsets, K-steps 0 (whole) and 1 (A half) staged, A(2) / W(1) requested, bias into the accumulators"""
    if geglu:
        for cst, sg in zip(GELU_C[1:], GC):
            e.ins(f"s_mov_b32 {sg}, {cst}")
            e.ins(f"s_mov_b32 s{int(sg[1:]) + 1}, {cst}")
    e.ins("v_mov_b32 v128, %[tbl]")
    e.lds_op("ds_read_b128 v[132:135], v128", writes=range(132, 136))
    e.drain_lds()
    for k, dst in enumerate((TA, TW, TC, TB)):
        e.ins(f"v_readfirstlane_b32 {dst}, v{132 + k}")
    e.ins(f"s_mov_b32 {TIDX}, 0")
    e.ins(f"s_mov_b32 {PAR}, 0")
    set_a_offsets(e, TA)
    set_w_offsets(e, TW)
    e.ins(f"s_mov_b32 {KA}, 0")
    e.ins(f"s_mov_b32 {KW}, 0")
    for q in range(NQ):
        load_piece(e, q)                                          # K-step 0 -> P
    e.ins(f"s_mov_b32 {KA}, 128")
    for q in range(8):                                            # K-step 1, A pieces -> fragment set 1's registers v128..v159
        e.vm_load(f"buffer_load_dwordx4 {vr(128 + 4 * q)}, v{AO + q}, %[adesc], {KA} offen", 16 + q)
    bias_loads(e, TB, geglu)
    for q in range(NQ):
        e.need_loaded(q)
        write_piece(e, q, 0)
    e.ins(f"s_mov_b32 {KW}, 128")
    for q in range(8, NQ):
        load_piece(e, q)                                          # K-step 1, W pieces -> P[8..15]
    for q in range(8):
        e.need_loaded(16 + q)
        e.lds_op(f"ds_write_b128 %[lwa], {vr(128 + 4 * q)} offset:{ABUF + q * 4096}", reads=range(128 + 4 * q, 132 + 4 * q))
    e.ins(f"s_mov_b32 {KA}, 256")
    for q in range(8):
        load_piece(e, q)                                          # K-step 2, A pieces -> P[0..7]
    acc_init(e, geglu)
    e.drain_lds()
    e.ins("s_barrier")
    for kind, idx in FRAG_ORDER:
        read_frag(e, 0, kind, idx, 0)
    e.drain_lds()


def build(mfma, geglu, cvt):
    e = E2()
    e.in_loop = False
    prologue(e, geglu)
    tile_start(e)
    pro = e.out

    def body(buf, kind):
        e.out = []
        step_body(e, buf, mfma, kind)
        return e.out

    prev = None
    for _ in range(3):  # steady state to a fixed point
        m0, m1 = body(0, "M"), body(1, "M")
        if prev == (m0, m1):
            break
        prev = (m0, m1)
    texts = {"M0": m0, "M1": m1}
    for p in (0, 1):  # a tile ending at parity p, then the next tile's first step at parity p ^ 1
        texts[f"T3_{p}"], texts[f"T2_{p ^ 1}"], texts[f"T1_{p}"] = body(p, "T3"), body(p ^ 1, "T2"), body(p, "T1")
        e.out = []
        epilogue(e, geglu, cvt)
        texts.setdefault("EPI", e.out)
        assert texts["EPI"] == e.out
        e.out = []
        tile_start(e)
        texts.setdefault("START", e.out)
        assert texts["START"] == e.out
        texts[f"F{p ^ 1}"] = body(p ^ 1, "M")
        a, b = body(p, "M"), body(p ^ 1, "M")  # behind F the steady bodies must fit again
        assert (a, b) == ((m0, m1) if p == 0 else (m1, m0)), "steady state not re-entered behind F"
    L = lambda name: f"LWS_{name}_%="
    lines = list(pro)
    lines.append(f"s_branch {L('MAIN0')}")          # the workgroup's first tile: parity 0, no stores in flight -> steady bodies
    lines.append(L("TILE") + ":")
    lines += texts["START"]
    lines += [f"s_cmp_eq_u32 {PAR}, 0", f"s_cbranch_scc0 {L('F1')}"]
    for p in (0, 1):
        lines.append(L(f"F{p}") + ":")
        lines += texts[f"F{p}"]
        lines += [f"s_sub_u32 {NCNT}, {NCNT}, 1", f"s_cmp_eq_u32 {NCNT}, 0", f"s_cbranch_scc1 {L(f'TAIL{p ^ 1}')}", f"s_branch {L(f'MAIN{p ^ 1}')}"]
    lines.append(L("MAIN0") + ":")
    lines += texts["M0"]
    lines += [f"s_sub_u32 {NCNT}, {NCNT}, 1", f"s_cmp_eq_u32 {NCNT}, 0", f"s_cbranch_scc1 {L('TAIL1')}"]
    lines.append(L("MAIN1") + ":")
    lines += texts["M1"]
    lines += [f"s_sub_u32 {NCNT}, {NCNT}, 1", f"s_cmp_eq_u32 {NCNT}, 0", f"s_cbranch_scc1 {L('TAIL0')}", f"s_branch {L('MAIN0')}"]
    for p in (0, 1):
        lines.append(L(f"TAIL{p}") + ":")
        lines += texts[f"T3_{p}"] + texts[f"T2_{p ^ 1}"] + texts[f"T1_{p}"]
        lines.append(f"s_branch {L('EPI')}")
    lines.append(L("EPI") + ":")
    lines += texts["EPI"]
    # next tile: parity continues with the K-step count; the scalars of the tile just prefetched become the current ones
    lines += [f"s_and_b32 {STMP}, %[nk], 1", f"s_xor_b32 {PAR}, {PAR}, {STMP}", f"s_add_u32 {TIDX}, {TIDX}, 1",
              f"s_mov_b32 {TA}, {TAN}", f"s_mov_b32 {TW}, {TWN}", f"s_mov_b32 {TC}, {TCN}", f"s_mov_b32 {TB}, {TBN}",
              f"s_cmp_lt_u32 {TIDX}, %[ntl]", f"s_cbranch_scc1 {L('TILE')}"]
    lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return lines


def clobbers():
    v = ",".join(f'"v{r}"' for r in range(48, 256))
    a = ",".join(f'"a{r}"' for r in range(256))
    s = ",".join(f'"s{r}"' for r in range(70, 100))
    return f"#define PM_WSTREAM_CLOBBERS {v},{a},{s},\"memory\",\"scc\"\n"


if __name__ == "__main__":
    print("// GENERATED by tools/gen_wide_stream.py - do not edit")
    print(clobbers())
    for tag, mf, cvt in (("BF16", "v_mfma_f32_16x16x32_bf16", "bf16"), ("F16", "v_mfma_f32_16x16x32_f16", "f16")):
        for fl, geglu in (("LIN", False), ("GEGLU", True)):
            print(c_string(f"PM_WSTREAM_{fl}_{tag}", build(mf, geglu, cvt)))
            print()

"""Generator of the hand-scheduled main loop of gemm_wide_kernel (csrc/gemm_wide.hip): 256 x 256 output tile, FOUR waves (one
per SIMD) of 128 x 128 each - 8 x 8 blocks of v_mfma_f32_16x16x32 = 256 accumulator registers a[0:255] per lane - fed the way
the vendor library's 256x256 kernels are (profiles/r04/hipblaslt_kernels.txt: MIWT8_8, PGR): every wave stages its share of the
A and W K-tiles global -> VGPR (buffer_load_dwordx4, one K-step ahead in 64 registers) -> LDS (ds_write_b128 into the other
of two 64-KiB buffers) and reads fragments with ds_read_b128, double-buffered by k-substep.  One s_barrier per K-step.

    python tools/gen_wide_loop.py > open-pandora_amd/csrc/gemm_wide_loop.inc

The text is ONE inline-asm statement body per (dtype, A mode); hipcc cannot hold this tile (profiles/r03/
negative_result_gemm256w_4waves_128x128_agpr.txt: with all 256 AGPRs taken its accumulators are copied, not updated in place).
Fixed registers (clobbered by the statement): v64-v127 fragment set 0, v128-v191 fragment set 1 (each: W blocks j at +4j, A blocks
i at +32+4i), v192-v255 the staging set P (A pieces q = 0..7, W pieces q = 8..15), a0-a255 the accumulators (block (i, j) at
4 (8 i + j)).  v0-v63 stay the compiler's (operands).  LDS: A buffers at 0 / 32 KiB, W buffers at 64 / 96 KiB; rows of 128 bytes,
16-byte chunk c of row r stored at chunk c ^ (r & 7) (the layout of gemm.hip's loaders: conflict-free ds_write_b128 by 8-lane
rows and ds_read_b128 of the 16x16x32 fragments).

Waits are placed by a scoreboard: LDS operations retire in order (lgkmcnt, 4 bits), vector-memory loads in order (vmcnt)."""

NQ = 16          # staging pieces per wave and K-step (8 A + 8 W), 1 KiB each
ABUF, WBUF = 32768, 32768
# timing-only ablations (diagnostics variants; results are wrong by construction): drop the global requests / the LDS
# staging writes / the barrier / the fragment reads inside the loop; "consume": wait for a staging write to retire before its
# registers are requested again (the safe form; 0 = trust the in-order issue of ds_write's source read)
OPT = dict(load=1, write=1, bar=1, read=1, consume=1, stamp=0, vmwait=1, lgkwait=1)


class Emit:
    def __init__(self):
        self.out = []
        self.lds = []      # issue-ordered LDS ops since the last lgkmcnt(0): (set of vgprs written, set of vgprs read)
        self.lds_done = 0  # how many of them are known complete
        self.vm = []       # issue-ordered vector-memory loads: the staging piece each one fills
        self.vm_done = 0
        self.in_loop = False  # (timing-only ablations drop waits inside the loop, never in the prologue)

    def ins(self, s):
        self.out.append(s)
        if len(self.vm) > 128:  # (only the tail matters: older requests are complete by construction)
            drop = len(self.vm) - 64
            self.vm, self.vm_done = self.vm[drop:], max(0, self.vm_done - drop)

    def lds_op(self, text, writes=(), reads=()):
        self.ins(text)
        self.lds.append((set(writes), set(reads)))

    def _wait_idx(self, idx):
        """make LDS op number idx (0-based in self.lds) complete"""
        if idx < self.lds_done:
            return
        n_after = len(self.lds) - 1 - idx
        n = min(n_after, 15)
        if OPT["lgkwait"] or not self.in_loop:
            self.ins(f"s_waitcnt lgkmcnt({n})")
        self.lds_done = len(self.lds) - n

    def need_written(self, regs):
        """before reading `regs`: the youngest LDS read that writes any of them must have returned"""
        regs = set(regs)
        for idx in range(len(self.lds) - 1, -1, -1):
            if self.lds[idx][0] & regs:
                self._wait_idx(idx)
                return

    def need_consumed(self, regs):
        """before overwriting `regs`: the youngest ds_write that sources any of them must have left"""
        regs = set(regs)
        for idx in range(len(self.lds) - 1, -1, -1):
            if self.lds[idx][1] & regs:
                self._wait_idx(idx)
                return

    def vm_load(self, text, q):
        self.ins(text)
        self.vm.append(q)

    def need_loaded(self, q):
        """before ds_write sources P[q]: its youngest request must have landed (loads return in order)"""
        for idx in range(len(self.vm) - 1, -1, -1):
            if self.vm[idx] == q:
                if idx >= self.vm_done:
                    n = len(self.vm) - 1 - idx
                    assert n < 64
                    if OPT["vmwait"] or not self.in_loop:
                        self.ins(f"s_waitcnt vmcnt({n})")
                    self.vm_done = idx + 1
                return
        raise AssertionError("piece never requested")

    def drain_vm(self):
        self.ins("s_waitcnt vmcnt(0)")
        self.vm, self.vm_done = [], 0

    def drain_lds(self):
        self.ins("s_waitcnt lgkmcnt(0)")
        self.lds, self.lds_done = [], 0


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def frag_w(s, j):
    return 64 + 64 * s + 4 * j


def frag_a(s, i):
    return 64 + 64 * s + 32 + 4 * i


def preg(q):
    return 192 + 4 * q


def acc(i, j):
    return 4 * (8 * i + j)


def load_piece(e, q, amode):
    """global -> P[q] for the K-step whose byte offset is in %[kld] (dense) / whose descriptors are current (conv)"""
    if OPT["consume"]:
        e.need_consumed(range(preg(q), preg(q) + 4))
    if q < 8:
        off = f"%[ao{q}]"
        if amode != "dense":  # conv: padding rows get offset ~0 -> out of range -> zeros (tools/probes/buffer_lds_oob.hip)
            e.ins(f"v_bfe_i32 %[tmp], %[inv{q}], %[tap], 1")
            e.ins(f"v_or_b32 %[tmp], %[tmp], %[ao{q}]")
            off = "%[tmp]"
        e.vm_load(f"buffer_load_dwordx4 {vr(preg(q))}, {off}, %[adesc], %[kld] offen", q)
    else:
        e.vm_load(f"buffer_load_dwordx4 {vr(preg(q))}, %[bo{q - 8}], %[wdesc], %[kld] offen", q)


def write_piece(e, q, buf):
    if q < 8:
        e.lds_op(f"ds_write_b128 %[lwa], {vr(preg(q))} offset:{buf * ABUF + q * 4096}", reads=range(preg(q), preg(q) + 4))
    else:
        e.lds_op(f"ds_write_b128 %[lww], {vr(preg(q))} offset:{buf * WBUF + (q - 8) * 4096}", reads=range(preg(q), preg(q) + 4))


def read_frag(e, s, kind, idx, buf):
    if kind == "w":
        r = frag_w(s, idx)
        e.lds_op(f"ds_read_b128 {vr(r)}, %[lrw{s}] offset:{buf * WBUF + idx * 2048}", writes=range(r, r + 4))
    else:
        r = frag_a(s, idx)
        e.lds_op(f"ds_read_b128 {vr(r)}, %[lra{s}] offset:{buf * ABUF + idx * 2048}", writes=range(r, r + 4))


FRAG_ORDER = [("w", j) for j in range(8)] + [("a", i) for i in range(8)]


def tmp_reg(q):
    return 128 + 4 * q  # prologue only: the A pieces of K-step 1 wait in fragment set 1's registers


def step_body(e, buf, mfma, amode, sched):
    """One K-step on LDS buffer `buf`: 128 MFMAs, one filler (fragment read / staging write / global request) behind every
    second MFMA: a 16-cycle v_mfma_f32_16x16x32 holds the SIMD's issue for 8 cycles, so one 13-24-cycle filler per MFMA
    stretches every gap (the first schedule: 32 reads cost 17 %, 16 writes 10 %, 16 requests 14 % of the MFMA-only loop,
    profiles/r06/wide_loop_ablation.txt) while one per TWO MFMAs fits.
    first half : fragment set 1 (this buffer, k-substep 1);  W pieces of K-step +1 -> other buffer, re-requested for +2;  barrier
    second half: fragment set 0 of K-step +1 (other buffer);  A pieces of K-step +2 -> THIS buffer (free behind the barrier),
                 re-requested for +3.
    `sched` = (pattern of the four filler positions of a group, barrier slot)"""
    pattern, bar_at = sched
    fill = {}
    cover = pattern in ("B", "C", "D")
    if pattern == "B":
        # reads first, then the write / request pairs (a request two fillers behind its write): ONE lgkmcnt wait covers a
        # fragment set (the waits are instructions too: ~4 issue cycles each, 48 of them per K-step in the interleaved form)
        seqs = [["r"] * 16 + ["w", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "w", "l", "l"]] * 2
    elif pattern in ("C", "D"):
        # every LDS operation of the first half well ahead of the barrier's lgkmcnt(0) (the requests, which are not LDS
        # operations, fill the slots in front of it): a drain right behind a ds_write waits out the write's latency
        seqs = [["r", "r", "w"] * 8 + ["l"] * 8, ["r"] * 16 + ["w"] * 8 + ["l"] * 8]
    else:
        seqs = [[pattern[pidx % len(pattern)] for pidx in range(32)]] * 2
    for half in (0, 1):
        nr = nw = nl = 0
        for pidx in range(32):
            slot = 64 * half + 2 * pidx + 1
            kind = seqs[half][pidx]
            if kind == "r":
                k, idx = FRAG_ORDER[nr]
                fill.setdefault(slot, []).append(("rd", 1 - half, k, idx, buf if half == 0 else buf ^ 1))
                nr += 1
            elif kind == "w":
                if nw == 0 and cover:
                    fill.setdefault(slot, []).append(("covervm", (8 if half == 0 else 0) + 7))
                fill.setdefault(slot, []).append(("wr", (8 if half == 0 else 0) + nw, buf ^ 1 if half == 0 else buf))
                nw += 1
            elif kind == "l":
                fill.setdefault(slot, []).append(("ld", (8 if half == 0 else 0) + nl))
                nl += 1
        assert (nr, nw, nl) == (16, 8, 8), (nr, nw, nl)
    if cover and pattern != "D":
        fill.setdefault(0, []).append(("cover", 0))
    fill.setdefault(bar_at, []).append(("bar",))
    fill.setdefault(64, []).append(("adv",))
    for m in range(128):
        for f in fill.get(m, []):
            if f[0] == "rd":
                if OPT["read"]:
                    read_frag(e, f[1], f[2], f[3], f[4])
            elif f[0] == "wr":
                if OPT["write"]:
                    if OPT["load"]:
                        e.need_loaded(f[1])
                    write_piece(e, f[1], f[2])
            elif f[0] == "ld":
                if OPT["load"]:
                    load_piece(e, f[1], amode)
            elif f[0] == "bar":
                e.drain_lds()
                if OPT["bar"]:
                    e.ins("s_barrier")
            elif f[0] == "adv":
                advance_k(e, amode)
            elif f[0] == "cover":
                if OPT["read"]:
                    e.need_written(range(64 + 64 * f[1], 128 + 64 * f[1]))
            elif f[0] == "covervm":
                if OPT["load"] and OPT["write"]:
                    e.need_loaded(f[1])
        s, idx = divmod(m, 64)
        i, j = divmod(idx, 8)
        e.need_written(list(range(frag_w(s, j), frag_w(s, j) + 4)) + list(range(frag_a(s, i), frag_a(s, i) + 4)))
        c = acc(i, j)
        e.ins(f"{mfma} a[{c}:{c + 3}], {vr(frag_w(s, j))}, {vr(frag_a(s, i))}, a[{c}:{c + 3}]")


def advance_k(e, amode):
    """K offset of the next requests (clamped to the last K-step: the requests past the end re-read it, harmlessly)"""
    if amode == "dense":
        e.ins("s_add_u32 %[kld], %[kld], 128")
        e.ins("s_min_u32 %[kld], %[kld], %[kmax]")
    else:
        raise NotImplementedError(amode)


def kernel_text(mfma, amode, sched):
    e = Emit()
    # ---- prologue.  State the loop expects at K-step 0: buffer 0 whole; the A half of buffer 1 (K-step 1); P[0..7] requested
    # for K-step 2 (A), P[8..15] for K-step 1 (W); fragment set 0 read; %[kld] = offset of K-step 2.
    for q in range(NQ):
        load_piece(e, q, amode)                                      # K-step 0 -> P
    advance_k(e, amode)
    for q in range(8):                                               # K-step 1, A pieces -> fragment set 1's registers
        e.vm_load(f"buffer_load_dwordx4 {vr(tmp_reg(q))}, %[ao{q}], %[adesc], %[kld] offen", 16 + q)
    for r in range(256):
        e.ins(f"v_accvgpr_write_b32 a{r}, 0")
    for q in range(NQ):
        e.need_loaded(q)
        write_piece(e, q, 0)
    for q in range(8, NQ):
        load_piece(e, q, amode)                                      # K-step 1, W pieces -> P[8..15]
    for q in range(8):
        e.need_loaded(16 + q)
        e.lds_op(f"ds_write_b128 %[lwa], {vr(tmp_reg(q))} offset:{ABUF + q * 4096}", reads=range(tmp_reg(q), tmp_reg(q) + 4))
    advance_k(e, amode)
    for q in range(8):
        load_piece(e, q, amode)                                      # K-step 2, A pieces -> P[0..7]
    e.drain_lds()
    e.ins("s_barrier")
    for kind, idx in FRAG_ORDER:
        read_frag(e, 0, kind, idx, 0)
    e.drain_lds()
    if OPT["stamp"]:  # diagnostics: shader clock at the loop's entry (tools/wide_probe.py --stamps)
        e.ins("s_memtime %[tpro]")
        e.ins("s_waitcnt lgkmcnt(0)")
    pro = e.out
    e.in_loop = True
    # ---- steady state: simulate until the text of a (buffer 0, buffer 1) pair repeats
    bodies = []
    for it in range(3):
        pair = []
        for buf in (0, 1):
            e.out = []
            step_body(e, buf, mfma, amode, sched)
            pair.append(e.out)
        bodies.append(pair)
    assert bodies[1] == bodies[2], "scoreboard did not reach a fixed point"
    # the first trip runs the steady-state text too: its waits assume MORE operations in flight than the prologue left (safe)
    b0, b1 = bodies[2]
    lines = list(pro)
    lines.append("1:")
    lines += b0
    lines.append("s_sub_u32 %[nk], %[nk], 1")
    lines.append("s_cmp_eq_u32 %[nk], 0")
    lines.append("s_cbranch_scc1 2f")
    lines += b1
    lines.append("s_sub_u32 %[nk], %[nk], 1")
    lines.append("s_cmp_lg_u32 %[nk], 0")
    lines.append("s_cbranch_scc1 1b")
    lines.append("2:")
    # drain: every request past the end has landed in P (nothing reads it), the MFMA results are readable by v_accvgpr_read
    lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    lines.append("s_nop 7")
    lines.append("s_nop 7")
    return lines


def c_string(name, lines):
    out = [f"#define {name} \\"]
    for ln in lines:
        out.append(f'  "{ln}\\n\\t" \\')
    out.append('  ""')
    return "\n".join(out)


def clobbers():
    v = ",".join(f'"v{r}"' for r in range(64, 256))
    a = ",".join(f'"a{r}"' for r in range(256))
    return f"#define PM_WIDE_CLOBBERS {v},{a},\"memory\",\"scc\"\n"


VARIANTS = [  # (tag, schedule, options); V0 ships, the others exist in the diagnostics build only (tools/wide_probe.py --variants)
    ("V0", ("B", 62), dict(consume=0)),
    ("V1", ("B", 62), dict(consume=1)),
    ("V3", ("B", 62), dict(consume=0, load=0)),
    ("V4", ("B", 62), dict(consume=0, load=0, write=0)),
    ("V6", ("B", 62), dict(consume=0, load=0, write=0, bar=0, read=0)),
    ("V7", ("B", 62), dict(consume=0, stamp=1)),
]

if __name__ == "__main__":
    # schedule = (set-1 reads from MFMA slot, first ds_write slot, slots between writes, request lag behind its write, barrier
    #             slot, set-0 reads of the next step from slot 64 + ..)
    print("// GENERATED by tools/gen_wide_loop.py - do not edit")
    print(clobbers())
    base = dict(OPT)
    for tag, mf in (("BF16", "v_mfma_f32_16x16x32_bf16"), ("F16", "v_mfma_f32_16x16x32_f16")):
        for vt, sched, opt in VARIANTS:
            if vt != "V0" and tag != "BF16":
                continue
            OPT.update(base)
            OPT.update(opt)
            print(c_string(f"PM_WIDE_LOOP_DENSE_{tag}_{vt}", kernel_text(mf, "dense", sched)))
            print()
    OPT.update(base)

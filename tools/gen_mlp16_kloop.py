#!/usr/bin/env python
"""Generator of the hand-scheduled k-loops of the large-row 16-bit MLP kernels (gaussianprediction_amd/csrc/deform_mlp16.hip).

    python tools/gen_mlp16_kloop.py            # rewrites gaussianprediction_amd/csrc/deform_mlp16_kloop.inc

One layer's product (K = 256: sixteen k-steps of v_mfma_f32_32x32x16) is ONE inline-asm statement with a fixed register
plan, because hipcc's version of the loop (a) waits for every LDS fragment right in front of the MFMA that reads it,
(b) prefetches the weight fragments one k-step deep behind `s_waitcnt vmcnt(0)`, and (c) has no registers left to carry
the saved-tensor stores of the tile the product is reading (round 4 / 5: the C++ attempts spilled inside the loop).

What the statement does per k-step (split mode, 64 rows x 64 features per wave, accumulators = operands):
  * 12 MFMAs:  am += Wh . Xh,  ax += Wl' . Xh,  ax += Wh . Xl'
  * the weight fragments of k-step ks + D (D = prefetch depth) requested from L2 (4 x 1 KB per wave),
  * the activation fragments of k-step ks + 1 read from LDS (4 x ds_read_b128),
  * CARRY: one 1 KB store of the tile being read (the saved tensor of the previous layer, blocked [16 rows][feature][16]):
    two ds_read_b64_tr_b16 (the LDS transpose read) gather 8 rows of one feature per lane, one global_store_dwordx4 writes them.
The waits are COUNTED: the generator keeps the two in-order queues (vmcnt: loads and stores together; lgkmcnt: LDS) and
emits, in front of every instruction, the largest count that still guarantees its operands -- never 0 inside the loop.

Register plan (clobbered, fixed): see `Plan`.  Everything else (accumulators, addresses) are asm operands.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gaussianprediction_amd", "csrc", "deform_mlp16_kloop.inc")


class Queue:
    """An in-order completion counter (vmcnt or lgkmcnt): tags of the operations still outstanding, oldest first."""

    def __init__(self, name, limit):
        self.name, self.limit, self.q = name, limit, []

    def issue(self, tag):
        self.q.append(tag)
        assert len(self.q) <= self.limit, (self.name, len(self.q))

    def wait_for(self, tag):
        """Return the count to wait for so that every operation tagged `tag` has completed (None: already complete)."""
        idx = [i for i, t in enumerate(self.q) if t == tag]
        if not idx:
            return None
        n = len(self.q) - 1 - idx[-1]
        self.q = self.q[idx[-1] + 1:]
        return n

    def wait_all(self):
        if not self.q:
            return None
        self.q = []
        return 0


class Emitter:
    def __init__(self):
        self.lines = []
        self.vm = Queue("vmcnt", 63)
        self.lgkm = Queue("lgkmcnt", 15)

    def ins(self, s):
        self.lines.append(s)

    def need_vm(self, tag):
        n = self.vm.wait_for(tag)
        if n is not None:
            self.ins(f"s_waitcnt vmcnt({n})")

    def need_lgkm(self, tag):
        n = self.lgkm.wait_for(tag)
        if n is not None:
            self.ins(f"s_waitcnt lgkmcnt({n})")

    def text(self):
        return "".join(f'    "{l}\\n\\t"\n' for l in self.lines)


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


class Plan:
    """Fixed VGPRs of one variant.  `first` is the lowest clobbered register."""

    def __init__(self, first, depth, n_b, n_a, carry):
        self.depth = depth
        self.n_b, self.n_a = n_b, n_a
        r = first
        self.B = r; r += 4 * n_b * (depth + 1)          # ring of depth + 1 weight-fragment groups
        self.A = r; r += 4 * n_a * 2                    # two activation-fragment groups
        self.ST = r; r += 4                             # the carried store's 16 bytes
        self.va = r; r += 1                             # LDS address of the activation fragments
        self.vw = r; r += 1                             # byte offset of the weight fragments / of the store
        self.t0 = r; r += 1                             # LDS addresses of the gather
        self.t1 = r; r += 1
        self.end = r
        self.first = first
        assert r <= 256, r

    def b(self, ks, q):
        return vr(self.B + 4 * (self.n_b * (ks % (self.depth + 1)) + q), 4)

    def a(self, ks, q):
        return vr(self.A + 4 * (self.n_a * (ks % 2) + q), 4)

    def clobbers(self):
        return ", ".join(f'"v{i}"' for i in range(self.first, self.end))


def gen_split_path(e, p, nks, carry, init, depth):
    """One product (nks k-steps) of the split-fp16 form into the emitter `e`.  init = "bias": am00 / am01 hold the bias on entry
    (every row tile starts from the same 32 features' biases: the first MFMA of am10 / am11 reads am00 / am01 as its addend BEFORE
    these are overwritten, so no accumulator is ever initialised by VALU moves); init = "zero": nothing is read on entry.
    The cross-term accumulators start from the inline constant 0."""
    KST = 8192                                        # bytes per k-step of the fragment-packed weights (8 feature tiles x 1 KB)
    A_OFF = [0, 512, 32768, 33280]                    # ah rt0, al' rt0, ah rt1, al' rt1

    def load_b(ks):
        if ks == 0:
            off = "%[voff]"
        else:
            e.ins(f"v_add_u32 {vr(p.vw)}, {ks * KST}, %[voff]")
            off = vr(p.vw)
        for q in range(4):
            base = "%[swh]" if q < 2 else "%[swl]"
            e.ins(f"global_load_dwordx4 {p.b(ks, q)}, {off}, {base}" + (" offset:1024" if q & 1 else ""))
            e.vm.issue(("B", ks))

    def read_a(ks, qs):
        if qs[0] == 0:
            if ks == 0:
                e.ins(f"v_mov_b32 {vr(p.va)}, %[abase]")
            else:
                e.ins(f"v_xor_b32 {vr(p.va)}, {ks * 32}, %[abase]")
        for q in qs:
            e.ins(f"ds_read_b128 {p.a(ks, q)}, {vr(p.va)}" + (f" offset:{A_OFF[q]}" if A_OFF[q] else ""))
            e.lgkm.issue(("A", ks, q))

    def gather(g, ps):
        # ds_read_b64_tr_b16 = every lane reads 8 bytes at ITS address, then lane l of a 16-lane group receives element (l & 3) of lanes
        # (l >> 2), 4 + (l >> 2), 8 + (l >> 2), 12 + (l >> 2) (measured: tools/probe/ds_tr_probe.hip).  Lane 4 j + q of a group
        # reads features 4 q .. 4 q + 3 of row j, so lane l receives rows 0 .. 3 of feature l: the 4 x 16 transpose in one
        # instruction; two passes (rows 0-3, 4-7) make the 16 bytes of the blocked layout.  (The d16 sub-dword loads cannot build it:
        # with SRAM-ECC a d16 load zeroes the other half of the register.)
        t = vr(p.t0 if ps == 0 else p.t1)
        lit = 64 * (g ^ ps)
        if lit:
            e.ins(f"v_xor_b32 {t}, {lit}, %[sbt]")
            src = t
        else:
            src = "%[sbt]"
        e.ins(f"ds_read_b64_tr_b16 {vr(p.ST + 2 * ps, 2)}, {src}" + (" offset:4096" if ps else ""))
        e.lgkm.issue(("G", g))

    def store(g):
        e.need_lgkm(("G", g))
        if g == 0:
            off = "%[vost]"
        else:
            e.ins(f"v_add_u32 {vr(p.vw)}, {g * 1024}, %[vost]")
            off = vr(p.vw)
        e.ins(f"global_store_dwordx4 {off}, {vr(p.ST, 4)}, %[sst]")
        e.vm.issue(("S", g))

    def mfma(acc, ks, bq, aq, src=None):
        e.need_vm(("B", ks))
        e.need_lgkm(("A", ks, aq))
        e.ins(f"v_mfma_f32_32x32x16_f16 %[{acc}], {p.b(ks, bq)}, {p.a(ks, aq)}, {src if src is not None else '%[' + acc + ']'}")

    # prologue: the first `depth` weight groups and the first activation group
    for ks in range(min(depth, nks)):
        load_b(ks)
    read_a(0, [0, 1, 2, 3])
    for ks in range(nks):
        # (acc, weight fragment, activation fragment, addend): hi x hi first, then the two cross terms
        seq = [("am00", 0, 0, None), ("am01", 1, 0, None), ("am10", 0, 2, None), ("am11", 1, 2, None),
               ("ax00", 2, 0, None), ("ax01", 3, 0, None), ("ax10", 2, 2, None), ("ax11", 3, 2, None),
               ("ax00", 0, 1, None), ("ax01", 1, 1, None), ("ax10", 0, 3, None), ("ax11", 1, 3, None)]
        if ks == 0:
            z = "0"
            if init == "bias":      # row tile 1 first: it reads the biases out of am00 / am01
                seq[0:4] = [("am10", 0, 2, "%[am00]"), ("am11", 1, 2, "%[am01]"), ("am00", 0, 0, None), ("am01", 1, 0, None)]
            else:
                seq[0:4] = [(a, b, c, z) for a, b, c, _ in seq[0:4]]
            seq[4:8] = [(a, b, c, z) for a, b, c, _ in seq[4:8]]
        for m, (acc, bq, aq, src) in enumerate(seq):
            mfma(acc, ks, bq, aq, src)
            # the slots behind the MFMAs (the matrix pipe is busy for 32 cycles per MFMA: anything here issues for free)
            if m == 0 and ks + depth < nks:
                load_b(ks + depth)
            if carry and m in (1, 2):
                gather(ks, m - 1)
            if m == 5 and ks + 1 < nks:
                read_a(ks + 1, [0, 1])
            if m == 6 and ks + 1 < nks:
                read_a(ks + 1, [2, 3])
            if carry and m == 9:
                store(ks)
    assert not e.lgkm.q, e.lgkm.q
    e.vm.q = []                 # (stores may still be in flight at the end of a path: nothing below depends on the count)


def gen_split(paths, init, depth=3):
    """One asm statement = the split-fp16 product of a 64-row workgroup (RT = 2) with one or two PATHS selected by the scalar operand
    `sel` (0 = the first path): the K = 256 layers, and the layer with its own k-step count (forward: layer 0, K = in_pad; data
    backward: the output layer, K = 16).  Both paths use the same operands and registers, so the layer loop around the statement is
    one rolled loop with ONE statement in it -- with several statements (or the compiler's loop beside it) hipcc assigned the 128
    accumulator registers differently per path and moved them between the paths (240 v_mov per layer).
    paths = [(nks, carry), ...].  Operands: am00 am01 am10 am11 ax00 ax01 ax10 ax11 (f32x16), abase (LDS byte address of this lane's row,
    swizzle folded in), voff (lane * 16), swh / swl (SGPR pairs: this wave's hi / lo' weight fragments), sel, and for a carrying path sbt
    (LDS byte address this lane supplies to the transpose reads), vost (this lane's byte offset in a 1 KB chunk of the saved tensor) and
    sst (SGPR pair: this wave's 16-row block of the saved tensor)."""
    p = Plan(256 - (16 * (depth + 1) + 32 + 8), depth, 4, 4, any(c for _, c in paths))
    e = Emitter()
    if len(paths) > 1:
        e.ins("s_cmp_eq_u32 %[sel], 0")
        e.ins("s_cbranch_scc0 71f")
    gen_split_path(e, p, paths[0][0], paths[0][1], init, depth)
    if len(paths) > 1:
        e.ins("s_branch 72f")
        e.lines.append("71:")
        gen_split_path(e, p, paths[1][0], paths[1][1], init, depth)
        e.lines.append("72:")
    # the accumulators are read by VALU code right behind this statement: XDL write -> VALU read needs passes + 3 wait states
    e.ins("s_nop 15")
    return e.text(), p


def gen_plain_path(e, p, nks, carry, init, depth, mn):
    """One product (nks k-steps) of the plain 16-bit form (fp16 / bf16 operands, `mn` = the MFMA mnemonic) of a 128-row workgroup: RT = 4
    row tiles x 2 feature tiles per wave = 8 accumulators c<rt><nt>, 8 MFMAs per k-step from TWO weight fragments (1 KB each) and FOUR
    activation fragments.  Carried store: wave w owns row blocks 2 w and 2 w + 1; k-step g writes features 32 (g & 7) .. + 31 of row
    block 2 w + (g >> 3) -- the byte offset in the wave's part of the saved tensor is again g KB.
    init = "bias": c00 / c01 hold the biases on entry and row tiles 1 .. 3 take them as the addend of their first MFMA."""
    KST = 8192
    A_OFF = [0, 16384, 32768, 49152]

    def load_b(ks):
        if ks == 0:
            off = "%[voff]"
        else:
            e.ins(f"v_add_u32 {vr(p.vw)}, {ks * KST}, %[voff]")
            off = vr(p.vw)
        for q in range(2):
            e.ins(f"global_load_dwordx4 {p.b(ks, q)}, {off}, %[swh]" + (" offset:1024" if q else ""))
            e.vm.issue(("B", ks))

    def read_a(ks, qs):
        if qs[0] == 0:
            if ks == 0:
                e.ins(f"v_mov_b32 {vr(p.va)}, %[abase]")
            else:
                e.ins(f"v_xor_b32 {vr(p.va)}, {ks * 32}, %[abase]")
        for q in qs:
            e.ins(f"ds_read_b128 {p.a(ks, q)}, {vr(p.va)}" + (f" offset:{A_OFF[q]}" if A_OFF[q] else ""))
            e.lgkm.issue(("A", ks, q))

    def gather(g, ps):                      # (see gen_split_path: the LDS transpose read; rows are 512 bytes here)
        t = vr(p.t0 if ps == 0 else p.t1)
        lit = 64 * ((g & 7) ^ ps)
        if lit:
            e.ins(f"v_xor_b32 {t}, {lit}, %[sbt]")
            src = t
        else:
            src = "%[sbt]"
        off = 2048 * ps + 8192 * (g >> 3)
        e.ins(f"ds_read_b64_tr_b16 {vr(p.ST + 2 * ps, 2)}, {src}" + (f" offset:{off}" if off else ""))
        e.lgkm.issue(("G", g))

    def store(g):
        e.need_lgkm(("G", g))
        if g == 0:
            off = "%[vost]"
        else:
            e.ins(f"v_add_u32 {vr(p.vw)}, {g * 1024}, %[vost]")
            off = vr(p.vw)
        e.ins(f"global_store_dwordx4 {off}, {vr(p.ST, 4)}, %[sst]")
        e.vm.issue(("S", g))

    def mfma(acc, ks, bq, aq, src=None):
        e.need_vm(("B", ks))
        e.need_lgkm(("A", ks, aq))
        e.ins(f"{mn} %[{acc}], {p.b(ks, bq)}, {p.a(ks, aq)}, {src if src is not None else '%[' + acc + ']'}")

    for ks in range(min(depth, nks)):
        load_b(ks)
    read_a(0, [0, 1, 2, 3])
    for ks in range(nks):
        order = [0, 1, 2, 3]
        if ks == 0 and init == "bias":
            order = [1, 2, 3, 0]            # row tile 0 last: the others read the biases out of its accumulators
        seq = []
        for rt in order:
            for nt in range(2):
                src = None
                if ks == 0:
                    src = "0" if init == "zero" else (f"%[c0{nt}]" if rt else None)
                seq.append((f"c{rt}{nt}", nt, rt, src))
        for m, (acc, bq, aq, src) in enumerate(seq):
            mfma(acc, ks, bq, aq, src)
            if m == 0 and ks + depth < nks:
                load_b(ks + depth)
            if carry and m in (1, 2):
                gather(ks, m - 1)
            if m == 3 and ks + 1 < nks:
                read_a(ks + 1, [0, 1])
            if m == 4 and ks + 1 < nks:
                read_a(ks + 1, [2, 3])
            if carry and m == 6:
                store(ks)
    assert not e.lgkm.q, e.lgkm.q
    e.vm.q = []


def gen_plain(paths, init, mn, depth=5):
    """The plain 16-bit statement (see gen_split for the two-path form).  Operands: c00 .. c31 (f32x16), abase, voff, swh, sel, and for a
    carrying path sbt, vost, sst."""
    p = Plan(256 - (8 * (depth + 1) + 32 + 8), depth, 2, 4, any(c for _, c in paths))
    e = Emitter()
    if len(paths) > 1:
        e.ins("s_cmp_eq_u32 %[sel], 0")
        e.ins("s_cbranch_scc0 71f")
    gen_plain_path(e, p, paths[0][0], paths[0][1], init, depth, mn)
    if len(paths) > 1:
        e.ins("s_branch 72f")
        e.lines.append("71:")
        gen_plain_path(e, p, paths[1][0], paths[1][1], init, depth, mn)
        e.lines.append("72:")
    e.ins("s_nop 15")
    return e.text(), p


HEADER = """// GENERATED by tools/gen_mlp16_kloop.py -- do not edit.  The hand-scheduled k-loops of the large-row 16-bit MLP kernels:
// one inline-asm statement per (layer product, variant); register plan, counted waits and the carried saved-tensor stores are
// described in the generator.
"""


def main():
    parts = [HEADER]
    plan = None
    for name, paths, init in (("M16S_FWD_TRAIN", [(16, True), (7, False)], "bias"), ("M16S_FWD_INFER", [(16, False), (7, False)], "bias"),
                              ("M16S_BWD_DATA", [(16, True), (1, False)], "zero")):
        text, plan = gen_split(paths, init)
        parts.append(f"#define {name}_ASM \\\n" + text.replace("\n", " \\\n").rstrip(" \\\n") + "\n")
    parts.append(f"#define M16S_KLOOP_CLOBBERS {plan.clobbers()}\n")
    for tag, mn in (("F16", "v_mfma_f32_32x32x16_f16"), ("BF16", "v_mfma_f32_32x32x16_bf16")):
        for name, paths, init in ((f"M16P_{tag}_FWD_TRAIN", [(16, True), (7, False)], "bias"), (f"M16P_{tag}_FWD_INFER", [(16, False), (7, False)], "bias"),
                                  (f"M16P_{tag}_BWD_DATA", [(16, True), (1, False)], "zero")):
            text, plan = gen_plain(paths, init, mn)
            parts.append(f"#define {name}_ASM \\\n" + text.replace("\n", " \\\n").rstrip(" \\\n") + "\n")
    parts.append(f"#define M16P_KLOOP_CLOBBERS {plan.clobbers()}\n")
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()

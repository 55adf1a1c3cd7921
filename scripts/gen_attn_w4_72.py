#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attention_w472.h (one wave per SIMD, d_h 72: the PixArt heads at 1024 x 1024, N = 4096): writes
ecad_amd/csrc/attention_w472_asm.h.  scripts/gen_attn_w4.py (d_h 128, FLUX) re-cut for a head of 72 = 4.5 k-steps of 16: five k-steps of S = K Q~^T (the last one half
zero: Q~ is zeroed there, the K ring holds zeros), three 32-row blocks of O^T (rows 72..95 unused) -- 20 + 24 MFMAs per 64-key tile instead of 32 + 32 -- and

  * THE ROW SUM COMES OUT OF THE MATRIX PIPE: row 72 of the V^T operand is a column of ones (LDS preset once; the DMA lanes that would write source pieces >= 9 are
    masked off with EXEC, which also keeps every read inside the head's 144 bytes), so O^T[72][q] = sum_k P[q][k] -- the 64 v_add_f32 per tile of the d_h 128 stream
    are gone, and with them the l_run registers, their recentre and their half-exchange: at 44 MFMA gaps per tile there is room for ~1050 issue cycles of fillers
    and the 128-wide stream's ~1230 would not fit.  The sum is of the ROUNDED P (what the numerator multiplies), in fp32.


hipcc will not keep this loop in the shape it is written in (round 5: IR passes sink the softmax slices across sched_barrier, the score accumulators land in
AGPRs and come back through 2000 v_accvgpr moves, the LDS-DMA builtin drags s_waitcnt vmcnt(0) in front of every transposed read).  So the per-work-item body --
Q prescale, pipeline fill, the tile loop, the rare recentre / rescale paths, the output stage -- is ONE asm statement with every register assigned here, and the
C++ around it (attention_w4.h) only walks the work items and computes addresses.  Structure and budget: cdna_hip_programming.md "Fused attention prefill",
MI355X_MICROARCH.md "Per-instruction cycle constants" (<= 24 issue cycles of fillers per 32-cycle MFMA gap, at most one 8-cycle v_exp among them).

Register map (per lane; VGPR v0..v244 and AGPR a0..a255 belong to the statement, v245.. stay with the compiler for the operands):
  v[0:31]    S, key block 0          [rq][16]      scores minus the running maximum, log2 domain (the -m comes in as the C operand of the first MFMA of a tile)
  v[32:63]   S, key block 1, bank X  [rq][16]      (tiles t even)
  v[64:95]   S, key block 1, bank Y                (tiles t odd)
  v[96:127]  C-init                  [rq][16]      every register = -m_run[rq]
  v[128:143] P k-steps 0,1, bank PX  [rq][kst][4]  (tiles t even)
  v[144:159] P k-steps 0,1, bank PY                (tiles t odd)
  v[160:175] P k-steps 2,3           [rq][kst-2][4]
  v[176:195] K fragment window (5 x 4)     v[196:219] V fragment window (6 x 4)
  v[220:224] K read addresses [ks]         v[225:227] V read addresses [db] (+ 64 KiB: the V ring)
  v[228:231] DMA lane offsets K [piece]    v232 DMA lane offset V (piece 0; pieces 1..3 by scalar base)
  v[233:234] pending alpha[rq]   v[235:236] lane maxima   v[237:244] temporaries
  a[0:95]    O^T accumulators [rq][db][16] (db 2: rows 64..71 of the head, row 72 = the row sums)         a[96:135] Q~ fragments [rq][ks][4]
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
import os
ABLATE = set(filter(None, os.environ.get("W472_ABLATE", "").split(",")))   # timing experiments only (wrong results): dma, exp, kread, vread, max, sum, cvt

S0, SX, SY, CI = 0, 32, 64, 96
PX, PY, PZ = 128, 144, 160
KW, VW = 176, 196              # fragment windows: K 5 x 4 registers, V 6 x 4 (one deeper than the d_h 128 stream: its 8 K + 4 V address registers are 5 + 3 here)
KA, VA = 220, 225
KOFF, VOFF = 228, 232          # K: one lane offset per piece (the row XOR differs per piece); V: piece 0, pieces 1..3 by scalar base
PEND, MXP = 233, 235
LRUN = MXP                     # output stage only: the row sums read back from O^T (the lane maxima are dead by then)
TMP = 237                      # 8 temporaries: v237..v244
NV = 245                       # v0..v244 belong to the statement
OACC, QF = 0, 96
NKS, NDB = 5, 3               # k-steps of 16 over the head dim (80 >= 72), 32-row blocks of O^T (96 >= 73)
N1, N2 = 4 * NKS, 8 * NDB     # MFMAs per phase with RQ = 2: S(t+1) = K Q~^T, O^T += V^T P^T
# EXEC masks of the DMA pieces: lane (g = lane >> 4, pos = lane & 15) of piece j writes LDS position pos of row 4 j + g (mod 16) with source piece pos ^ swizzle;
# only source pieces 0..8 (144 bytes) exist
KMASK = [sum(1 << (16 * g + pos) for g in range(4) for pos in range(16) if (pos ^ ((4 * j + g) & 15)) < 9) for j in range(4)]
VMASK = sum(1 << (16 * g + pos) for g in range(4) for pos in range(16) if (pos ^ (g << 2)) < 9)
K_AHEAD, K_WIN = 4, 5
V_AHEAD, V_WIN = 5, 6
THR = 8.0
SLOT = 16384
VRING = 65536

# scalar registers owned by the statement (clobbered); operands are named %[...]
SG = dict(
    loop=52, pend=53, ret=54, icnt=55, ldsw=56, always_lo=58, always_hi=59, tmp0=60, tmp1=61, tmp2=62, tmp3=63,
    kp=64, kp_hi=65, vp=66, vp_hi=67, cmask=68, cmask_hi=69, m0save=70,
    vp1=78, vp1_hi=79, vp2=80, vp2_hi=81, vp3=82, vp3_hi=83, pv=85,
    km0=86, km1=88, km2=90, km3=92, vm=94,     # EXEC masks of the DMA pieces (pairs)
)


def s(name):
    return f"s{SG[name]}"


def s2(name):
    return f"s[{SG[name]}:{SG[name] + 1}]"


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


class Gen:
    def __init__(self, dtype: str, rq: int = 2):
        assert dtype in ("bf16", "f16") and rq in (1, 2)
        self.dt = dtype
        self.RQ = rq   # 32-row query blocks per wave: 2 (256-row work items) or 1 (128-row items: a partial last round of the chip); same arithmetic per row
        self.lines: list[str] = []
        self.mfma = f"v_mfma_f32_32x32x16_{dtype}"
        self.cvt = f"v_cvt_pk_{dtype}_f32"
        self.lds_q: list[str] = []  # outstanding LDS reads (tags), in issue order

    # ---- emission
    def I(self, text):
        self.lines.append("\t" + text)

    def L(self, label):
        self.lines.append(f"{label}_%=:")

    def C(self, text):
        self.lines.append(f"\t; {text}")

    def lbl(self, label):
        return f"{label}_%="

    # ---- LDS reads with counted waits
    def ds(self, text, tag):
        self.I(text)
        self.lds_q.append(tag)

    def need(self, tag):
        """wait until the LDS read(s) tagged `tag` have returned: all reads issued after the last of them may stay in flight"""
        if tag not in self.lds_q:
            return
        last = max(i for i, t in enumerate(self.lds_q) if t == tag)
        after = len(self.lds_q) - 1 - last
        assert after <= 15
        self.I(f"s_waitcnt lgkmcnt({after})")
        self.lds_q = self.lds_q[last + 1:]

    def lds_drain(self):
        if self.lds_q:
            self.I("s_waitcnt lgkmcnt(0)")
            self.lds_q = []

    # ---- pieces of the stream
    def k_read(self, f, slot):
        kb, ks = divmod(f, NKS)
        if "kread" in ABLATE:
            return
        self.ds(f"ds_read_b128 {v(KW + 4 * (f % K_WIN), 4)}, {v(KA + ks)} offset:{slot * SLOT + kb * 8192}", f"K{f}")

    def v_read(self, f, slot):
        db, kst = divmod(f, 4)
        base = VW + 4 * (f % V_WIN)
        if "vread" in ABLATE:
            return
        self.ds(f"ds_read_b64_tr_b16 {v(base, 2)}, {v(VA + db)} offset:{slot * SLOT + kst * 4096}", f"V{f}a")
        self.ds(f"ds_read_b64_tr_b16 {v(base + 2, 2)}, {v(VA + db)} offset:{slot * SLOT + kst * 4096 + 2048}", f"V{f}")

    def qk_mfma(self, i, s1_bank):
        """MFMA i (0..19) of S = K Q~^T: key block i // 10, k-step (i % 10) // 2, query block i % 2"""
        kb, rem = divmod(i, 2 * NKS)
        ks, rq = divmod(rem, 2)
        f = kb * NKS + ks
        d = (S0 if kb == 0 else s1_bank) + 16 * rq
        c = CI + 16 * rq if ks == 0 else d
        if rq >= self.RQ:
            return
        if rq == 0:
            self.need(f"K{f}")
        self.I(f"{self.mfma} {v(d, 16)}, {v(KW + 4 * (f % K_WIN), 4)}, {a(QF + (rq * NKS + ks) * 4, 4)}, {v(c, 16)}")

    def pv_mfma(self, j, p01):
        """MFMA j (0..23) of O^T += V^T P^T: head-dim block j // 8, k-step (j % 8) // 2, query block j % 2"""
        db, rem = divmod(j, 8)
        kst, rq = divmod(rem, 2)
        f = db * 4 + kst
        o = OACC + (rq * NDB + db) * 16
        pf = (p01 + rq * 8 + kst * 4) if kst < 2 else (PZ + rq * 8 + (kst - 2) * 4)
        if rq >= self.RQ:
            return
        if rq == 0:
            self.need(f"V{f}")
        self.I(f"{self.mfma} {a(o, 16)}, {v(VW + 4 * (f % V_WIN), 4)}, {v(pf, 4)}, {a(o, 16)}")

    def dma_piece(self, j, slot):
        """piece j (0..3 K, 4..7 V) of the tile that goes into ring slot `slot`; returns the instructions as a list so the caller can place them"""
        ring = 0 if j < 4 else VRING
        imm = ring + slot * SLOT + (j % 4) * 1024
        base = ["kp", "kp", "kp", "kp", "vp", "vp1", "vp2", "vp3"][j]
        off = KOFF + j if j < 4 else VOFF
        mask = f"km{j}" if j < 4 else "vm"
        return [f"s_add_u32 m0, {s('ldsw')}, {imm}", f"s_mov_b64 exec, {s2(mask)}", f"global_load_lds_dwordx4 {v(off)}, {s2(base)}", "s_mov_b64 exec, -1"]

    def dma_advance(self):
        """after the 8 pieces of a tile: next tile's bases; the [text | image] seam and the hand-over to the next work item's first tiles"""
        out = []
        for p, st in (("kp", "%[tsk]"), ("vp", "%[tsv]")):
            out += [f"s_add_u32 {s(p)}, {s(p)}, {st}", f"s_addc_u32 {s(p + '_hi')}, {s(p + '_hi')}, 0"]
        out += [f"s_add_u32 {s('icnt')}, {s('icnt')}, 1",
                f"s_cmp_eq_u32 {s('icnt')}, %[nt_text]",
                f"s_cselect_b64 {s2('kp')}, %[kimg], {s2('kp')}",
                f"s_cselect_b64 {s2('vp')}, %[vimg], {s2('vp')}",
                f"s_cmp_eq_u32 {s('icnt')}, %[nt]",
                f"s_cselect_b64 {s2('kp')}, %[knext], {s2('kp')}",
                f"s_cselect_b64 {s2('vp')}, %[vnext], {s2('vp')}"]
        out += self.piece_bases()
        return out

    def piece_bases(self):
        out = []
        for p, st in (("vp", s("pv")),):
            prev = p
            for n in (1, 2, 3):
                cur = f"{p}{n}"
                out += [f"s_add_u32 {s(cur)}, {s(prev)}, {st}", f"s_addc_u32 {s(cur + '_hi')}, {s(prev + '_hi')}, 0"]
                prev = cur
        return out

    # exponentials of one S tile half: scores listed as (register, rq); P destination registers per pair
    def exp_stream(self, regs, pdst):
        """returns per-score filler lists: [v_exp], and the lagging conversions, software-pipelined by one slot (no row-sum adds: the ones row of V^T, see above)"""
        ops = []
        n = len(regs)
        for k, (reg, rq) in enumerate(regs):
            cur = [f"{'v_mov_b32_e32' if 'exp' in ABLATE else 'v_exp_f32_e32'} {v(TMP + k % 4)}, {v(reg)}"]
            if k >= 1 and (k - 1) % 2 == 1:
                cur.append(f"{self.cvt} {v(pdst[(k - 1) // 2])}, {v(TMP + (k - 2) % 4)}, {v(TMP + (k - 1) % 4)}")
            ops.append(cur)
        tail = ["s_nop 0", f"{self.cvt} {v(pdst[(n - 1) // 2])}, {v(TMP + (n - 2) % 4)}, {v(TMP + (n - 1) % 4)}"]   # a transcendental's result is not interlocked for the next VALU
        ops.append(tail)
        return ops

    def s_regs(self, bank):
        """the 32 score registers of one key block (both query blocks) in P-fragment order, and their P destinations relative to a bank of [rq][kst][4]"""
        regs, pd = [], []
        for rq in range(self.RQ):
            for e in range(16):
                regs.append((bank + 16 * rq + e, rq))
            for w in range(8):
                pd.append(rq * 8 + w)
        return regs, pd

    def max_stream(self, s1_bank):
        """lane maxima of S(t+1): four chains (rq x key block), then combined into MXP[rq]"""
        ops = []
        t = {(0, 0): TMP + 4, (1, 0): TMP + 5, (0, 1): TMP + 6, (1, 1): TMP + 7}
        for kb in range(2):
            bank = S0 if kb == 0 else s1_bank
            chains = []
            for rq in range(self.RQ):
                b = bank + 16 * rq
                c = [f"v_max3_f32 {v(t[(rq, kb)])}, {v(b)}, {v(b + 1)}, {v(b + 2)}"]
                e = 3
                while e < 16:
                    e2 = min(e + 1, 15)
                    c.append(f"v_max3_f32 {v(t[(rq, kb)])}, {v(t[(rq, kb)])}, {v(b + e)}, {v(b + e2)}")
                    e += 2
                chains.append(c)
            for grp in zip(*chains):
                ops += list(grp)
        for rq in range(self.RQ):
            ops.append(f"v_max_f32_e32 {v(MXP + rq)}, {v(t[(rq, 0)])}, {v(t[(rq, 1)])}")
        return ops

    def decide(self, ret_id):
        """pair maximum over the two lane halves; if any row's tile maximum exceeds the threshold, recentre (rare)"""
        for rq in range(self.RQ):
            self.I(f"v_mov_b32_e32 {v(TMP + 4 + rq)}, {v(MXP + rq)}")
        self.I("s_nop 1")
        for rq in range(self.RQ):
            self.I(f"v_permlane32_swap_b32 {v(TMP + 4 + rq)}, {v(MXP + rq)}")
        for rq in range(self.RQ):
            self.I(f"v_max_f32_e32 {v(MXP + rq)}, {v(MXP + rq)}, {v(TMP + 4 + rq)}")
        self.I(f"v_cmp_lt_f32_e32 vcc, {THR}, {v(MXP)}")
        self.I(f"s_mov_b64 {s2('cmask')}, vcc")
        if self.RQ == 2:
            self.I(f"v_cmp_lt_f32_e32 vcc, {THR}, {v(MXP + 1)}")
            self.I(f"s_or_b64 {s2('cmask')}, {s2('cmask')}, vcc")
        self.I(f"s_cmp_lg_u64 {s2('cmask')}, 0")
        self.I(f"s_movk_i32 {s('ret')}, {ret_id}")
        self.I(f"s_cbranch_scc1 {self.lbl('RECENTRE')}")
        self.L(f"RECENTRE_RET{ret_id}")

    def ret_table(self, prefix, ids):
        for i in ids:
            self.I(f"s_cmp_eq_u32 {s('ret')}, {i}")
            self.I(f"s_cbranch_scc1 {self.lbl(f'{prefix}_RET{i}')}")
        self.I("s_trap 2")

    def recentre_block(self, ids):
        """delta = row maximum where it exceeds the threshold (or everywhere: pipeline fill), subtracted from every live score register and from C-init;
        the pending accumulator factor takes 2^-delta.  No exponential of the tile has been taken yet, so nothing else is at the old scale."""
        self.L("RECENTRE")
        for rq in range(self.RQ):
            self.I(f"v_cmp_lt_f32_e32 vcc, {THR}, {v(MXP + rq)}")
            self.I(f"s_or_b64 vcc, vcc, {s2('always_lo')}")
            self.I(f"v_cndmask_b32_e32 {v(TMP + rq)}, 0, {v(MXP + rq)}, vcc")     # delta
            self.I(f"v_sub_f32_e32 {v(TMP + 2 + rq)}, 0, {v(TMP + rq)}")
        for rq in range(self.RQ):
            self.I(f"v_exp_f32_e32 {v(TMP + 2 + rq)}, {v(TMP + 2 + rq)}")           # alpha
        for bank in (S0, SX, SY, CI):
            for rq in range(self.RQ):
                for e in range(16):
                    r = bank + 16 * rq + e
                    self.I(f"v_sub_f32_e32 {v(r)}, {v(r)}, {v(TMP + rq)}")
        for rq in range(self.RQ):
            self.I(f"v_mul_f32_e32 {v(PEND + rq)}, {v(PEND + rq)}, {v(TMP + 2 + rq)}")   # the row sums live in O^T: they follow with the accumulators
        self.I(f"s_mov_b32 {s('pend')}, 1")
        self.I("s_nop 1")  # VALU writes of S / C-init -> the next MFMA that reads them
        self.ret_table("RECENTRE", ids)

    def rescale_block(self, ids):
        """O *= pending alpha, after every PV MFMA of the tile before has left the pipe"""
        self.L("RESCALE")
        self.I("s_nop 15")
        for rq in range(self.RQ):
            for i in range(16 * NDB):
                r = OACC + rq * 16 * NDB + i
                t = TMP + 2 + (i % 6)
                self.I(f"v_accvgpr_read_b32 {v(t)}, {a(r)}")
                self.I(f"v_mul_f32_e32 {v(t)}, {v(t)}, {v(PEND + rq)}")
                self.I(f"v_accvgpr_write_b32 {a(r)}, {v(t)}")
        for rq in range(self.RQ):
            self.I(f"v_mov_b32_e32 {v(PEND + rq)}, 1.0")
        self.I(f"s_mov_b32 {s('pend')}, 0")
        self.I("s_nop 2")
        self.ret_table("RESCALE", ids)

    # ---- one pipelined iteration; j = position in the unrolled body (ring slots are static: g = 0 mod 4 at the start of a work item)
    def iteration(self, j, has_next, ret_id):
        s1_read = SX if j % 2 == 0 else SY
        s1_write = SY if j % 2 == 0 else SX
        p_cur = PX if j % 2 == 0 else PY
        p_next = PY if j % 2 == 0 else PX
        k_slot, v_slot, d_slot = (j + 1) % 4, j % 4, (j + 3) % 4
        NF_K, NF_V = 2 * NKS, 4 * NDB          # K / V fragments per tile
        self.C(f"==== iteration j={j} has_next={has_next}: phase 1 ({N1} MFMAs)")
        gaps = [[] for _ in range(N1)]
        regs, pd = self.s_regs(s1_read)
        ex = self.exp_stream(regs, [PZ + w for w in pd])
        n = len(regs)                       # 32 scores (RQ 2) or 16
        for k in range(n):
            gaps[k * N1 // n] += ex[k]
        gaps[N1 - 1] += ex[n]
        # DMA of tile t + 3: one piece every second gap (M0, EXEC mask, load, EXEC back), the scalar bookkeeping spread over phase 2
        for pc in range(0 if "dma" in ABLATE else 8):
            m0w, ex_on, ld, ex_off = self.dma_piece(pc, d_slot)
            g = 1 + 2 * pc
            gaps[g] = [m0w, ex_on, ld, ex_off] + gaps[g]     # the EXEC write sits between the M0 write and the DMA that reads it
        adv = self.dma_advance()
        for i in range(N1):
            f = i // 2
            if has_next:
                self.qk_mfma(i, s1_write)
            for t in gaps[i]:
                self.I(t)
            if has_next and i % 2 == 0 and f + K_AHEAD < NF_K:
                self.k_read(f + K_AHEAD, k_slot)
            if i >= N1 - 2 * V_AHEAD and i % 2 == 0:   # first V fragments of this tile for phase 2
                self.v_read((i - (N1 - 2 * V_AHEAD)) // 2, v_slot)
        self.C(f"==== iteration j={j}: phase 2 ({N2} MFMAs)")
        gaps = [[] for _ in range(N2)]
        if has_next:
            mx = self.max_stream(s1_write)
            for k, t in enumerate(mx):
                gaps[k * 6 // len(mx)].append(t)   # 34 (17) instructions over gaps 0..5
            regs, pd = self.s_regs(S0)
            ex = self.exp_stream(regs, [p_next + w for w in pd])
            first, last = 8, N2 - 1
            n = len(regs)
            for k in range(n):
                gaps[first + (k * (last - first + 1)) // n] += ex[k]
            gaps[N2 - 1] += ex[n]
        for k, t in enumerate(adv):
            gaps[7 + k // 3].append(t)           # 25 scalar instructions over gaps 7..15
        bar_gap = N2 - 2 * K_AHEAD - 1
        for jj in range(N2):
            f = jj // 2
            self.pv_mfma(jj, p_cur)
            for t in gaps[jj]:
                self.I(t)
            if jj % 2 == 0 and f + V_AHEAD < NF_V:
                self.v_read(f + V_AHEAD, v_slot)
            if has_next and jj == 6:
                self.decide(ret_id)
            if jj == bar_gap:
                # every V(t) read of this wave has been issued: wait for them; wait for this wave's pieces of tile t + 2 -- the K fragments read right after the
                # barrier are the NEXT iteration's tile t + 2; only tile t + 3, issued in this iteration's phase 1, may stay in flight -- then meet
                self.lds_drain()
                self.I("s_waitcnt vmcnt(8)")
                self.I("s_barrier")
            if jj > bar_gap and jj % 2 == 0:
                self.k_read((jj - bar_gap - 1) // 2, (k_slot + 1) % 4)   # first K fragments of the NEXT iteration's tile
        self.I(f"s_cmp_lg_u32 {s('pend')}, 0")
        self.I(f"s_movk_i32 {s('ret')}, {ret_id}")
        self.I(f"s_cbranch_scc1 {self.lbl('RESCALE')}")
        self.L(f"RESCALE_RET{ret_id}")

    # ---- whole statement
    def build(self):
        self.C(f"attention_w472 asm body ({self.dt}); generated by scripts/gen_attn_w4_72.py -- do not edit")
        I = self.I
        T = TMP
        I(f"s_mov_b32 {s('m0save')}, m0")
        # ---- lane geometry
        I(f"v_mbcnt_lo_u32_b32 {v(T)}, -1, 0")
        I(f"v_mbcnt_hi_u32_b32 {v(T)}, -1, {v(T)}")             # lane
        I(f"v_and_b32_e32 {v(T + 1)}, 31, {v(T)}")              # r
        I(f"v_lshrrev_b32_e32 {v(T + 2)}, 5, {v(T)}")           # hh
        I(f"v_and_b32_e32 {v(T + 3)}, 15, {v(T + 1)}")          # r & 15
        I(f"v_lshlrev_b32_e32 {v(T + 4)}, 8, {v(T + 1)}")       # r * 256
        I(f"v_add_u32_e32 {v(T + 4)}, %[lds], {v(T + 4)}")
        for ks in range(NKS):
            # K fragment (A operand): key row 32 kb + r, piece (2 ks + hh) ^ (r & 15)
            I(f"v_or_b32_e32 {v(T + 5)}, {2 * ks}, {v(T + 2)}")
            I(f"v_xor_b32_e32 {v(T + 5)}, {v(T + 5)}, {v(T + 3)}")
            I(f"v_lshl_add_u32 {v(KA + ks)}, {v(T + 5)}, 4, {v(T + 4)}")
        # V fragment (transposed read): q4 = (lane & 15) >> 2, pp = lane & 3, dhalf = (lane >> 4) & 1:
        #   (4 hh + q4) * 256 + ((4 db + 2 dhalf + (pp >> 1)) ^ (q4 << 2)) * 16 + 8 (pp & 1)   [+ V ring]
        I(f"v_bfe_u32 {v(T + 3)}, {v(T)}, 2, 2")                # q4
        I(f"v_and_b32_e32 {v(T + 4)}, 3, {v(T)}")               # pp
        I(f"v_bfe_u32 {v(T + 5)}, {v(T)}, 4, 1")                # dhalf
        I(f"v_lshl_add_u32 {v(T + 6)}, {v(T + 2)}, 2, {v(T + 3)}")      # 4 hh + q4
        I(f"v_lshlrev_b32_e32 {v(T + 6)}, 8, {v(T + 6)}")               # * 256
        I(f"v_lshrrev_b32_e32 {v(T + 7)}, 1, {v(T + 4)}")               # pp >> 1
        I(f"v_lshl_add_u32 {v(T + 7)}, {v(T + 5)}, 1, {v(T + 7)}")      # 2 dhalf + (pp >> 1)
        I(f"v_and_b32_e32 {v(T + 4)}, 1, {v(T + 4)}")                   # pp & 1
        I(f"v_lshl_add_u32 {v(T + 6)}, {v(T + 4)}, 3, {v(T + 6)}")      # + 8 (pp & 1)
        I(f"v_add_u32_e32 {v(T + 6)}, %[lds], {v(T + 6)}")
        I(f"v_add_u32_e32 {v(T + 6)}, {VRING}, {v(T + 6)}")
        I(f"v_lshlrev_b32_e32 {v(T + 3)}, 2, {v(T + 3)}")               # q4 << 2
        for db in range(NDB):
            I(f"v_or_b32_e32 {v(T + 5)}, {4 * db}, {v(T + 7)}")
            I(f"v_xor_b32_e32 {v(T + 5)}, {v(T + 5)}, {v(T + 3)}")
            I(f"v_lshl_add_u32 {v(VA + db)}, {v(T + 5)}, 4, {v(T + 6)}")
        # DMA lane offsets: piece j of this wave fills LDS rows 16 wave + 4 j + (lane >> 4), position lane & 15, with source piece
        # position ^ (row & 15) (K) / position ^ ((row & 3) << 2) (V)
        I(f"v_lshrrev_b32_e32 {v(T + 3)}, 4, {v(T)}")                   # lam = lane >> 4 = row & 3
        I(f"v_and_b32_e32 {v(T + 4)}, 15, {v(T)}")                      # position
        I(f"s_lshl_b32 {s('tmp0')}, %[wave], 4")
        for j in range(4):
            I(f"v_add_u32_e32 {v(T + 5)}, {4 * j}, {v(T + 3)}")         # row & 15
            I(f"v_xor_b32_e32 {v(T + 6)}, {v(T + 5)}, {v(T + 4)}")      # source piece
            I(f"v_lshlrev_b32_e32 {v(T + 6)}, 4, {v(T + 6)}")
            I(f"v_add_u32_e32 {v(T + 5)}, {s('tmp0')}, {v(T + 5)}")     # row
            I(f"v_mad_u32_u24 {v(KOFF + j)}, {v(T + 5)}, %[rsk], {v(T + 6)}")
        I(f"v_add_u32_e32 {v(T + 5)}, {s('tmp0')}, {v(T + 3)}")         # row of piece 0
        I(f"v_lshlrev_b32_e32 {v(T + 6)}, 2, {v(T + 3)}")
        I(f"v_xor_b32_e32 {v(T + 6)}, {v(T + 6)}, {v(T + 4)}")
        I(f"v_lshlrev_b32_e32 {v(T + 6)}, 4, {v(T + 6)}")
        I(f"v_mad_u32_u24 {v(VOFF)}, {v(T + 5)}, %[rsv], {v(T + 6)}")
        I(f"s_lshl_b32 {s('tmp0')}, %[wave], 12")
        I(f"s_add_u32 {s('ldsw')}, %[lds], {s('tmp0')}")
        I(f"s_lshl_b32 {s('pv')}, %[rsv], 2")
        I(f"s_mov_b64 {s2('kp')}, %[kp]")
        I(f"s_mov_b64 {s2('vp')}, %[vp]")
        I(f"s_mov_b32 {s('icnt')}, 3")
        for t in self.piece_bases():
            I(t)
        for j, m in enumerate(KMASK + [VMASK]):
            base = SG[f"km{j}"] if j < 4 else SG["vm"]
            I(f"s_mov_b32 s{base}, 0x{m & 0xffffffff:08x}")
            I(f"s_mov_b32 s{base + 1}, 0x{m >> 32:08x}")

        # ---- Q: 10 loads of 16 bytes per lane (lane (r, hh) holds Q[q0 + 32 rq + r][16 ks + 8 hh ..]), prescaled by scale * log2 e into a[96:135].  The fifth k-step
        # covers columns 64..79 of a 72-wide head: only the lower lane half (hh = 0: columns 64..71) loads it, the upper half's registers are zeros
        for rq in range(self.RQ):
            for ks in range(NKS - 1):
                I(f"global_load_dwordx4 {v(4 * (rq * NKS + ks), 4)}, %[q{rq}], off offset:{32 * ks}")
            for e in range(4):
                I(f"v_mov_b32_e32 {v(4 * (rq * NKS + NKS - 1) + e)}, 0")
        I("s_mov_b32 exec_hi, 0")
        for rq in range(self.RQ):
            I(f"global_load_dwordx4 {v(4 * (rq * NKS + NKS - 1), 4)}, %[q{rq}], off offset:{32 * (NKS - 1)}")
        I("s_mov_b32 exec_hi, -1")
        I("s_waitcnt vmcnt(0)")   # the Q data -- and this wave's pieces of the first tiles of the work item
        for i in range(4 * NKS * self.RQ):
            if self.dt == "f16":
                I(f"v_pk_mul_f16 {v(i)}, {v(i)}, %[cq]")
            else:
                I(f"v_lshlrev_b32_e32 {v(T)}, 16, {v(i)}")
                I(f"v_and_b32_e32 {v(T + 1)}, 0xffff0000, {v(i)}")
                I(f"v_mul_f32_e32 {v(T)}, %[cq], {v(T)}")
                I(f"v_mul_f32_e32 {v(T + 1)}, %[cq], {v(T + 1)}")
                I(f"{self.cvt} {v(i)}, {v(T)}, {v(T + 1)}")
            I(f"v_accvgpr_write_b32 {a(QF + i)}, {v(i)}")
        # ---- state of a fresh work item: O = 0, running maximum 0 (the first tile recentres unconditionally), l = 0
        for i in range(16 * NDB * self.RQ):
            I(f"v_accvgpr_write_b32 {a(OACC + i)}, 0")
        for i in range(16 * self.RQ):
            I(f"v_mov_b32_e32 {v(CI + i)}, 0")
        for rq in range(self.RQ):
            I(f"v_mov_b32_e32 {v(PEND + rq)}, 1.0")
        I(f"s_mov_b32 {s('pend')}, 0")
        I(f"s_mov_b64 {s2('always_lo')}, -1")
        I("s_barrier")            # every wave's pieces of tiles 0, 1 (and 2) have landed
        # ---- pipeline fill: S(0) with nothing beside it, its row maxima, the unconditional recentre, the exponentials of key block 0
        self.C("==== pipeline fill")
        for f in range(K_AHEAD):
            self.k_read(f, 0)
        for i in range(N1):
            self.qk_mfma(i, SX)
            f = i // 2
            if i % 2 == 0 and f + K_AHEAD < 2 * NKS:
                self.k_read(f + K_AHEAD, 0)
        I("s_nop 15")
        for t in self.max_stream(SX):
            I(t)
        self.decide_always(ret_id=5)
        I(f"s_mov_b64 {s2('always_lo')}, 0")
        I(f"s_mov_b32 {s('pend')}, 0")           # nothing accumulated yet: no rescale due
        for rq in range(self.RQ):
            I(f"v_mov_b32_e32 {v(PEND + rq)}, 1.0")
        regs, pd = self.s_regs(S0)
        for grp in self.exp_stream(regs, [PX + w for w in pd]):
            for t in grp:
                I(t)
        for f in range(K_AHEAD):
            self.k_read(f, 1)
        # ---- the tile loop: 4 iterations per trip (static ring slots), the last iteration of the work item without a successor
        I(f"s_mov_b32 {s('loop')}, %[trips]")
        self.L("LOOP")
        canonical = [f"K{f}" for f in range(K_AHEAD)]
        for j in range(3):
            self.lds_q = list(canonical)
            self.iteration(j, True, j)
        I(f"s_sub_u32 {s('loop')}, {s('loop')}, 1")
        I(f"s_cmp_eq_u32 {s('loop')}, 0")
        I(f"s_cbranch_scc1 {self.lbl('TAIL')}")
        self.lds_q = list(canonical)
        self.iteration(3, True, 3)
        I(f"s_branch {self.lbl('LOOP')}")
        self.L("TAIL")
        self.lds_q = list(canonical)
        self.iteration(3, False, 4)
        # ---- output: the row sums are row 72 of O^T (block 2, register 4 of the lanes hh = 0: rows 8 (e >> 2) + 4 hh + (e & 3)); broadcast to both lane halves,
        # normalise, 8-byte stores (4 consecutive head-dim columns per lane and register group) of rows 0..71
        self.C("==== output")
        self.lds_drain()
        I("s_nop 15")
        for rq in range(self.RQ):
            I(f"v_accvgpr_read_b32 {v(LRUN + rq)}, {a(OACC + (rq * NDB + 2) * 16 + 4)}")
        for rq in range(self.RQ):
            I(f"v_mov_b32_e32 {v(T + rq)}, {v(LRUN + rq)}")
        I("s_nop 1")
        for rq in range(self.RQ):
            I(f"v_permlane32_swap_b32 {v(T + rq)}, {v(LRUN + rq)}")      # upper half of T <-> lower half of LRUN: T = the lower half's value in both halves
        for rq in range(self.RQ):
            I(f"v_rcp_f32_e32 {v(LRUN + rq)}, {v(T + rq)}")
        n = 0
        for rq in range(self.RQ):
            for db in range(NDB):
                for i4 in range(4 if db < 2 else 1):
                    base = 4 * (n % 24)      # rotating registers v0..v95: a store's data is not rewritten while it may still be in flight
                    n += 1
                    for e in range(4):
                        I(f"v_accvgpr_read_b32 {v(base + e)}, {a(OACC + (rq * NDB + db) * 16 + 4 * i4 + e)}")
                    for e in range(4):
                        I(f"v_mul_f32_e32 {v(base + e)}, {v(base + e)}, {v(LRUN + rq)}")
                    I(f"{self.cvt} {v(base)}, {v(base)}, {v(base + 1)}")
                    I(f"{self.cvt} {v(base + 1)}, {v(base + 2)}, {v(base + 3)}")
                    I(f"global_store_dwordx2 %[o{rq}], {v(base, 2)}, off offset:{(32 * db + 8 * i4) * 2}")
        I(f"s_branch {self.lbl('DONE')}")
        # ---- rare paths, shared by every iteration copy (returns through a compare chain on the caller's id)
        self.recentre_block([0, 1, 2, 3, 5])
        self.rescale_block([0, 1, 2, 3, 4])
        self.L("DONE")
        I(f"s_mov_b32 m0, {s('m0save')}")
        return self.lines

    def decide_always(self, ret_id):
        for rq in range(self.RQ):
            self.I(f"v_mov_b32_e32 {v(TMP + 4 + rq)}, {v(MXP + rq)}")
        self.I("s_nop 1")
        for rq in range(self.RQ):
            self.I(f"v_permlane32_swap_b32 {v(TMP + 4 + rq)}, {v(MXP + rq)}")
        for rq in range(self.RQ):
            self.I(f"v_max_f32_e32 {v(MXP + rq)}, {v(MXP + rq)}, {v(TMP + 4 + rq)}")
        self.I(f"s_movk_i32 {s('ret')}, {ret_id}")
        self.I(f"s_branch {self.lbl('RECENTRE')}")
        self.L(f"RECENTRE_RET{ret_id}")


OPERANDS = """\
      : /* no outputs */
      : [kp] "s"(kp), [vp] "s"(vp), [kimg] "s"(kimg), [vimg] "s"(vimg), [knext] "s"(knext), [vnext] "s"(vnext), [tsk] "s"(tsk), [tsv] "s"(tsv), [rsk] "s"(rsk),
        [rsv] "s"(rsv), [nt_text] "s"(nt_text), [nt] "s"(nt), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [cq] "s"(cq), [q0] "v"(q0), [q1] "v"(q1),
        [o0] "v"(o0), [o1] "v"(o1)
"""


def clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"v{i}"' for i in range(NV)]
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"s{i}"' for i in range(52, 96)]
    return ", ".join(c)


def check(lines):
    """static checks on the linear stream: MFMA result -> VALU / LDS / store reader distance, transcendental -> user distance"""
    reg_re = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")

    def regs_of(tok):
        out = set()
        for m in reg_re.finditer(tok):
            if m.group(1):
                out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
            else:
                out.add((m.group(4), int(m.group(5))))
        return out

    last_mfma_write = {}
    last_trans_write = {}
    n = 0
    worst = 10 ** 9
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.endswith(":"):
            if t.endswith(":"):
                last_mfma_write.clear()   # a label: other paths join here; those paths put their own s_nop in front of any reader
            continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")]
        n += 1 + (int(ops[0]) if op == "s_nop" else 0)
        if op.startswith("v_mfma"):
            d, srcs = regs_of(ops[0]), regs_of(ops[1]) | regs_of(ops[2])
            c = regs_of(ops[3])
            for r in srcs | (c - d):
                if r in last_mfma_write:
                    dist = n - last_mfma_write[r]
                    assert dist >= 14, f"MFMA operand {r} read {dist} after an MFMA wrote it: {t}"
            for r in d:
                last_mfma_write[r] = n
            continue
        if op.startswith(("s_", "ds_read", "global_load_lds")) and not op.startswith("ds_read"):
            continue
        allr = set()
        for o in ops:
            allr |= regs_of(o)
        for r in allr:
            if r in last_mfma_write:
                dist = n - last_mfma_write[r]
                worst = min(worst, dist)
                assert dist >= 14, f"{r} touched {dist} instructions after an MFMA wrote it: {t}"
            if r in last_trans_write and not op.startswith(("v_exp", "v_rcp")):
                dist = n - last_trans_write[r]
                assert dist >= 2 or r in regs_of(ops[0]) and False, f"{r} used {dist} after a transcendental wrote it: {t}"
        if op.startswith(("v_exp", "v_rcp")):
            for r in regs_of(ops[0]):
                last_trans_write[r] = n
        else:
            for r in regs_of(ops[0]):
                last_trans_write.pop(r, None)
    return worst


def main():
    out = ROOT / "ecad_amd" / "csrc" / "attention_w472_asm.h"
    parts = ["// attention_w472_asm.h -- GENERATED by scripts/gen_attn_w4_72.py (register map, schedule and rationale there); do not edit.\n#pragma once\n"]
    for dt in ("bf16", "f16"):
        g = Gen(dt, 2)
        lines = g.build()
        worst = check(lines)
        n_mfma = sum("v_mfma" in x for x in lines)
        print(f"{dt}: {len(lines)} lines, {n_mfma} MFMAs, closest non-MFMA touch of an MFMA result: {worst} instructions", file=sys.stderr)
        body = "\n".join('    "' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n"' for ln in lines)
        parts.append(f"#define W472_ASM_{dt.upper()}(kp, vp, kimg, vimg, knext, vnext, tsk, tsv, rsk, rsv, nt_text, nt, trips, lds, wave, cq, q0, q1, o0, o1) \\\n  asm volatile( \\\n"
                     + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                     + "\n".join(x + " \\" for x in OPERANDS.rstrip("\n").split("\n")) + "\n"
                     + "      : " + clobbers() + ")\n")
    text = "\n".join(parts)
    if "--check" in sys.argv:   # the committed header is what this generator writes (tests/test_cpu_host.py)
        if ABLATE or out.read_text() != text:
            print(f"{out} differs from the generator's output", file=sys.stderr)
            sys.exit(1)
        return
    out.write_text(text)
    print(f"wrote {out}", file=sys.stderr)


if __name__ == "__main__":
    main()

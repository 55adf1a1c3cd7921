#!/usr/bin/env python3
"""Generator of the hand-placed K loops of the one-wave-per-SIMD GEMMs (ecad_amd/csrc/gemm4_asm.h): C[256 x BN] += A[256 x K] W[BN x K]^T per workgroup.

Why: the 8-wave GEMMs of this tree (fgemm.h 64 x 128 wave tiles, gemm2w.h 64 x 144) read 22-23 bytes of LDS per KFLOP and issue their MFMAs 65-72 % of the time
under hipcc's schedule; the attention kernel of round 5 (attention_w4.h) showed what a hand-placed one-wave-per-SIMD stream does on this chip (789 -> 1207 TFLOP/s
for the same algorithm, 22 % less energy per flop).  Here: 4 waves as 2 x 2, a wave owns 128 x BN/2 of the tile -- 8 x BN/32 accumulator fragments of
v_mfma_f32_16x16x32 (16-bit operands) or v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, unit block scales: fgemm8.h) in AGPRs; at BN = 256 a 32-deep K step
reads 8 A + 8 W fragments (16 KiB) for 64 MFMAs: 15.6 B/KFLOP, a third less than the 8-wave tile.
Stages of 128 bytes of K per row (64 16-bit or 128 e4m3 elements: A rows then W rows, 16-byte pieces XOR-swizzled by (row >> 1) & 7 on the DMA source side: the
layout of fgemm.h / fgemm8.h) arrive by LDS-DMA into a ring of NS slots (2 at BN 256: 128 KiB; 3 at BN 128: 144 KiB); ONE barrier per stage behind a counted
s_waitcnt vmcnt; fragment reads run ahead of their MFMAs in register sets that alternate.

The statement leaves the accumulators in AGPRs (fragment (i, j) = a[4 (8 i + j) ..]: element r is row m = 16 i + (lane & 15), column n = 16 j + 4 (lane >> 4) + r
of the wave's block -- the orientation of fgemm.h's acc[i][j]) and the C++ epilogue reads them with v_accvgpr_read statements (gemm4.h).

Measured on the FLUX shapes (scripts/probe_gemm4.hip, bf16, BN 256, TFLOP/s at 9216 x 21504 x 3072 / 8192 x 12288 x 3072; fgemm_kernel<256>: 1258 / 1251):
   DMA pieces every 2nd MFMA of sub-step 0, barrier after MFMA 29 of sub-step 1, reads every 2nd MFMA behind it   1378 / 1354
   DMA every 4th (the whole sub-step), barrier after MFMA 45, one read per MFMA behind it                          1453 / 1479   <- product
   DMA every 3rd, barrier 45                                                                                      1434 / 1460
   DMA every 2nd, barrier 45                                                                                      1374 / 1387
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
import os
ABLATE = set(filter(None, os.environ.get("G4_ABLATE", "").split(",")))   # timing experiments only (wrong results): mfma, read, dma
SG = dict(loop=60, ldsw=61, ap=62, ap_hi=63, wp=64, wp_hi=65, m0save=66, tmp0=67, tmp1=68, ldsw2=69, wdst=70, rstep=75, segc=70)   # wdst: s70..s74, rstep: s75 (BN 144 only); segc: s70 (gathered A only)


def s(n):
    return f"s{SG[n]}"


def s2(n):
    return f"s[{SG[n]}:{SG[n] + 1}]"


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


class Gen:
    """dtype bf16 | f16: two 32-deep sub-steps per 64-element stage, fragments of 4 registers, two whole fragment sets alternate.
    dtype fp8: one 128-deep step per stage, fragments of 8 registers (two 16-byte reads), A in two alternating sets, W in a rolling window of four."""

    def __init__(self, dtype, bn, stage=False, gather=False):
        self.dt, self.BN = dtype, bn
        # gathered A (conv4_kernel of conv4.h: a 3 x 3 convolution over zero-bordered NHWC activations as an implicit GEMM): the byte offsets of the lane's eight A
        # pieces come in as operands %[ao0] .. %[ao7] instead of row * stride, and the A pointer walks a cycle of %[kt] steps: %[da] bytes per step, %[db] into the
        # %[kw]-th and the 2 %[kw]-th, the 64-bit %[dc] : %[dch] (usually negative) into the %[kt]-th, which starts the cycle again.  Two K orders use it: 64-channel
        # chunk, kernel row, tap (da = one pixel, db = to the next image row, dc = back to the first tap and 128 bytes on: a tile's nine visits to a pixel's chunk are
        # nine consecutive stages, so it comes from HBM once) and kernel row, tap, channel (da = 128, db = the row jump, kt = all stages: conv3s_kernel's order)
        self.gather = gather
        self.stage = stage                       # BN 288 / 144: after the K loop the statement itself rounds acc + bias into the row-major C tile in LDS (see stage_c)
        self.fp8 = dtype == "fp8"
        # BN 144 (gemm3s4.h: the N = 1152 read-modify-write GEMMs and FF2 of the PixArt path, gemm3s.h's 256 x 144 tile): the waves are stacked 4 x 1 -- a wave owns
        # 64 rows x all 144 columns, 4 A + 9 W fragments per 32-deep step for 36 MFMAs (0.36 fragment reads per MFMA against the 8-wave kernel's 0.61), its 144
        # accumulators in a[0:143]; stages of 50 KiB in a ring of three.  18 W pieces do not divide by four waves: every wave issues five, the last two of wave 3 fetch
        # pieces 0 and 1 a second time (same bytes to the same place; 4 % more operand traffic, one vmcnt for all waves).
        self.narrow = bn == 144
        self.FM = 4 if self.narrow else 8        # A fragments per wave
        self.FN = 9 if self.narrow else bn // 32  # W fragments per wave (wave tile 128 x BN/2, or 64 x 144)
        self.NS = 3 if bn in (128, 144) else 2   # ring slots (BN 288: the PixArt FF1 / K,V tile of gemm2w.h, 2 x 68 KiB)
        self.STAGE = (256 + bn) * 128
        self.WP = 5 if self.narrow else bn // 8 // 4   # W pieces per wave and stage
        self.NP = 8 + self.WP                    # DMA pieces per wave and stage
        self.NM = self.FM * self.FN              # MFMAs per (sub-)step
        self.AX = 0
        if self.fp8:
            self.AY, self.WB = 64, 128            # A sets X / Y (8 x 8 registers), W window 4 x 8
            nxt = 160
        else:
            self.WX, self.AY = 4 * self.FM, 4 * self.FM + 4 * self.FN
            self.WY = self.AY + 4 * self.FM
            nxt = self.WY + 4 * self.FN
        self.AADR = nxt                           # [slot][2]: sub-step 0 / 1 (16-bit) or low / high 16-byte half (e4m3)
        self.WADR = self.AADR + 2 * self.NS
        self.AOFF = self.WADR + 2 * self.NS
        self.WOFF = self.AOFF + 8
        self.TMP = self.WOFF + self.WP
        self.SC = self.TMP + 8                    # e4m3: the unit block scale
        self.NV = self.SC + 1
        if stage and not self.narrow:
            self.C8 = self.NV                     # the ninth accumulator column in registers of the statement (nobody outside reads the accumulators of this variant)
            self.NV += 32
        if self.narrow:
            self.ROFF = self.NV                   # byte offsets of this lane's residual chunks (two mappings, see resid_prefetch)
            self.NV += 2
        self.lines, self.q = [], []
        self.mfma = "v_mfma_scale_f32_16x16x128_f8f6f4" if self.fp8 else f"v_mfma_f32_16x16x32_{dtype}"
        nreads = 10 if self.fp8 else self.FM + self.FN   # read groups behind the barrier
        self.barrier_at = self.NM - nreads - 3
        # every piece of a stage is issued BEFORE the barrier of the (sub-)step that carries it: the counted vmcnt there relies on it.  16-bit: sub-step 0 carries the
        # pieces, sub-step 1 the barrier; e4m3: one step carries both
        self.dma_stride = max(1, (self.barrier_at // self.NP) if self.fp8 else (self.NM // self.NP))
        assert self.dma_stride * (self.NP - 1) < (self.barrier_at if self.fp8 else self.NM)
        self.period = self.NS if (not self.fp8 or self.NS % 2 == 0) else 2 * self.NS

    def acc(self, i, j):
        """accumulator fragment (i, j): AGPRs a[4 (8 i + j) ..] for j < 8; the ninth column of the 288-wide tile lives in eight "+v" operands of the statement"""
        if self.narrow:
            return a(4 * (9 * i + j), 4)
        if j < 8:
            return a(4 * (8 * i + j), 4)
        return v(self.C8 + 4 * i, 4) if self.stage else f"%[c8_{i}]"

    def I(self, t):
        if ("mfma" in ABLATE and t.startswith("v_mfma")) or ("read" in ABLATE and t.startswith("ds_read")) or ("dma" in ABLATE and t.startswith("global_load_lds")):
            return
        self.lines.append("\t" + t)

    def L(self, l):
        self.lines.append(f"{l}_%=:")

    def lbl(self, l):
        return f"{l}_%="

    def ds(self, text, tag):
        self.I(text)
        self.q.append(tag)

    def need(self, tag):
        if tag not in self.q:
            return
        last = max(i for i, t in enumerate(self.q) if t == tag)
        after = len(self.q) - 1 - last
        self.I(f"s_waitcnt lgkmcnt({min(after, 15)})")
        self.q = self.q[last + 1:]

    def dma_piece(self, which, p, slot):
        off = (self.AOFF if which == "a" else self.WOFF) + p
        if self.gather and which == "a":
            ptr, base, imm = "ap", "ldsw", slot * self.STAGE + p * 1024
            return [f"s_add_u32 m0, {s(base)}, {imm}", "s_nop 0", f"global_load_lds_dwordx4 %[ao{p}], {s2(ptr)}"]
        ptr = "ap" if which == "a" else "wp"
        base = "ldsw" if which == "a" else "ldsw2"
        imm = slot * self.STAGE + (0 if which == "a" else 32768) + p * 1024
        if self.narrow and which == "w":       # LDS address of W piece (5 wave + p) mod 18 of slot 0, in its own SGPR (the wrap is not an affine function of the wave)
            return [f"s_add_u32 m0, s{SG['wdst'] + p}, {slot * self.STAGE}", "s_nop 0", f"global_load_lds_dwordx4 {v(off)}, {s2(ptr)}"]
        return [f"s_add_u32 m0, {s(base)}, {imm}", "s_nop 0", f"global_load_lds_dwordx4 {v(off)}, {s2(ptr)}"]

    def advance(self):
        if self.gather:
            t = s('segc')
            return [f"s_add_u32 {t}, {t}, 1", f"s_cmp_eq_u32 {t}, %[kw]", f"s_cselect_b32 {s('tmp0')}, %[db], %[da]", f"s_cmp_eq_u32 {t}, %[kw2]",
                    f"s_cselect_b32 {s('tmp0')}, %[db], {s('tmp0')}", f"s_cmp_eq_u32 {t}, %[kt]", f"s_cselect_b32 {s('tmp0')}, %[dc], {s('tmp0')}",
                    f"s_cselect_b32 {s('tmp1')}, %[dch], 0", f"s_cselect_b32 {t}, 0, {t}", f"s_add_u32 {s('ap')}, {s('ap')}, {s('tmp0')}",
                    f"s_addc_u32 {s('ap_hi')}, {s('ap_hi')}, {s('tmp1')}", f"s_add_u32 {s('wp')}, {s('wp')}, 128", f"s_addc_u32 {s('wp_hi')}, {s('wp_hi')}, 0"]
        return [f"s_add_u32 {s('ap')}, {s('ap')}, 128", f"s_addc_u32 {s('ap_hi')}, {s('ap_hi')}, 0", f"s_add_u32 {s('wp')}, {s('wp')}, 128", f"s_addc_u32 {s('wp_hi')}, {s('wp_hi')}, 0"]

    def dma_stage(self, slot):
        out = {}
        k = 0
        for which, n in (("a", 8), ("w", self.WP)):
            for p in range(n):
                out[self.dma_stride * k] = self.dma_piece(which, p, slot)
                k += 1
        last = self.NM - 1
        out[last] = out.get(last, []) + self.advance()
        return out

    # ------------------------------------------------------------------ 16-bit operands
    def frag_read16(self, which, f, slot, sub, aset, wset):
        base = (aset + 4 * f) if which == "a" else (wset + 4 * f)
        adr = (self.AADR if which == "a" else self.WADR) + 2 * slot + sub
        self.ds(f"ds_read_b128 {v(base, 4)}, {v(adr)} offset:{f * 2048}", f"{aset}{which}{f}")

    def order16(self):
        return [("w", 0)] + [("a", i) for i in range(self.FM)] + [("w", j) for j in range(1, self.FN)]

    def substep16(self, aset, wset, naset, nwset, fillers, reads_next, barrier=None, after_barrier_reads=None):
        rd, ab = list(reads_next), list(after_barrier_reads or [])
        n = 0
        for j in range(self.FN):
            for i in range(self.FM):
                if i == 0:
                    self.need(f"{aset}w{j}")
                if j == 0:
                    self.need(f"{aset}a{i}")
                self.I(f"{self.mfma} {self.acc(i, j)}, {v(wset + 4 * j, 4)}, {v(aset + 4 * i, 4)}, {self.acc(i, j)}")
                for t in fillers.get(n, []):
                    self.I(t)
                if barrier is not None and n == barrier[0]:
                    self.I(f"s_waitcnt vmcnt({barrier[1]})")
                    self.I("s_waitcnt lgkmcnt(0)")
                    self.q = []
                    self.I("s_barrier")
                if rd and n % 2 == 1 and (barrier is None or n < barrier[0]):
                    self.frag_read16(*rd.pop(0), naset, nwset)
                elif ab and barrier is not None and n > barrier[0]:
                    self.frag_read16(*ab.pop(0), naset, nwset)
                n += 1
        assert not rd and not ab, (len(rd), len(ab))

    def step16(self, slot, dma_slot, vm, last, extra=None):
        """one 64-deep stage of ring slot `slot`: sub-step 0 from set X carries the DMA of `dma_slot` (None: nothing left to fetch) and the reads of sub-step 1 into
        set Y; sub-step 1 from set Y carries the barrier (vmcnt(vm)) and the reads of the next slot's sub-step 0 into set X.  `extra`: more fillers of sub-step 0."""
        nslot = (slot + 1) % self.NS
        fill = self.dma_stage(dma_slot) if dma_slot is not None else {}
        for n, ts in (extra or {}).items():
            fill[n] = fill.get(n, []) + ts
        self.substep16(self.AX, self.WX, self.AY, self.WY, fill, [(w, f, slot, 1) for w, f in self.order16()])
        if last:
            self.substep16(self.AY, self.WY, self.AX, self.WX, {}, [])
        else:
            self.substep16(self.AY, self.WY, self.AX, self.WX, {}, [], barrier=(self.barrier_at, vm), after_barrier_reads=[(w, f, nslot, 0) for w, f in self.order16()])

    # ------------------------------------------------------------------ e4m3 operands
    def a_read8(self, i, slot, aset):
        for h in range(2):
            self.ds(f"ds_read_b128 {v(aset + 8 * i + 4 * h, 4)}, {v(self.AADR + 2 * slot + h)} offset:{i * 2048}", f"{aset}a{i}" if h else f"{aset}a{i}lo")

    def w_read8(self, j, slot):
        base = self.WB + 8 * (j % 4)
        for h in range(2):
            self.ds(f"ds_read_b128 {v(base + 4 * h, 4)}, {v(self.WADR + 2 * slot + h)} offset:{j * 2048}", f"w{j}" if h else f"w{j}lo")

    def step8(self, slot, dma_slot, vm, last, aset, naset):
        """one 128-deep stage: W fragment j + 2 is requested when fragment j's MFMAs start (window of four); behind the barrier the next slot's A fragments (other
        set) and its first two W fragments"""
        nslot = (slot + 1) % self.NS
        fill = self.dma_stage(dma_slot) if dma_slot is not None else {}
        ab = [] if last else [("w", 0)] + [("a", i) for i in range(8)] + [("w", 1)]
        n = 0
        for j in range(self.FN):
            for i in range(8):
                if i == 0:
                    self.need(f"w{j}")
                    if j + 2 < self.FN:
                        self.w_read8(j + 2, slot)
                if j == 0:
                    self.need(f"{aset}a{i}")
                acc = 4 * (8 * i + j)
                self.I(f"{self.mfma} {a(acc, 4)}, {v(self.WB + 8 * (j % 4), 8)}, {v(aset + 8 * i, 8)}, {a(acc, 4)}, {v(self.SC)}, {v(self.SC)} op_sel_hi:[0,0,0]")
                for t in fill.get(n, []):
                    self.I(t)
                if not last and n == self.barrier_at:
                    self.I(f"s_waitcnt vmcnt({vm})")
                    self.I("s_waitcnt lgkmcnt(0)")
                    self.q = []
                    self.I("s_barrier")
                if ab and n > self.barrier_at:
                    # the window registers of W 0 / W 1 of the next stage are those of this stage's fragments FN-4 / FN-3: free once their MFMAs have been issued
                    w, f = ab[0]
                    free_at = 8 * (self.FN - 4 + f) + 7 if w == "w" else -1
                    if n > free_at:
                        ab.pop(0)
                        if w == "a":
                            self.a_read8(f, nslot, naset)
                        else:
                            self.w_read8(f, nslot)
                n += 1
        assert not ab, ab

    # ------------------------------------------------------------------ whole statement
    def build(self):
        I, T = self.I, self.TMP
        NS = self.NS
        I(f"s_mov_b32 {s('m0save')}, m0")
        I(f"v_mbcnt_lo_u32_b32 {v(T)}, -1, 0")
        I(f"v_mbcnt_hi_u32_b32 {v(T)}, -1, {v(T)}")                 # lane
        I(f"v_and_b32_e32 {v(T + 1)}, 15, {v(T)}")                  # l15
        I(f"v_lshrrev_b32_e32 {v(T + 2)}, 4, {v(T)}")               # g
        I(f"v_bfe_u32 {v(T + 3)}, {v(T)}, 1, 3")                    # swz = (l15 >> 1) & 7
        b0 = len(self.lines)
        if self.narrow:
            I(f"s_lshl_b32 {s('tmp0')}, %[wave], 13")               # wave * 64 rows * 128 B
            I(f"s_mov_b32 {s('tmp1')}, 0")                          # every wave reads all 144 W rows
        else:
            I(f"s_and_b32 {s('tmp0')}, %[wave], 1")                     # wm
            I(f"s_lshr_b32 {s('tmp1')}, %[wave], 1")                    # wn
            I(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 14")               # wm * 128 rows * 128 B
            I(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, {self.BN // 2 * 128}")   # wn * BN/2 rows * 128 B
        I(f"v_lshlrev_b32_e32 {v(T + 4)}, 7, {v(T + 1)}")           # l15 * 128
        I(f"v_add_u32_e32 {v(T + 4)}, %[lds], {v(T + 4)}")
        for slot in range(NS):
            for h in range(2):
                # 16-bit: sub-step h reads piece 4 h + g; e4m3: the lane's 32 K-elements are pieces 2 g and 2 g + 1
                if self.fp8:
                    I(f"v_lshl_add_u32 {v(T + 5)}, {v(T + 2)}, 1, {h}")
                else:
                    I(f"v_add_u32_e32 {v(T + 5)}, {4 * h}, {v(T + 2)}")
                I(f"v_xor_b32_e32 {v(T + 5)}, {v(T + 5)}, {v(T + 3)}")
                I(f"v_lshl_add_u32 {v(T + 5)}, {v(T + 5)}, 4, {v(T + 4)}")
                I(f"v_add_u32_e32 {v(self.AADR + 2 * slot + h)}, {s('tmp0')}, {v(T + 5)}")
                I(f"v_add_u32_e32 {v(self.WADR + 2 * slot + h)}, {s('tmp1')}, {v(T + 5)}")
                if slot:
                    I(f"v_add_u32_e32 {v(self.AADR + 2 * slot + h)}, {slot * self.STAGE}, {v(self.AADR + 2 * slot + h)}")
                I(f"v_add_u32_e32 {v(self.WADR + 2 * slot + h)}, {slot * self.STAGE + 32768}, {v(self.WADR + 2 * slot + h)}")
        b1 = len(self.lines)
        # DMA lane offsets: piece q of this wave's share covers rows 8 (n wave + q) + (lane >> 3); position lane & 7 holds source piece position ^ ((row >> 1) & 7),
        # (row >> 1) & 7 = (4 (n wave + q) + (lane >> 4)) & 7 = ((q & 1) << 2) | (lane >> 4) because 4 n wave is a multiple of 8 (n = 8 or 4)
        I(f"v_lshrrev_b32_e32 {v(T + 4)}, 3, {v(T)}")               # lane >> 3
        I(f"v_and_b32_e32 {v(T + 5)}, 7, {v(T)}")                   # position
        for which, n, off, rs in (("a", 8, self.AOFF, "%[rsa]"), ("w", self.WP, self.WOFF, "%[rsw]")):
            if self.gather and which == "a":
                continue
            if self.narrow and which == "w":
                for q in range(n):
                    # W piece (5 wave + q) mod 18: rows 8 piece + (lane >> 3), (row >> 1) & 7 = ((4 piece) & 7) | (lane >> 4); its LDS address in slot 0 -> s[wdst + q]
                    I(f"s_mul_i32 {s('tmp1')}, %[wave], 5")
                    I(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {q}")
                    I(f"s_sub_u32 {s('tmp0')}, {s('tmp1')}, 18")
                    I(f"s_cmp_ge_u32 {s('tmp1')}, 18")
                    I(f"s_cselect_b32 {s('tmp1')}, {s('tmp0')}, {s('tmp1')}")
                    I(f"s_lshl_b32 {s('tmp0')}, {s('tmp1')}, 10")
                    I(f"s_add_u32 s{SG['wdst'] + q}, %[lds], {s('tmp0')}")
                    I(f"s_add_u32 s{SG['wdst'] + q}, s{SG['wdst'] + q}, 32768")
                    I(f"s_lshl_b32 {s('tmp0')}, {s('tmp1')}, 3")      # first row of the piece
                    I(f"s_lshl_b32 {s('tmp1')}, {s('tmp1')}, 2")
                    I(f"s_and_b32 {s('tmp1')}, {s('tmp1')}, 7")
                    I(f"v_or_b32_e32 {v(T + 6)}, {s('tmp1')}, {v(T + 2)}")
                    I(f"v_xor_b32_e32 {v(T + 6)}, {v(T + 6)}, {v(T + 5)}")
                    I(f"v_lshlrev_b32_e32 {v(T + 6)}, 4, {v(T + 6)}")
                    I(f"v_add_u32_e32 {v(T + 7)}, {s('tmp0')}, {v(T + 4)}")
                    I(f"v_mad_u32_u24 {v(off + q)}, {v(T + 7)}, {rs}, {v(T + 6)}")
                continue
            I(f"s_mul_i32 {s('tmp0')}, %[wave], {8 * n}")            # first row of this wave's pieces
            for q in range(n):
                # (row >> 1) & 7 with row = 8 (n wave + q) + (lane >> 3): ((4 (n wave + q)) & 7) | (lane >> 4) -- n = 9 makes the first term depend on the wave
                I(f"s_mul_i32 {s('tmp1')}, %[wave], {4 * n}")
                I(f"s_add_u32 {s('tmp1')}, {s('tmp1')}, {4 * q}")
                I(f"s_and_b32 {s('tmp1')}, {s('tmp1')}, 7")
                I(f"v_or_b32_e32 {v(T + 6)}, {s('tmp1')}, {v(T + 2)}")
                I(f"v_xor_b32_e32 {v(T + 6)}, {v(T + 6)}, {v(T + 5)}")
                I(f"v_lshlrev_b32_e32 {v(T + 6)}, 4, {v(T + 6)}")
                I(f"v_add_u32_e32 {v(T + 7)}, {8 * q}, {v(T + 4)}")
                I(f"v_add_u32_e32 {v(T + 7)}, {s('tmp0')}, {v(T + 7)}")
                I(f"v_mad_u32_u24 {v(off + q)}, {v(T + 7)}, {rs}, {v(T + 6)}")
        I(f"s_lshl_b32 {s('tmp0')}, %[wave], 13")                   # this wave's 8 A pieces
        I(f"s_add_u32 {s('ldsw')}, %[lds], {s('tmp0')}")
        I(f"s_mul_i32 {s('tmp0')}, %[wave], {self.WP * 1024}")       # and its WP W pieces
        I(f"s_add_u32 {s('ldsw2')}, %[lds], {s('tmp0')}")
        I(f"s_mov_b64 {s2('ap')}, %[ap]")
        I(f"s_mov_b64 {s2('wp')}, %[wp]")
        if self.gather:
            I(f"s_mov_b32 {s('segc')}, 0")
        b2 = len(self.lines)
        if self.fp8:
            I(f"v_mov_b32_e32 {v(self.SC)}, 0x7f7f7f7f")
        I("s_cmp_eq_u32 %[zero], 0")      # a second K segment (split-A GEMM: another A buffer for the rest of K) goes on accumulating
        I(f"s_cbranch_scc1 {self.lbl('KEEP')}")
        if self.narrow:
            for r in range(4 * self.FM * self.FN):
                I(f"v_accvgpr_write_b32 {a(r)}, 0")
        else:
            for i in range(8):
                for j in range(min(self.FN, 8)):     # (the ninth column's operands are zeroed by the caller)
                    for r in range(4):
                        I(f"v_accvgpr_write_b32 {a(4 * (8 * i + j) + r)}, 0")
        self.L("KEEP")
        if self.stage and not self.narrow:
            for r in range(32):
                I(f"v_mov_b32_e32 {v(self.C8 + r)}, 0")
        if self.narrow:
            # residual chunks of this lane (resid_prefetch): byte offsets from %[rbase] = &h[m0 + 64 wave][n0]
            I(f"v_lshrrev_b32_e32 {v(T + 6)}, 4, {v(T)}")                        # lane >> 4
            I(f"v_and_b32_e32 {v(T + 7)}, 15, {v(T)}")                           # lane & 15
            I(f"v_lshlrev_b32_e32 {v(T + 7)}, 4, {v(T + 7)}")
            I(f"v_mad_u32_u24 {v(self.ROFF)}, {v(T + 6)}, %[ldrb], {v(T + 7)}")  # mapping A: row 4 it + (lane >> 4), chunk lane & 15
            I(f"v_lshrrev_b32_e32 {v(T + 6)}, 1, {v(T)}")                        # lane >> 1
            I(f"v_and_b32_e32 {v(T + 7)}, 1, {v(T)}")
            I(f"v_lshlrev_b32_e32 {v(T + 7)}, 4, {v(T + 7)}")
            I(f"v_add_u32_e32 {v(T + 7)}, 256, {v(T + 7)}")
            I(f"v_mad_u32_u24 {v(self.ROFF + 1)}, {v(T + 6)}, %[ldrb], {v(T + 7)}")   # mapping B: row 32 (it - 16) + (lane >> 1), chunk 16 + (lane & 1)
            I(f"s_lshl_b32 s{SG['rstep']}, %[ldrb], 2")                          # four rows
        b3 = len(self.lines)
        for slot in range(NS - 1):          # the first NS - 1 stages
            for which, n in (("a", 8), ("w", self.WP)):
                for p in range(n):
                    for t in self.dma_piece(which, p, slot):
                        I(t)
            for t in self.advance():
                I(t)
        if self.narrow:
            # BN 144: the first two stages are requested as soon as their addresses exist; the fragment addresses, the 144 accumulator zeros and the residual offsets
            # are computed while they travel (every block below sets the temporaries it uses; a one-round kernel pays this prologue once per launch)
            L = self.lines
            self.lines = L[:b0] + L[b1:b2] + L[b3:] + L[b0:b1] + L[b2:b3]
        I(f"s_waitcnt vmcnt({self.NP * (NS - 2)})")
        I("s_barrier")
        if self.fp8:
            self.w_read8(0, 0)
            for i in range(8):
                self.a_read8(i, 0, self.AX)
            self.w_read8(1, 0)
        else:
            for w, f in self.order16():
                self.frag_read16(w, f, 0, 0, self.AX, self.WX)
        canonical = list(self.q)
        # %[trips] = K stages / period - 1 full trips, then the last trip: its stages fetch nothing beyond the end of K
        I(f"s_mov_b32 {s('loop')}, %[trips]")
        I(f"s_cmp_eq_u32 {s('loop')}, 0")
        I(f"s_cbranch_scc1 {self.lbl('LAST')}")
        vm = self.NP * (NS - 2)

        def emit(k, dma, vmc, last, extra=None):
            slot = k % NS
            if self.fp8:
                aset, naset = (self.AX, self.AY) if k % 2 == 0 else (self.AY, self.AX)
                self.q = [f"{aset}{t[len(str(self.AX)):]}" if t.startswith(f"{self.AX}a") else t for t in canonical]
                self.step8(slot, (slot + NS - 1) % NS if dma else None, vmc, last, aset, naset)
            else:
                self.q = list(canonical)
                self.step16(slot, (slot + NS - 1) % NS if dma else None, vmc, last, extra)

        self.L("LOOP")
        for k in range(self.period):
            emit(k, True, vm, False)
        I(f"s_sub_u32 {s('loop')}, {s('loop')}, 1")
        I(f"s_cmp_eq_u32 {s('loop')}, 0")
        I(f"s_cbranch_scc0 {self.lbl('LOOP')}")
        self.L("LAST")
        for k in range(self.period):
            left = self.period - 1 - k           # stages after this one
            if self.narrow and self.stage and left == 1:
                # the second-to-last stage has no DMA left to issue: the NRES residual loads of the read-modify-write epilogue go out here, behind every operand piece,
                # so this stage's counted wait lets exactly them stay in flight (vmcnt retires in order) -- an HBM round trip that travels under two stages of MFMAs
                # and the C-tile staging instead of in front of the row pass (gemm3s.h issues its residual loads in the same place)
                emit(k, False, self.NRES, False, self.resid_prefetch())
                continue
            emit(k, left >= NS - 1, vm if left >= NS - 1 else 0, left == 0)
        I("s_nop 15")
        I("s_barrier")   # every wave is done with the ring: the epilogue's C tile may overwrite it
        if self.stage:
            if self.narrow:
                self.stage_c144()
                I("s_waitcnt vmcnt(0)")   # the residual chunks are the statement's outputs: the compiler does not know they are loads in flight
            else:
                self.stage_c()
        I(f"s_mov_b32 m0, {s('m0save')}")
        return self.lines

    NRES = 18

    def resid_prefetch(self):
        """fillers of sub-step 0 (even MFMA slots; the fragment reads sit on the odd ones): this lane's 18 sixteen-byte chunks of the wave's 64 x 144 block of hidden
        states into the statement's outputs %[h0] .. %[h17].  Mapping A (h0 .. h15): row 4 it + (lane >> 4), chunk lane & 15 -- 256 contiguous bytes of four rows per
        load; mapping B (h16, h17): row 32 (it - 16) + (lane >> 1), chunk 16 + (lane & 1).  The row pass of gemm3s4.h uses the same mapping."""
        out = {}
        for it in range(self.NRES):
            ts = []
            if 1 <= it <= 15:
                ts.append(f"v_add_u32_e32 {v(self.ROFF)}, s{SG['rstep']}, {v(self.ROFF)}")
            if it == 17:
                ts.append(f"s_lshl_b32 {s('tmp0')}, %[ldrb], 5")
                ts.append(f"v_add_u32_e32 {v(self.ROFF + 1)}, {s('tmp0')}, {v(self.ROFF + 1)}")
            ts.append(f"global_load_dwordx4 %[h{it}], {v(self.ROFF if it < 16 else self.ROFF + 1)}, %[rbase]")
            out[2 * it] = ts
        return out

    def stage_c144(self):
        """BN 144: round(acc + bias) of the wave's 64 x 144 block into the row-major C tile at the start of the LDS (row stride 304 bytes = gemm3s.h's CS), the 144 bias
        values read from LDS at %[lds] + CST144.  Rows 64 wave .. 64 wave + 63 are written and later read by this wave only: no barrier behind it."""
        I, T = self.I, self.TMP
        BV, TT, PK, B0 = 0, 40, 56, 64
        CS = 304
        cvt = "v_cvt_pk_bf16_f32" if self.dt == "bf16" else "v_cvt_pk_f16_f32"
        I(f"v_mbcnt_lo_u32_b32 {v(T)}, -1, 0")
        I(f"v_mbcnt_hi_u32_b32 {v(T)}, -1, {v(T)}")
        I(f"v_and_b32_e32 {v(T + 1)}, 15, {v(T)}")                  # l15
        I(f"v_lshrrev_b32_e32 {v(T + 2)}, 4, {v(T)}")               # g
        I(f"s_mul_i32 {s('tmp0')}, %[wave], {64 * CS}")
        I(f"v_mul_u32_u24_e32 {v(B0)}, {CS}, {v(T + 1)}")           # l15 rows
        I(f"v_add_u32_e32 {v(B0)}, {s('tmp0')}, {v(B0)}")
        I(f"v_lshl_add_u32 {v(B0)}, {v(T + 2)}, 3, {v(B0)}")        # + 4 g columns x 2 bytes
        I(f"v_add_u32_e32 {v(B0)}, %[lds], {v(B0)}")
        I(f"s_add_u32 {s('tmp1')}, %[lds], {CST144}")
        I(f"v_lshl_add_u32 {v(T + 4)}, {v(T + 2)}, 4, {s('tmp1')}")  # bias of column 16 j + 4 g: 64 j + 16 g bytes
        for j in range(9):
            I(f"ds_read_b128 {v(BV + 4 * j, 4)}, {v(T + 4)} offset:{64 * j}")
        I("s_waitcnt lgkmcnt(0)")
        n = 0
        for i in range(4):
            for j in range(9):
                t, pk = TT + 4 * (n % 4), PK + 2 * (n % 4)
                for r in range(4):
                    I(f"v_accvgpr_read_b32 {v(t + r)}, {a(4 * (9 * i + j) + r)}")
                I(f"v_pk_add_f32 {v(t, 2)}, {v(t, 2)}, {v(BV + 4 * j, 2)}")
                I(f"v_pk_add_f32 {v(t + 2, 2)}, {v(t + 2, 2)}, {v(BV + 4 * j + 2, 2)}")
                I(f"{cvt} {v(pk)}, {v(t)}, {v(t + 1)}")
                I(f"{cvt} {v(pk + 1)}, {v(t + 2)}, {v(t + 3)}")
                I(f"ds_write_b64 {v(B0)}, {v(pk, 2)} offset:{i * 16 * CS + 32 * j}")
                n += 1
        I("s_waitcnt lgkmcnt(0)")

    def stage_c(self):
        """round(acc + bias) of the wave's 128 x 144 block into the row-major C tile at the start of the LDS (row stride 592 bytes = gemm2w.h's G2_CS), the 288 bias
        values read from LDS at %[cst].  Inside the statement because hipcc treats the AGPRs as free spill space once the K loop is behind it: a C++ epilogue that is
        short of registers (the read-modify-write row pass holds 36 residual chunks per lane) had its spills land in the accumulators.  Same arithmetic as the C++
        epilogues: fp32 add, then one conversion with the default rounding."""
        I, T = self.I, self.TMP
        BV, TT, PK, B0 = 0, 40, 56, 64       # bias [9][4] | 4 x 4 temporaries | 4 x 2 packed | two row bases (fragment rows 0-3 / 4-7: 16-bit ds offsets)
        cvt = "v_cvt_pk_bf16_f32" if self.dt == "bf16" else "v_cvt_pk_f16_f32"
        I(f"v_mbcnt_lo_u32_b32 {v(T)}, -1, 0")
        I(f"v_mbcnt_hi_u32_b32 {v(T)}, -1, {v(T)}")
        I(f"v_and_b32_e32 {v(T + 1)}, 15, {v(T)}")                  # l15
        I(f"v_lshrrev_b32_e32 {v(T + 2)}, 4, {v(T)}")               # g
        I(f"s_and_b32 {s('tmp0')}, %[wave], 1")                     # wm
        I(f"s_lshr_b32 {s('tmp1')}, %[wave], 1")                    # wn
        I(f"s_mul_i32 {s('tmp0')}, {s('tmp0')}, {128 * 592}")
        I(f"s_mul_i32 {s('tmp1')}, {s('tmp1')}, 144")               # first column of the wave's block
        I(f"v_lshl_add_u32 {v(T + 3)}, {v(T + 2)}, 2, {s('tmp1')}")  # column wn 144 + 4 g
        I(f"v_mul_u32_u24_e32 {v(B0)}, 592, {v(T + 1)}")             # (wm 128 + l15) rows (VOP3 takes no literal on gfx9)
        I(f"v_add_u32_e32 {v(B0)}, {s('tmp0')}, {v(B0)}")
        I(f"v_lshl_add_u32 {v(B0)}, {v(T + 3)}, 1, {v(B0)}")         # + 2 bytes per column
        I(f"v_add_u32_e32 {v(B0)}, %[lds], {v(B0)}")
        I(f"v_add_u32_e32 {v(B0 + 1)}, {4 * 16 * 592}, {v(B0)}")
        I(f"v_lshl_add_u32 {v(T + 4)}, {v(T + 3)}, 2, %[cst]")       # bias: 4 bytes per column
        for j in range(9):
            I(f"ds_read_b128 {v(BV + 4 * j, 4)}, {v(T + 4)} offset:{64 * j}")
        I("s_waitcnt lgkmcnt(0)")
        n = 0
        for i in range(8):
            for j in range(9):
                t, pk = TT + 4 * (n % 4), PK + 2 * (n % 4)
                if j < 8:
                    for r in range(4):
                        I(f"v_accvgpr_read_b32 {v(t + r)}, {a(4 * (8 * i + j) + r)}")
                    src = t
                else:
                    src = self.C8 + 4 * i
                I(f"v_pk_add_f32 {v(t, 2)}, {v(src, 2)}, {v(BV + 4 * j, 2)}")
                I(f"v_pk_add_f32 {v(t + 2, 2)}, {v(src + 2, 2)}, {v(BV + 4 * j + 2, 2)}")
                I(f"{cvt} {v(pk)}, {v(t)}, {v(t + 1)}")
                I(f"{cvt} {v(pk + 1)}, {v(t + 2)}, {v(t + 3)}")
                I(f"ds_write_b64 {v(B0 + i // 4)}, {v(pk, 2)} offset:{(i % 4) * 16 * 592 + 32 * j}")
                n += 1
        I("s_waitcnt lgkmcnt(0)")


CST144 = 3 * (256 + 144) * 128      # gemm3s4.h: bias[144] fp32 | gate table[144] fp32 | gate modulation[144] T, behind the three ring slots


OPERANDS = """\
      : /* accumulators stay in AGPRs */
      : [ap] "s"(ap), [wp] "s"(wp), [rsa] "s"(rsa), [rsw] "s"(rsw), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero)
"""
OPERANDS_G = ("""\
      : /* accumulators stay in AGPRs */
      : [ap] "s"(ap), [wp] "s"(wp), [rsw] "s"(rsw), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero), [da] "s"(da), [db] "s"(db), [dc] "s"(dc), [dch] "s"(dch), [kw] "s"(kw), [kw2] "s"(kw2), [kt] "s"(kt), """
              + ", ".join(f'[ao{i}] "v"(ao[{i}])' for i in range(8)) + "\n")
OPERANDS_288 = """\
      : [c8_0] "+v"(c8[0]), [c8_1] "+v"(c8[1]), [c8_2] "+v"(c8[2]), [c8_3] "+v"(c8[3]), [c8_4] "+v"(c8[4]), [c8_5] "+v"(c8[5]), [c8_6] "+v"(c8[6]), [c8_7] "+v"(c8[7])
      : [ap] "s"(ap), [wp] "s"(wp), [rsa] "s"(rsa), [rsw] "s"(rsw), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero)
"""


OPERANDS_288S = """\
      : /* accumulators stay inside the statement: it leaves the rounded C tile in LDS */
      : [ap] "s"(ap), [wp] "s"(wp), [rsa] "s"(rsa), [rsw] "s"(rsw), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero), [cst] "s"(cst)
"""


OPERANDS_144S = ("      : " + ", ".join(f'[h{i}] "=&v"(h[{i}])' for i in range(18)) + """
      : [ap] "s"(ap), [wp] "s"(wp), [rsa] "s"(rsa), [rsw] "s"(rsw), [trips] "s"(trips), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero), [rbase] "s"(rbase), [ldrb] "s"(ldrb)
""")


def clobbers(nv, s_hi=70):
    c = ['"memory"', '"vcc"', '"scc"'] + [f'"v{i}"' for i in range(nv)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(60, s_hi)]
    return ", ".join(c)


def main():
    out = ROOT / "ecad_amd" / "csrc" / "gemm4_asm.h"
    parts = ["// gemm4_asm.h -- GENERATED by scripts/gen_gemm4.py (register map, schedule and rationale there); do not edit.\n#pragma once\n"]
    for dt in ("bf16", "f16", "fp8"):
        for bn in (256, 128, 288):
            if bn == 288 and dt == "fp8":
                continue
            g = Gen(dt, bn)
            lines = g.build()
            print(f"{dt} BN {bn}: {len(lines)} lines, {sum('v_mfma' in x for x in lines)} MFMAs, {g.NV} VGPRs, stages per trip {g.period}, barrier at {g.barrier_at}, DMA stride {g.dma_stride}", file=sys.stderr)
            body = "\n".join('    "' + ln + '\\n"' for ln in lines)
            parts.append(f"#define GEMM4_PERIOD_{dt.upper()}_{bn} {g.period}  // K stages per trip of the loop: a K segment must hold a whole number of them\n"
                         f"#define GEMM4_ASM_{dt.upper()}_{bn}(ap, wp, rsa, rsw, trips, lds, wave, zero{', c8' if bn == 288 else ''}) \\\n  asm volatile( \\\n" + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                         + "\n".join(x + " \\" for x in (OPERANDS_288 if bn == 288 else OPERANDS).rstrip("\n").split("\n")) + "\n      : " + clobbers(g.NV) + ")\n")
    for dt in ("bf16", "f16"):
        g = Gen(dt, 288, stage=True)
        lines = g.build()
        print(f"{dt} BN 288 + C-tile staging: {len(lines)} lines, {g.NV} VGPRs", file=sys.stderr)
        body = "\n".join('    "' + ln + '\\n"' for ln in lines)
        parts.append(f"#define GEMM4_ASM_{dt.upper()}_288S(ap, wp, rsa, rsw, trips, lds, wave, zero, cst) \\\n  asm volatile( \\\n" + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                     + "\n".join(x + " \\" for x in OPERANDS_288S.rstrip("\n").split("\n")) + "\n      : " + clobbers(g.NV) + ")\n")
    for dt in ("bf16", "f16"):
        g = Gen(dt, 144, stage=True)
        lines = g.build()
        print(f"{dt} BN 144 (4 x 1 waves) + residual prefetch + C-tile staging: {len(lines)} lines, {sum('v_mfma' in x for x in lines)} MFMAs, {g.NV} VGPRs, stages per trip {g.period}, "
              f"barrier at {g.barrier_at}, DMA stride {g.dma_stride}", file=sys.stderr)
        body = "\n".join('    "' + ln + '\\n"' for ln in lines)
        parts.append(f"#define GEMM4_CST144 {CST144}  // LDS offset of the epilogue constants of the BN 144 statements\n"
                     f"#define GEMM4_PERIOD_144 {g.period}\n" if dt == "bf16" else "")
        parts.append(f"#define GEMM4_ASM_{dt.upper()}_144S(ap, wp, rsa, rsw, trips, lds, wave, zero, rbase, ldrb, h) \\\n  asm volatile( \\\n" + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                     + "\n".join(x + " \\" for x in OPERANDS_144S.rstrip("\n").split("\n")) + "\n      : " + clobbers(g.NV, 76) + ")\n")
    for dt in ("bf16", "f16"):
        for bn in (256, 128):
            g = Gen(dt, bn, gather=True)
            lines = g.build()
            print(f"{dt} BN {bn}, gathered A: {len(lines)} lines, {sum('v_mfma' in x for x in lines)} MFMAs, {g.NV} VGPRs", file=sys.stderr)
            body = "\n".join('    "' + ln + '\\n"' for ln in lines)
            parts.append(f"#define GEMM4_ASM_{dt.upper()}_{bn}G(ap, wp, rsw, trips, lds, wave, zero, da, db, dc, dch, kw, kw2, kt, ao) \\\n  asm volatile( \\\n" + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                         + "\n".join(x + " \\" for x in OPERANDS_G.rstrip("\n").split("\n")) + "\n      : " + clobbers(g.NV, 71) + ")\n")
    text = "\n".join(parts)
    if "--check" in sys.argv:   # the committed header is what this generator writes (tests/test_cpu_host.py)
        if ABLATE or out.read_text() != text:
            print(f"{out} differs from the generator's output", file=sys.stderr)
            sys.exit(1)
        return
    out.write_text(text)


if __name__ == "__main__":
    main()

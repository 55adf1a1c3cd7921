#!/usr/bin/env python3
"""Generator of the hand-placed K loop of the one-wave-per-SIMD GEMM (ecad_amd/csrc/gemm4_asm.h): C[256 x 256] += A[256 x K] W[256 x K]^T per workgroup.

Why: the 8-wave GEMMs of this tree (fgemm.h 64 x 128 wave tiles, gemm2w.h 64 x 144) read 22-23 bytes of LDS per KFLOP and issue their MFMAs 65-72 % of the time
under hipcc's schedule; the attention kernel of round 5 (attention_w4.h) showed what a hand-placed one-wave-per-SIMD stream does on this chip (789 -> 1207 TFLOP/s
for the same algorithm, 22 % less energy per flop).  Here: 4 waves as 2 x 2, a wave owns 128 x 128 of the tile -- 8 x 8 accumulator fragments of
v_mfma_f32_16x16x32 in a[0:255] -- so a 32-deep K step reads 8 A + 8 W fragments (16 KiB) for 64 MFMAs: 15.6 B/KFLOP, a third less than the 8-wave tile.
Stages of 64 K elements (A rows then W rows, 128-byte rows, 16-byte pieces XOR-swizzled by (row >> 1) & 7 on the DMA source side: the layout of fgemm.h) arrive
by LDS-DMA into a two-slot ring; one barrier per stage; fragment reads run one 32-deep sub-step ahead of their MFMAs in two register sets.

The statement leaves the accumulators in a[0:255] (fragment (i, j) = a[4 (8 i + j) ..]: element r is row m = 16 i + (lane & 15), column n = 16 j + 4 (lane >> 4) + r
of the wave's 128 x 128 block -- the orientation of fgemm.h's acc[i][j]) and the C++ epilogue reads them with v_accvgpr_read statements (gemm4.h).

Register map: a[0:255] accumulators; v[0:63] fragment set X (A 0..7, W 0..7), v[64:127] set Y; v[128:131] A read addresses [stage][sub-step], v[132:135] W read
addresses; v[136:143] DMA lane offsets A [piece], v[144:151] W [piece]; v[152:159] temporaries.
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SETX, SETY = 0, 64
AADR, WADR = 128, 132
AOFF, WOFF = 136, 144
TMP = 152
NV = 160
STAGE = 65536          # bytes per ring slot: (256 + 256) rows x 128 B
SG = dict(loop=60, ldsw=61, ap=62, ap_hi=63, wp=64, wp_hi=65, m0save=66, tmp0=67, tmp1=68)


def s(n):
    return f"s{SG[n]}"


def s2(n):
    return f"s[{SG[n]}:{SG[n] + 1}]"


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


class Gen:
    def __init__(self, dtype, dma_stride=2, barrier_at=29, reads_per_mfma_after=0.5):
        self.dma_stride, self.barrier_at, self.rpm = dma_stride, barrier_at, reads_per_mfma_after
        self.dt = dtype
        self.mfma = f"v_mfma_f32_16x16x32_{dtype}"
        self.lines = []
        self.q = []

    def I(self, t):
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

    def frag_read(self, which, f, stage, sub, fset):
        """fragment f (0..7) of A ('a') or W ('w') for K sub-step `sub` of ring slot `stage` into register set fset"""
        base = fset + (0 if which == "a" else 32) + 4 * f
        adr = (AADR if which == "a" else WADR) + 2 * stage + sub
        self.ds(f"ds_read_b128 {v(base, 4)}, {v(adr)} offset:{f * 2048}", f"{fset}{which}{f}")

    def dma_piece(self, which, p, stage):
        off = (AOFF if which == "a" else WOFF) + p
        ptr = "ap" if which == "a" else "wp"
        imm = stage * STAGE + (0 if which == "a" else 32768) + p * 1024
        return [f"s_add_u32 m0, {s('ldsw')}, {imm}", "s_nop 0", f"global_load_lds_dwordx4 {v(off)}, {s2(ptr)}"]

    def substep(self, stage, sub, cur, nxt, fillers, reads_next, barrier_at=None, after_barrier_reads=None):
        """64 MFMAs of one 32-deep sub-step from register set `cur`; `reads_next` = list of (which, f, stage, sub) issued one per two MFMAs into set `nxt`;
        `fillers` = dict MFMA index -> instruction list; at MFMA `barrier_at`: drain + barrier, after which `after_barrier_reads` are issued"""
        rd = list(reads_next)
        ab = list(after_barrier_reads or [])
        n = 0
        for j in range(8):            # W fragment j with every A fragment: the W operand stays, accumulators a[4 (8 i + j)]
            for i in range(8):
                if i == 0:
                    self.need(f"{cur}w{j}")
                if j == 0:
                    self.need(f"{cur}a{i}")
                acc = 4 * (8 * i + j)
                self.I(f"{self.mfma} {a(acc, 4)}, {v(cur + 32 + 4 * j, 4)}, {v(cur + 4 * i, 4)}, {a(acc, 4)}")
                for t in fillers.get(n, []):
                    self.I(t)
                if barrier_at is not None and n == barrier_at:
                    self.I("s_waitcnt vmcnt(0)")
                    self.I("s_waitcnt lgkmcnt(0)")
                    self.q = []
                    self.I("s_barrier")
                if rd and n % 2 == 1 and (barrier_at is None or n < barrier_at):
                    self.frag_read(*rd.pop(0), nxt)
                elif ab and barrier_at is not None and n > barrier_at and (self.rpm >= 1 or n % 2 == 1):
                    self.frag_read(*ab.pop(0), nxt)
                n += 1
        assert not rd and not ab, (len(rd), len(ab))

    def step(self, stage, last):
        """one 64-deep stage: sub-step 0 from set X (already read), its MFMAs carry the DMA of the next stage and the reads of sub-step 1 into set Y;
        sub-step 1 from set Y carries the barrier and the reads of the next stage's sub-step 0 into set X"""
        nst = stage ^ 1
        fill = {}
        if not last:
            k = 0
            for which in ("a", "w"):
                for p in range(8):
                    fill[self.dma_stride * k] = self.dma_piece(which, p, nst)
                    k += 1
            fill[63] = fill.get(63, []) + [f"s_add_u32 {s('ap')}, {s('ap')}, 128", f"s_addc_u32 {s('ap_hi')}, {s('ap_hi')}, 0",
                        f"s_add_u32 {s('wp')}, {s('wp')}, 128", f"s_addc_u32 {s('wp_hi')}, {s('wp_hi')}, 0"]
        order = [("w", 0), ("a", 0), ("a", 1), ("a", 2), ("a", 3), ("a", 4), ("a", 5), ("a", 6), ("a", 7)] + [("w", j) for j in range(1, 8)]
        self.substep(stage, 0, SETX, SETY, fill, [(w, f, stage, 1) for w, f in order])
        if last:
            self.substep(stage, 1, SETY, SETX, {}, [])
        else:
            self.substep(stage, 1, SETY, SETX, {}, [], barrier_at=self.barrier_at, after_barrier_reads=[(w, f, nst, 0) for w, f in order])

    def build(self):
        I = self.I
        T = TMP
        I(f"s_mov_b32 {s('m0save')}, m0")
        I(f"v_mbcnt_lo_u32_b32 {v(T)}, -1, 0")
        I(f"v_mbcnt_hi_u32_b32 {v(T)}, -1, {v(T)}")                 # lane
        I(f"v_and_b32_e32 {v(T + 1)}, 15, {v(T)}")                  # l15
        I(f"v_lshrrev_b32_e32 {v(T + 2)}, 4, {v(T)}")               # g
        I(f"v_bfe_u32 {v(T + 3)}, {v(T)}, 1, 3")                    # swz = (l15 >> 1) & 7
        I(f"s_and_b32 {s('tmp0')}, %[wave], 1")                     # wm
        I(f"s_lshr_b32 {s('tmp1')}, %[wave], 1")                    # wn
        I(f"s_lshl_b32 {s('tmp0')}, {s('tmp0')}, 14")               # wm * 128 rows * 128 B
        I(f"s_lshl_b32 {s('tmp1')}, {s('tmp1')}, 14")
        I(f"v_lshlrev_b32_e32 {v(T + 4)}, 7, {v(T + 1)}")           # l15 * 128
        I(f"v_add_u32_e32 {v(T + 4)}, %[lds], {v(T + 4)}")
        for stage in range(2):
            for sub in range(2):
                I(f"v_add_u32_e32 {v(T + 5)}, {4 * sub}, {v(T + 2)}")            # piece 4 sub + g
                I(f"v_xor_b32_e32 {v(T + 5)}, {v(T + 5)}, {v(T + 3)}")
                I(f"v_lshl_add_u32 {v(T + 5)}, {v(T + 5)}, 4, {v(T + 4)}")
                I(f"v_add_u32_e32 {v(AADR + 2 * stage + sub)}, {s('tmp0')}, {v(T + 5)}")
                I(f"v_add_u32_e32 {v(WADR + 2 * stage + sub)}, {s('tmp1')}, {v(T + 5)}")
                if stage:
                    I(f"v_add_u32_e32 {v(AADR + 2 * stage + sub)}, {STAGE}, {v(AADR + 2 * stage + sub)}")
                I(f"v_add_u32_e32 {v(WADR + 2 * stage + sub)}, {stage * STAGE + 32768}, {v(WADR + 2 * stage + sub)}")
        # DMA lane offsets: piece q of this wave = piece 8 wave + q of the 32 (rows 8 (8 wave + q) + (lane >> 3)), position lane & 7 holds source piece
        # position ^ ((row >> 1) & 7) = position ^ (((q & 1) << 2) | (lane >> 4))
        I(f"v_lshrrev_b32_e32 {v(T + 4)}, 3, {v(T)}")               # lane >> 3
        I(f"v_and_b32_e32 {v(T + 5)}, 7, {v(T)}")                   # position
        I(f"s_lshl_b32 {s('tmp0')}, %[wave], 6")                    # first row of this wave's pieces
        for q in range(8):
            I(f"v_or_b32_e32 {v(T + 6)}, {(q & 1) << 2}, {v(T + 2)}")
            I(f"v_xor_b32_e32 {v(T + 6)}, {v(T + 6)}, {v(T + 5)}")
            I(f"v_lshlrev_b32_e32 {v(T + 6)}, 4, {v(T + 6)}")
            I(f"v_add_u32_e32 {v(T + 7)}, {8 * q}, {v(T + 4)}")
            I(f"v_add_u32_e32 {v(T + 7)}, {s('tmp0')}, {v(T + 7)}")
            I(f"v_mad_u32_u24 {v(AOFF + q)}, {v(T + 7)}, %[rsa], {v(T + 6)}")
            I(f"v_mad_u32_u24 {v(WOFF + q)}, {v(T + 7)}, %[rsw], {v(T + 6)}")
        I(f"s_lshl_b32 {s('tmp0')}, %[wave], 13")                   # 8 pieces x 1 KiB
        I(f"s_add_u32 {s('ldsw')}, %[lds], {s('tmp0')}")
        I(f"s_mov_b64 {s2('ap')}, %[ap]")
        I(f"s_mov_b64 {s2('wp')}, %[wp]")
        I("s_cmp_eq_u32 %[zero], 0")      # a second K segment (split-A GEMM: another A buffer for the rest of K) goes on accumulating
        I(f"s_cbranch_scc1 {self.lbl('KEEP')}")
        for i in range(256):
            I(f"v_accvgpr_write_b32 {a(i)}, 0")
        self.L("KEEP")
        # stage 0 (ring slot 0), then the first fragment set
        for which in ("a", "w"):
            for p in range(8):
                for t in self.dma_piece(which, p, 0):
                    I(t)
        for t in (f"s_add_u32 {s('ap')}, {s('ap')}, 128", f"s_addc_u32 {s('ap_hi')}, {s('ap_hi')}, 0", f"s_add_u32 {s('wp')}, {s('wp')}, 128", f"s_addc_u32 {s('wp_hi')}, {s('wp_hi')}, 0"):
            I(t)
        I("s_waitcnt vmcnt(0)")
        I("s_barrier")
        order = [("w", 0)] + [("a", i) for i in range(8)] + [("w", j) for j in range(1, 8)]
        for w, f in order:
            self.frag_read(w, f, 0, 0, SETX)
        canonical = list(self.q)
        # pairs of stages (ring slots 0, 1); %[pairs] = K / 128 - 1 full pairs, then the last pair without a successor for its second stage
        I(f"s_mov_b32 {s('loop')}, %[pairs]")
        I(f"s_cmp_eq_u32 {s('loop')}, 0")
        I(f"s_cbranch_scc1 {self.lbl('LAST')}")
        self.L("LOOP")
        self.q = list(canonical)
        self.step(0, False)
        self.q = list(canonical)
        self.step(1, False)
        I(f"s_sub_u32 {s('loop')}, {s('loop')}, 1")
        I(f"s_cmp_eq_u32 {s('loop')}, 0")
        I(f"s_cbranch_scc0 {self.lbl('LOOP')}")
        self.L("LAST")
        self.q = list(canonical)
        self.step(0, False)
        self.q = list(canonical)
        self.step(1, True)
        I("s_nop 15")
        I("s_barrier")   # every wave is done with the ring: the epilogue's C tile may overwrite it
        I(f"s_mov_b32 m0, {s('m0save')}")
        return self.lines


OPERANDS = """\
      : /* accumulators stay in a[0:255] */
      : [ap] "s"(ap), [wp] "s"(wp), [rsa] "s"(rsa), [rsw] "s"(rsw), [pairs] "s"(pairs), [lds] "s"(lds), [wave] "s"(wave), [zero] "s"(zero)
"""


def clobbers():
    c = ['"memory"', '"vcc"', '"scc"'] + [f'"v{i}"' for i in range(NV)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(60, 69)]
    return ", ".join(c)


# measured on the FLUX shapes (scripts/probe_gemm4.hip, TFLOP/s at 9216 x 21504 x 3072 / 8192 x 12288 x 3072; fgemm_kernel<256>: 1258 / 1251):
#   DMA pieces every 2nd MFMA of sub-step 0, barrier after MFMA 29 of sub-step 1, reads every 2nd MFMA behind it   1378 / 1354
#   DMA every 4th (the whole sub-step), barrier after MFMA 45, one read per MFMA behind it                          1453 / 1479   <- product
#   DMA every 3rd, barrier 45                                                                                      1434 / 1460
#   DMA every 2nd, barrier 45                                                                                      1374 / 1387
VARIANTS = {0: dict(dma_stride=4, barrier_at=45, reads_per_mfma_after=1)}


def main():
    out = ROOT / "ecad_amd" / "csrc" / "gemm4_asm.h"
    parts = ["// gemm4_asm.h -- GENERATED by scripts/gen_gemm4.py (register map, schedule and rationale there); do not edit.\n#pragma once\n"]
    for dt in ("bf16", "f16"):
        for vid, kw in VARIANTS.items():
            g = Gen(dt, **kw)
            lines = g.build()
            print(f"{dt} v{vid}: {len(lines)} lines, {sum('v_mfma' in x for x in lines)} MFMAs", file=sys.stderr)
            body = "\n".join('    "' + ln + '\\n"' for ln in lines)
            parts.append(f"#define GEMM4_ASM_{dt.upper()}_V{vid}(ap, wp, rsa, rsw, pairs, lds, wave, zero) \\\n  asm volatile( \\\n" + "\n".join(x + " \\" for x in body.split("\n")) + "\n"
                         + "\n".join(x + " \\" for x in OPERANDS.rstrip("\n").split("\n")) + "\n      : " + clobbers() + ")\n")
    out.write_text("\n".join(parts))


if __name__ == "__main__":
    main()

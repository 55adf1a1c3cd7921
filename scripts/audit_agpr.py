#!/usr/bin/env python3
"""Build audit of the kernels whose K loops are generated asm statements (gemm4.h, gemm2w4.h, attention_w4.h): hipcc knows the AGPRs only as clobbers of the statement,
so once the statement is behind it they are free registers to its allocator -- and a C++ epilogue short of VGPRs spills INTO them.  That is harmless after the last
accumulator read (`g4_acc`: v_accvgpr_read inside its own small asm statements) and silently wrong before it (round 5: the read-modify-write epilogue of gemm2w4.h; the
fix was to round the accumulators into the C tile inside the statement).  This script compiles the two translation units to assembly and reports, per kernel, compiler-issued
AGPR accesses (outside ;;#ASMSTART .. ;;#ASMEND) and whether any of them comes before the kernel's last accumulator read.  Exit status 1 on a hazard.

    python scripts/audit_agpr.py                        # compiles both units with the product flags (~4 minutes of hipcc)
    python scripts/audit_agpr.py --asm build/engine-hip-amdgcn-amd-amdhsa-gfx950.s ...   # audits the assembly the product build kept (-save-temps): what the Makefile runs

The product build runs the second form on the very assembly its objects were made from (ecad_amd/csrc/Makefile), so `make` -- and with it `__graft_entry__.build()` -- fails on
a hazard.  Order is judged by text line inside a kernel body: right for the straight-line epilogues these kernels have (the tile walk of gemm2w4 is one loop whose body is
audited as a whole); a kernel that branches around its accumulator reads would need a control-flow walk.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffast-math -fno-finite-math-only -DECAD_PRODUCT_BUILD -S --cuda-device-only".split()  # BASEFLAGS of the Makefile
KERNELS = ("fgemm4", "fgemm48", "attn_w4a", "attn_w472", "gemm2w4", "gemm3s4", "conv4")


def audit(asm: str, unit: str) -> int:
    lines = asm.split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_ZN4ecad\S+):", l))]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    hazards = audited = 0
    for (a, name), b in zip(starts, ends):
        if not any(k in name for k in KERNELS):
            continue
        audited += 1
        inasm, last_read, first_c, cnt, scratch = False, None, None, 0, 0
        for i in range(a, b):
            l = lines[i]
            if "ASMSTART" in l:
                inasm = True
            elif "ASMEND" in l:
                inasm = False
            elif inasm and "v_accvgpr_read_b32" in l:
                last_read = i
            elif not inasm:
                if "scratch_" in l:
                    scratch += 1
                if "v_accvgpr" in l or re.search(r"[ ,]a\[?\d+", l.split(";")[0]):
                    cnt += 1
                    first_c = i if first_c is None else first_c
        hazard = first_c is not None and last_read is not None and first_c < last_read
        hazards += hazard
        if cnt or scratch:
            print(f"{unit}: {name[9:80]}: compiler AGPR accesses {cnt} ({'HAZARD: before the last accumulator read' if hazard else 'all after the last accumulator read'}), scratch instructions {scratch}")
    print(f"{unit}: {audited} kernels with asm-statement K loops audited, {hazards} hazards")
    return hazards


def main() -> int:
    bad = 0
    if "--asm" in sys.argv:
        files = [Path(a) for a in sys.argv[sys.argv.index("--asm") + 1:]]
        if not files:
            print("--asm needs at least one assembly file", file=sys.stderr)
            return 2
        total = 0
        for f in files:
            text = f.read_text()
            if "ASMSTART" not in text:
                print(f"{f}: no asm-statement markers: not the device assembly of a unit with generated K loops", file=sys.stderr)
                return 2
            bad += audit(text, f.name)
            total += 1
        return 1 if bad else 0
    with tempfile.TemporaryDirectory() as td:
        for unit in ("flux.hip", "engine.hip", "vae.hip"):
            out = Path(td) / (unit + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(ROOT / "ecad_amd" / "csrc" / unit), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
            bad += audit(out.read_text(), unit)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

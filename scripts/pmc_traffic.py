#!/usr/bin/env python3
"""HBM bytes per launch and per kernel class from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary --no-decode-tail
    python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_a_pmc_traffic.json [workload=alpha256_b16] [dtype=f16]

(the program directly after `--`, counter passes with no trace domains; `--no-secondary --no-decode-tail` matter: the default bench line also runs the VAE
tail and six child benchmarks, whose GEMM / attention launches would be folded into this workload's classes.  scripts/collect_profiles.sh runs exactly
these commands for alpha256, and with `--workload sigma1024` / `--workload flux1024` for BASELINE configs 3 / 5.)

Corrections exactly as the guide's HBM section prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so it is doubled; WRITE_SIZE is exact.  The output carries a `_stamp` (workload, dtype, sha256 of the kernel sources it was
measured on -- run this script on the same tree as the counter passes); bench.py quotes `roofline.traffic` only from a profile whose stamp
matches its own run and reports null otherwise.
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

# kernel symbol (demangled or mangled substring) -> class name used by the engine's profiler / bench.py.  TAG template arguments make
# every class its own symbol: gemm kernels <T, EPI, TAG> with TAG = the KClass index; replay <T, 0 | 1>.
GEMM_TAGS = {0: "gemm_qkv", 1: "gemm_attn1_out", 2: "gemm_attn2_q", 3: "gemm_attn2_out", 4: "gemm_ff1_gelu", 5: "gemm_ff2", 6: "gemm_text_kv",
             7: "gemm_caption", 8: "gemm_small", 9: "gemm_attn1_kv_wide"}


def classify_flux(name: str, tiles: int) -> str | None:
    """FLUX-1024 batch 2 (flux.hip): the GEMM symbols are fgemm_kernel<T, BN, EPI> (fgemm8_kernel in fp8 mode) without a class tag, so a class is
    (EPI, BN, grid): R = 9216 rows (single-stream), 8192 image / 1024 text rows (joint blocks), D = 3072.  The joint attention out-projection and
    the joint FF2 share symbol AND grid (N = 3072, gated residual epilogue; only K differs): one merged class."""
    m = re.search(r"fgemm(?:4|8|48)?_kernel.*?Li(\d+)ELi(\d+)E", name) or re.search(r"fgemm(?:4|8|48)?_kernel<[^,]+, *(\d+), *(\d+)>", name)  # gemm4.h: fgemm4 / fgemm48
    if m:
        bn, epi = int(m.group(1)), int(m.group(2))
        def t(rows, n):
            return ((rows + 255) // 256) * (n // bn)
        table = {
            (4, t(8192, 9216)): "gemm_joint_qkv", (4, t(1024, 9216)): "gemm_joint_qkv", (0, t(8192, 9216)): "gemm_joint_qkv", (0, t(1024, 9216)): "gemm_joint_qkv",
            (1, t(8192, 12288)): "gemm_joint_ff1_gelu", (1, t(1024, 12288)): "gemm_joint_ff1_gelu",
            (2, t(8192, 3072)): "gemm_joint_attn_out+ff2", (2, t(1024, 3072)): "gemm_joint_attn_out+ff2",
            (5, t(9216, 21504)): "gemm_single_qkv", (0, t(9216, 9216)): "gemm_single_qkv",
            (1, t(9216, 12288)): "gemm_single_mlp_gelu", (2, t(9216, 3072)): "gemm_single_proj_out",
        }
        return table.get((epi, tiles))
    if "attn128_kernel" in name or "attn_kernel" in name or "attn_dma_kernel" in name or "attn_w4a_kernel" in name:
        return "attn_joint"
    if "flux_ln_modulate" in name:
        return "ln_modulate"
    if "flux_replay" in name:
        return "cache_replay"
    if "flux_qk_norm_rope" in name:
        return "rope"
    if "quant_rows_fp8" in name:
        return "quant_fp8"
    return None


def classify(name: str) -> str | None:
    m4 = re.search(r"gemm3s4_kernel.*?Li(\d+)E", name) or re.search(r"gemm3s4_kernel<[^,]+, *(\d+)>", name)  # gemm3s4.h: <T, TAG>
    if m4:
        return GEMM_TAGS.get(int(m4.group(1)))
    m = re.search(r"(gemm2w4|gemm2w|gemm3s|pgemm)_kernel.*?Li(\d+)ELi(\d+)E", name) or re.search(r"(gemm2w4|gemm2w|gemm3s|pgemm)_kernel<[^,]+, *(\d+), *(\d+)>", name)
    if m:
        return GEMM_TAGS.get(int(m.group(3)))
    if "cache_accumulate" in name:  # cache_accumulate_kernel<T, TAG> / cache_accumulate_rows_kernel<T, TAG, R>
        tag = re.search(r"cache_accumulate\w*kernel.*?Li(\d)E", name) or re.search(r"cache_accumulate\w*kernel<[^,]+, *(\d)", name)
        return "cache_hit_step" if tag and tag.group(1) == "1" else "cache_replay"
    if "ln_modulate" in name:
        return "ln_modulate"
    if "attn_dma_kernel" in name or "attn_w472_kernel" in name:  # attention_dma.h / attention_w472.h: unmasked self-attention only
        return "attn_self"
    if "attn128_kernel" in name or "attn_kernel" in name:
        masked = re.search(r"Lb([01])E", name) or re.search(r"(true|false)", name)
        if masked is None:
            return "attn"
        return "attn_cross" if masked.group(1) in ("1", "true") else "attn_self"
    if "patch_embed" in name:
        return "patch_embed"
    if "final_layer_kernel" in name:
        return "step_tail"
    return None


def read_pass(directory: Path, counter: str, flux: bool = False):
    per = defaultdict(lambda: [0, 0.0, "", defaultdict(int)])  # class -> [launches, sum of counter values, symbol, launches per symbol]
    files = list(directory.rglob("*counter_collection.csv"))
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                if flux:
                    wg = max(1, int(float(row.get("Workgroup_Size") or 512)))
                    cls = classify_flux(row["Kernel_Name"], int(float(row.get("Grid_Size") or 0)) // wg)
                else:
                    cls = classify(row["Kernel_Name"])
                if cls is None:
                    continue
                e = per[cls]
                e[0] += 1
                e[1] += float(row["Counter_Value"])
                e[2] = row["Kernel_Name"]
                e[3][row["Kernel_Name"]] += 1
    return per


def main():
    fetch_dir, write_dir, out = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
    workload = sys.argv[4] if len(sys.argv) > 4 else "alpha256_b16"
    flux = workload.startswith("flux")
    fetch, write = read_pass(fetch_dir, "FETCH_SIZE", flux), read_pass(write_dir, "WRITE_SIZE", flux)
    res = {}
    for cls in sorted(set(fetch) | set(write)):
        nf, sf, sym, syms = fetch.get(cls, [0, 0.0, "", {}])
        nw, sw, sym2, syms2 = write.get(cls, [0, 0.0, "", {}])
        fkb = sf / nf if nf else 0.0
        wkb = sw / nw if nw else 0.0
        res[cls] = {"kernel": sym or sym2, "launches": nf or nw, "fetch_size_kb_raw_per_launch": fkb, "write_size_kb_per_launch": wkb,
                    "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
                    "note": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM); WRITE_SIZE exact; separate --pmc passes, eager (no graph) run"}
        # an engine op that goes out as several kernel launches (FLUX joint attention: whole rounds + the 128-row remainder, two symbols launched equally often):
        # the engine's profiler and bench.py count ONE launch per op, so the figure to set beside their algorithmic bytes is the sum over the op's launches
        counts = sorted((syms or syms2).values())
        if len(counts) > 1 and counts[0] == counts[-1]:
            res[cls]["kernel_launches_per_op"] = len(counts)
            res[cls]["hbm_bytes_per_op"] = res[cls]["hbm_bytes_per_launch"] * len(counts)
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench

    res["_stamp"] = {"workload": workload, "dtype": sys.argv[5] if len(sys.argv) > 5 else "f16",
                     "kernel_sources": bench.kernel_source_stamp()}
    out.write_text(json.dumps(res, indent=1))
    for k, v in res.items():
        if k.startswith("_"):
            continue
        print(f"{k:22s} launches {v['launches']:5d}  HBM {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB / launch")


if __name__ == "__main__":
    main()

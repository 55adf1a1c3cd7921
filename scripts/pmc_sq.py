#!/usr/bin/env python3
"""MFMA utilisation and wave-state fractions per kernel class from rocprofv3 counter passes over the PRODUCT library (no diagnostic build):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py ...
    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -- python3 bench.py ...
    python3 scripts/pmc_sq.py gpurun_out/pmc_sq gpurun_out/pmc_grbm <kernel_stats.csv of the --kernel-trace --stats run> profiles/r05_x_pmc_sq.json [workload] [dtype]

(scripts/collect_profiles.sh runs exactly these; the program sits directly after `--`, counter passes carry no trace domains.)

What the numbers mean (MI355X_MICROARCH.md "rocprofv3 PMC slots", "Per-instruction cycle constants"):
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over all SIMDs (4 x 256); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
    count quad-cycles summed over waves.  GRBM_GUI_ACTIVE is summed over the 8 XCDs: busy cycles of the launch = GRBM_GUI_ACTIVE / 8.
  * mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): the share of the launch's cycles in which the matrix pipes work --
    the counter-based MFMA utilisation north_star asks for (flops / time / peak additionally contains the clock).
  * effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / average duration of the class's launches in the kernel trace (reads high for launches under ~0.3 ms).
  * waves: parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY), active (SQ_ACTIVE_INST_ANY) as fractions of SQ_WAVE_CYCLES.
Counter passes serialise the launches and run 2-5 % below the un-profiled clock: fractions are meaningful, absolute times come from the kernel trace.
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic import classify, classify_flux  # noqa: E402

SQ = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]


def cls_of(row, flux):
    if flux:
        wg = max(1, int(float(row.get("Workgroup_Size") or 512)))
        return classify_flux(row["Kernel_Name"], int(float(row.get("Grid_Size") or 0)) // wg)
    return classify(row["Kernel_Name"])


def read_counters(directory: Path, names, flux):
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(int))
    files = list(directory.rglob("*counter_collection.csv"))
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") not in names:
                    continue
                c = cls_of(row, flux)
                if c is None:
                    continue
                per[c][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[c][row["Counter_Name"]] += 1
    return per, launches


def read_durations(stats_csv: Path, flux):
    """class -> (calls, total ns) from the kernel_stats.csv of a --kernel-trace --stats run of the same command (FLUX GEMM classes need the grid, which the stats
    file does not carry: they fall back to the durations of the counter pass's own kernel trace when present, else no clock is quoted)"""
    out = defaultdict(lambda: [0, 0.0])
    if not stats_csv.exists():
        return out
    with open(stats_csv, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name") or row.get("Kernel_Name") or ""
            c = classify_flux(name, -1) if flux else classify(name)
            if c is None:
                continue
            out[c][0] += int(float(row.get("Calls") or 0))
            out[c][1] += float(row.get("TotalDurationNs") or 0.0)
    return out


def main():
    sq_dir, grbm_dir, stats_csv, out = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3]), Path(sys.argv[4])
    workload = sys.argv[5] if len(sys.argv) > 5 else "alpha256_b16"
    flux = workload.startswith("flux")
    sq, n_sq = read_counters(sq_dir, set(SQ), flux)
    gr, n_gr = read_counters(grbm_dir, {"GRBM_GUI_ACTIVE"}, flux)
    dur = read_durations(stats_csv, flux)
    res = {}
    for c in sorted(set(sq) | set(gr)):
        s = sq.get(c, {})
        n1 = max(1, n_sq.get(c, {}).get("SQ_WAVE_CYCLES", 0))
        n2 = max(1, n_gr.get(c, {}).get("GRBM_GUI_ACTIVE", 0))
        gui = gr.get(c, {}).get("GRBM_GUI_ACTIVE", 0.0) / n2 / 8.0            # busy cycles per launch
        mfma = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n1
        wave = s.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches_sq_pass": n1, "launches_grbm_pass": n2, "busy_cycles_per_launch": gui,
             "mfma_busy_cycles_per_launch_all_simds": mfma,
             "mfma_busy_frac": mfma / (4 * 256 * gui) if gui else None,
             "wave_parked_frac": s.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
             "wave_issue_stalled_frac": s.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
             "wave_active_frac": s.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None,
             "cu_busy_quad_cycles_per_launch": s.get("SQ_BUSY_CU_CYCLES", 0.0) / n1}
        calls, ns = dur.get(c, [0, 0.0])
        if calls and gui:
            e["avg_duration_us_kernel_trace"] = ns / calls / 1e3
            e["effective_clock_ghz"] = gui / (ns / calls)
        res[c] = e
    import bench

    res["_stamp"] = {"workload": workload, "dtype": sys.argv[6] if len(sys.argv) > 6 else "f16", "kernel_sources": bench.kernel_source_stamp(),
                     "counters": SQ + ["GRBM_GUI_ACTIVE"], "units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*: quad-cycles summed over waves; GRBM_GUI_ACTIVE: cycles summed over 8 XCDs"}
    out.write_text(json.dumps(res, indent=1))
    for k, v in res.items():
        if k.startswith("_") or v["mfma_busy_frac"] is None:
            continue
        print(f"{k:24s} mfma busy {v['mfma_busy_frac']:.3f}  parked {v['wave_parked_frac'] or 0:.2f}  issue-stalled {v['wave_issue_stalled_frac'] or 0:.2f}  active {v['wave_active_frac'] or 0:.2f}"
              f"  clock {v.get('effective_clock_ghz', 0):.2f} GHz")


if __name__ == "__main__":
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    main()

#!/usr/bin/env python3
"""HBM traffic of the HIP VAE decode by kernel, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/time_vae.py:
    python scripts/pmc_vae.py <fetch_dir> <write_dir> <out.json> [decodes_in_the_run]
Same corrections as scripts/pmc_traffic.py (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md HBM; WRITE_SIZE exact; counter values in KiB).  The decoder's algorithmic bytes
per batch of 16 (every convolution reads its input once and writes its output once, GroupNorm apply reads + writes, weights once): DESIGN 6b."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def read_pass(directory: Path, counter: str):
    per = defaultdict(lambda: [0, 0.0])
    for f in directory.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter or "ecad" not in row["Kernel_Name"]:
                    continue
                name = row["Kernel_Name"].split("ecad", 1)[1]
                key = next((k for k in ("conv4_kernel", "conv3s_kernel", "conv_gemm_kernel", "gn_apply_pad", "gn_apply", "gn_stats", "gn_finalize", "upsample_pad", "vae_conv_out", "vae_conv_in",
                                        "softmax_rows", "transpose") if k in name), None)
                if key is None:
                    continue
                per[key][0] += 1
                per[key][1] += float(row["Counter_Value"])
    return per


def main():
    fetch, write = read_pass(Path(sys.argv[1]), "FETCH_SIZE"), read_pass(Path(sys.argv[2]), "WRITE_SIZE")
    decodes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    res, tot = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(k, [0, 0.0])
        nw, sw = write.get(k, [0, 0.0])
        b = (2.0 * sf + sw) * 1024.0 / decodes
        res[k] = {"launches_per_decode": (nf or nw) / decodes, "hbm_read_bytes_per_decode": 2.0 * sf * 1024.0 / decodes, "hbm_write_bytes_per_decode": sw * 1024.0 / decodes, "hbm_bytes_per_decode": b}
        tot += b
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench

    res["_total_hbm_bytes_per_decode"] = tot
    res["_stamp"] = {"workload": "vae_decode_b16_32x32", "kernel_sources": bench.kernel_source_stamp()}
    Path(sys.argv[3]).write_text(json.dumps(res, indent=1))
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:18s} {v['launches_per_decode']:6.1f} launches  read {v['hbm_read_bytes_per_decode'] / 1e6:9.1f} MB  write {v['hbm_write_bytes_per_decode'] / 1e6:9.1f} MB per decode")
    print(f"total {tot / 1e9:.2f} GB per decode")


if __name__ == "__main__":
    main()

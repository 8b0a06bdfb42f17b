#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs of tools/ntt_pmc.py (FETCH_SIZE, WRITE_SIZE) into
profiles/<tag>_ntt_traffic.json: calibrated HBM bytes per NTT launch and per transform."""
import csv
import glob
import json
import sys

LOG_N, COLS = 19, 1024
N = (1 << LOG_N) * COLS


def counters(d, name):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def main(fetch_dir, write_dir, out):
    f, w = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")

    def pick(rows, key):
        return [v for _, k, v in rows if key in k]

    # calibration: k_batch reads 16*N bytes, k_fill_random / k_batch write 8*N bytes
    cal_f = 16.0 * N / pick(f, "k_batch")[0]
    cal_w = 8.0 * N / pick(w, "k_batch")[0]
    cal_w2 = 8.0 * N / pick(w, "k_fill_random")[-1]
    ntt_f, ntt_w = pick(f, "k_ntt"), pick(w, "k_ntt")  # k_ntt3 (round 3) or k_ntt_tile
    per_launch = [(a * cal_f + b * cal_w) for a, b in zip(ntt_f, ntt_w)]
    launches = len(per_launch)
    per_transform = sum(per_launch) / (launches / 2)
    res = {
        "shape": f"forward NTT 2^{LOG_N} x {COLS}, 2 launches of the tile kernel per transform",
        "calibration": {"fetch_bytes_per_count": cal_f, "write_bytes_per_count": cal_w, "write_bytes_per_count_fill": cal_w2,
                        "method": "known 8-B/lane streams: k_batch<add> (16N read, 8N written), k_fill_random (8N written)"},
        "fetch_counts": ntt_f, "write_counts": ntt_w,
        "hbm_bytes_per_launch": per_launch,
        "hbm_bytes_per_transform": per_transform,
        "algorithmic_bytes_per_transform": 16.0 * N,
        "ratio_to_algorithmic": per_transform / (16.0 * N),
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])


if __name__ == "__main__":
    main(*sys.argv[1:4])

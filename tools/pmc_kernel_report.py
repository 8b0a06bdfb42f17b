#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: sums of every counter and the
derived issue-utilisation figures (SQ counters are in quad-cycles summed over all waves/SIMDs).
usage: pmc_kernel_report.py <dir> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(k):
    k = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", ""))
    return re.sub(r"^void ", "", k).strip()


def main(d, out):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    seen_in = defaultdict(set)  # a counter collected in several passes (files) is averaged, not added up
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen_in[(k, r["Counter_Name"])].add(f)
            launches[k].add(r["Dispatch_Id"])
    for (k, n), files in seen_in.items():
        agg[k][n] /= len(files)
    res = {}
    for k, c in agg.items():
        e = {"launches": len(launches[k]), **{n: v for n, v in sorted(c.items())}}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if n in c:
                    e[n + "/WAVE_CYCLES"] = round(c[n] / wc, 4)
        if c.get("SQ_BUSY_CU_CYCLES") and "SQ_ACTIVE_INST_VALU" in c:
            # one CU = 4 SIMDs; a SIMD issues one VALU instruction at a time
            e["valu_issue_util_per_simd"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"], 4)
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:  # (do not pipe into head: BrokenPipe)
        print(k, {n: v for n, v in e.items() if "/" in n or n.startswith("valu") or n == "launches"})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

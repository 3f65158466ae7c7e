#!/usr/bin/env python3
"""Condenses rocprofv3 output of `bench.py` into the summaries kept in this directory.

  summarize.py stats <run_kernel_stats.csv> <out.csv>        keeps the library's kernels, adds per-launch averages
  summarize.py pmc <fetch counter_collection.csv> <write counter_collection.csv> <frames_per_launch> <out.json>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte
requests as 64 bytes, so wide coalesced reads show half their bytes: `traffic_bytes` = 2 x FETCH + WRITE.  For
dword-per-lane access patterns the factor lies between 1 and 2 (the gray kernel, whose input is read exactly once,
shows 1.58), so the corrected figure is an upper estimate for those kernels; both raw values are kept.
"""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    return name.split("::")[-1].split("<")[0]


def main():
    if sys.argv[1] == "stats":
        rows = list(csv.DictReader(open(sys.argv[2])))
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "average_us", "total_ms", "percent_of_gpu_time"])
            for r in rows:
                if "orbx" in r["Name"] or "anonymous namespace" in r["Name"] and "at::" not in r["Name"]:
                    w.writerow([short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e3:.2f}",
                                f"{float(r['TotalDurationNs']) / 1e6:.3f}", r["Percentage"]])
    else:
        out = collections.OrderedDict()
        for key, path in (("fetch_kib", sys.argv[2]), ("write_kib", sys.argv[3])):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if ("orbx" in k or "anonymous namespace" in k) and "at::" not in k:
                    agg[short(k)].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out.setdefault(k, {})[key] = sum(v) / len(v)
                out[k]["launches_" + key.split("_")[0]] = len(v)
        for k, d in out.items():
            d["traffic_bytes"] = (2 * d.get("fetch_kib", 0.0) + d.get("write_kib", 0.0)) * 1024
            d["frames_per_launch"] = int(sys.argv[4])
        json.dump(out, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()

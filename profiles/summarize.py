#!/usr/bin/env python3
"""Condenses rocprofv3 output of `bench.py` into the summaries kept in this directory.

  summarize.py stats <run_kernel_stats.csv> <out.csv>        keeps the library's kernels, adds per-launch averages
  summarize.py pmc <fetch counter_collection.csv> <write counter_collection.csv> <frames_per_launch> <out.json>
  summarize.py sq <counter_collection.csv> <out.json>        per-launch averages of whatever SQ_* counters the pass held

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


def sq_summary(path, out_path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if ("orbx" in k or "anonymous namespace" in k) and "at::" not in k:
            agg[short(k)][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[short(k)].add(r["Dispatch_Id"])
    out = collections.OrderedDict()
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0) / max(len(launches[k]), 1)):
        n = max(len(launches[k]), 1)
        out[k] = collections.OrderedDict(launches=n)
        for c in sorted(agg[k]):
            out[k][c + "_per_launch"] = round(agg[k][c] / n, 1)
    json.dump(out, open(out_path, "w"), indent=1)


def main():
    if sys.argv[1] == "sq":
        return sq_summary(sys.argv[2], sys.argv[3])
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

#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV directory: per kernel name, launches and the mean of each counter."""
import csv, glob, os, sys, collections


def table(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "?")
            k = k.split("(")[0][:70]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%-72s %-22s %8s %16s %16s" % ("kernel", "counter", "launches", "mean", "sum"))
    for k in sorted(rows):
        for c in sorted(rows[k]):
            v = rows[k][c]
            print("%-72s %-22s %8d %16.1f %16.1f" % (k, c, len(v), sum(v) / len(v), sum(v)))


def traffic_json(fetch_dir, write_dir, out_path, batch, note):
    """profiles/<tag>_traffic.json: per-kernel HBM bytes per launch from the two PMC passes."""
    import json

    def load(d, counter):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}   # KiB -> bytes, mean per launch

    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"batch": batch, "note": note, "kernels": {k: {"fetch_bytes": fe.get(k, 0.0), "write_bytes": wr.get(k, 0.0)} for k in sorted(set(fe) | set(wr))}}
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6] if len(sys.argv) > 6 else "")
    else:
        table(sys.argv[1])

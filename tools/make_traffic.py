#!/usr/bin/env python3
"""profiles/r02_traffic.json from the three PMC passes of tools/sessions/gpu_session2.sh (DO_PMC=1): per kernel and launch the
HBM bytes (FETCH_SIZE, WRITE_SIZE: KiB units, separate passes) and the VALU wave-instructions (SQ_INSTS_VALU).
usage: python tools/make_traffic.py gpurun_out/<tag> <batch> [out.json]"""
import collections, csv, glob, json, os, sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    base, batch = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02_traffic.json")
    fe = load(os.path.join(base, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    wr = load(os.path.join(base, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    va = load(os.path.join(base, "pmc_SQ_WAVES"), "SQ_INSTS_VALU")
    ker = {}
    for k in sorted(set(fe) | set(wr) | set(va)):
        ker[k] = {"fetch_bytes": fe.get(k, 0.0) * 1024.0, "write_bytes": wr.get(k, 0.0) * 1024.0, "valu_wave_insts": va.get(k)}
    json.dump({"batch": batch, "clock_hz": 2.4e9,
               "note": "rocprofv3 --pmc, one pass per counter group (FETCH_SIZE | WRITE_SIZE | SQ_*), bench.py --steps 1 --warmup 0 "
                       "--batch %d; mean per launch; FETCH/WRITE in bytes (raw counters are KiB)" % batch,
               "kernels": ker}, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()

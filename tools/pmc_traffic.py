#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of `bench.py` into
profiles/<round>/pmc_traffic.json, which bench.py reads for roofline.traffic.

  python tools/pmc_traffic.py <key> <fetch_csv> <write_csv> [out_json]

key = "<params>_<gates>_<mode>", e.g. default128_65536_fft. FETCH_SIZE is in KiB and, on gfx950,
reports exactly half the bytes of wide coalesced 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM
section), so it is doubled; WRITE_SIZE is in KiB. Both sit on the L2's fabric side, i.e. they also
count Infinity-Cache hits.
"""
import csv
import json
import os
import sys


def kernel_counter(path, name):
    """First dispatch of the dominant kernel (profile with `bench.py --steps 1 --warmup 0 --cpu-sample 0
    --no-exact-check` so that it is the only blind-rotation launch)."""
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == name and "blind_rotate" in row["Kernel_Name"]:
            return (row["Kernel_Name"].split("<")[0].split("::")[-1].replace("void ", ""), float(row["Counter_Value"]))
    raise SystemExit("no blind_rotate dispatch in " + path)


def main():
    key, fetch_csv, write_csv = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "r01", "pmc_traffic.json")
    doc = json.load(open(out)) if os.path.exists(out) else {}
    kname, fetch = kernel_counter(fetch_csv, "FETCH_SIZE")
    _, write = kernel_counter(write_csv, "WRITE_SIZE")
    doc[key] = {"kernel": kname, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                "traffic_bytes": int(2 * fetch * 1024 + write * 1024)}
    json.dump(doc, open(out, "w"), indent=1)
    print(key, doc[key])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch.

    python tools/pmc_summary.py <dir-with-csvs> [more dirs] > profiles/rNN_pmc_summary.json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streaming reads, i.e. HALF the bytes (MI355X_MICROARCH.md, HBM section) -- the "bytes" fields below
apply that x2 correction to FETCH_SIZE and none to WRITE_SIZE (uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
            with open(f, newline="") as fh:
                rd = csv.DictReader(fh)
                cols = {c.lower(): c for c in rd.fieldnames or []}
                kn, cn, cv = cols.get("kernel_name"), cols.get("counter_name"), cols.get("counter_value")
                if not (kn and cn and cv):
                    continue
                for row in rd:
                    name = row[kn].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                    a = acc[name][row[cn]]
                    a[0] += float(row[cv])
                    a[1] += 1
    out = {}
    for name, ctrs in acc.items():
        e = {"dispatches": max(v[1] for v in ctrs.values())}
        for c, (s, n) in ctrs.items():
            e[c + "_mean"] = s / n
        if "FETCH_SIZE" in ctrs:
            e["read_bytes_per_launch"] = ctrs["FETCH_SIZE"][0] / ctrs["FETCH_SIZE"][1] * 1024 * 2
        if "WRITE_SIZE" in ctrs:
            e["write_bytes_per_launch"] = ctrs["WRITE_SIZE"][0] / ctrs["WRITE_SIZE"][1] * 1024
        out[name] = e
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/pmc_conv_bytes_per_launch.json (the `roofline.traffic` figures bench.py prints) from the PMC summaries:
dispatch-weighted mean of read_bytes_per_launch + write_bytes_per_launch over the conv kernels (and over the wgrad kernels).

    python tools/derive_traffic.py profiles/r01_pmc_summary_bf16_train.json bf16_train [more pairs...] > profiles/pmc_conv_bytes_per_launch.json"""
import json
import sys


def weighted(d, pred):
    n = sum(v["dispatches"] for k, v in d.items() if pred(k))
    b = sum(v["dispatches"] * (v.get("read_bytes_per_launch", 0.0) + v.get("write_bytes_per_launch", 0.0)) for k, v in d.items() if pred(k))
    return (int(b / n) if n else None), n


def main():
    out = {}
    for f, key in zip(sys.argv[1::2], sys.argv[2::2]):
        d = json.load(open(f))
        conv = lambda k: k.startswith("conv_igemm") or k.startswith("conv_streamk") or k.startswith("conv3x3_c64") or k.startswith("stem_direct") or k.startswith("pw_sums")          # noqa: E731
        wg = lambda k: k.startswith("wgrad_bf16_kernel") or k.startswith("wgrad_kernel") or k.startswith("wgrad_bf16_p4_kernel")     # noqa: E731
        out[key], n = weighted(d, conv)
        out["source_" + key] = "%s: dispatch-weighted (2*FETCH_SIZE + WRITE_SIZE)*1024 over %d conv_igemm* / conv_streamk / conv3x3_c64 / stem_direct / pw_sums dispatches (separate --pmc passes, gfx950 x2 FETCH correction)" % (f, n)
        w, nw = weighted(d, wg)
        if w:
            out[key + "_wgrad"] = w
            out["source_" + key + "_wgrad"] = "%s: same over %d wgrad dispatches (reads + partial-slab writes)" % (f, nw)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

"""Sum rocprofv3 --pmc SQ counters per kernel from a rocpd sqlite database (or several) and print ratios.

usage: python tools/sq_summary.py out.json db1 [db2 ...]
"""
import json
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"::(\w+)(<[^>]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    agg = {}
    for f in dbs:
        db = sqlite3.connect(f)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = "counters_collection" if "counters_collection" in tabs else None
        if not view:
            print("no counters_collection view in", f, [t for t in tabs if "pmc" in t or "counter" in t])
            continue
        cols = [d[1] for d in cur.execute("pragma table_info(%s)" % view)]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for name, cname, val in cur.execute("select %s, counter_name, sum(value) from %s group by 1, 2" % (kcol, view)):
            agg.setdefault(short(name), {})[cname] = agg.get(short(name), {}).get(cname, 0.0) + float(val)
    # kernel durations from the same databases (--kernel-trace): GRBM_GUI_ACTIVE / 8 XCDs / duration = the shader clock it ran at
    for f in dbs:
        db = sqlite3.connect(f)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "kernels" in tabs:
            cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
            if "start" in cols and "end" in cols and "name" in cols:
                for name, dur, n in cur.execute("select name, sum(end - start), count(*) from kernels group by 1"):
                    d = agg.setdefault(short(name), {})
                    d["duration_ns"] = d.get("duration_ns", 0.0) + float(dur)
                    d["dispatches"] = d.get("dispatches", 0) + int(n)
    for k, d in agg.items():
        if d.get("GRBM_GUI_ACTIVE") and d.get("duration_ns"):
            d["shader_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["duration_ns"]
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
                if c in d:
                    d[c + "/wave_cycles"] = d[c] / wc
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and d.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs (4 per CU)
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if d.get("SQ_LDS_BANK_CONFLICT") is not None and d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    json.dump(agg, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:12]:
        print(k, json.dumps({a: round(b, 4) if b < 100 else b for a, b in agg[k].items()}))


if __name__ == "__main__":
    main()

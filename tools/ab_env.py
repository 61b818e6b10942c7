#!/usr/bin/env python3
"""A/B of one environment switch on the timed training step: runs bench.py (headline config unless extra flags are given, no comparators)
once per value and repetition, ALTERNATING the values so that box drift hits all of them alike, and prints ms/step per run and the medians.

    python tools/ab_env.py fuse_bnwg 0 7 15 [--reps 2] [-- --depth 101 --frames 16 --clips 16]

The switch is a name of the MVF_POLICY table (mvfnet_amd/policy.py, csrc/common.h): each run gets MVF_POLICY="<name>=<value>" (appended to
an MVF_POLICY already in the environment)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    reps = 2
    if "--reps" in argv:
        i = argv.index("--reps")
        reps = int(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    var, values = argv[0], argv[1:]
    res = {v: [] for v in values}
    groups = {v: None for v in values}
    for r in range(reps):
        for v in values:
            base = os.environ.get("MVF_POLICY", "")
            if var.startswith("env:"):         # a plain environment variable instead (env:MVF_LIB_PATH <lib A> <lib B>: two builds of the library)
                env = dict(os.environ, **{var[4:]: v})
            else:
                env = dict(os.environ, MVF_POLICY=(base + "," if base else "") + "%s=%s" % (var.lower(), v))
            cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs", "--no-eager-compare"] + extra
            p = subprocess.run(cmd, capture_output=True, text=True, env=env)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not line:
                print("%s=%s FAILED: %s" % (var, v, (p.stderr or p.stdout)[-400:]))
                continue
            d = json.loads(line[-1])
            res[v].append(d["ms_per_step"])
            rf = d.get("roofline", {})
            groups[v] = {k: rf.get(k, {}).get("ms_per_step") for k in ("wgrad", "bn", "bn_wgrad", "mvf")}
            groups[v]["conv"] = rf.get("ms_per_step")
            print("%s=%s run %d: %.3f ms/step  %.1f clips/s" % (var, os.path.basename(v), r, d["ms_per_step"], d["value"]), flush=True)
    for v in values:
        xs = sorted(res[v])
        if xs:
            print("%s=%-4s median %.3f ms  (min %.3f, %d runs)  groups alone: %s" % (var, v, xs[len(xs) // 2], xs[0], len(xs), groups[v]))


if __name__ == "__main__":
    main()

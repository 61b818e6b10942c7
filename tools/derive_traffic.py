#!/usr/bin/env python3
"""profiles/pmc_traffic_per_step.json (the `roofline.*.traffic_per_step` / `roofline.step.counter_bytes` figures bench.py prints) from
the PMC summaries tools/pmc_summary.py writes (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, gfx950 x2 FETCH correction
already applied there): HBM bytes PER STEP of every kernel group = sum over the group's kernels of dispatches x (read + write bytes per
dispatch) / step-equivalents of the profiled run (= dispatches of a once-per-step kernel: the stem's forward conv).

    python tools/derive_traffic.py profiles/r04_pmc_summary_bf16_train.json bf16_train [more pairs...] > profiles/pmc_traffic_per_step.json

Per STEP on purpose: a strided data gradient is 4 dispatches of one launch and bench.py brackets launches, so per-dispatch counter means
cannot be divided by per-launch algorithmic bytes (round 3's "0.93" did exactly that)."""
import json
import sys

GROUPS = (
    ("conv", ("conv_igemm", "conv_streamk", "conv3x3_c64", "stem_direct", "pw_sums", "pw_bwd_fused")),
    ("bn_wgrad", ("bnbwd_wgrad",)),                             # [r4] BatchNorm backward apply + pointwise weight gradient in one pass
    ("wgrad", ("wgrad_", "wgrad3x3")),                                   # GEMMs + wgrad_reduce (the fp32 partial slabs are real traffic)
    ("bn", ("bn_apply", "bn_bwd_apply", "bn_bwd_reduce", "bn_stats_kernel")),
    ("bn_finalize", ("bn_stats_finalize", "bn_bwd_finalize")),
    ("mvf", ("mvf_nhwc_apply", "mvf_nhwc_stencil")),
    ("mvf_tapgrad", ("mvf_nhwc_tapgrad",)),
)
ONCE_PER_STEP = ("stem_prep_kernel", "head_clipmean_kernel", "maxpool_bn_fwd")


def group_of(name):
    for g, prefixes in GROUPS:
        if any(name.startswith(p) for p in prefixes):
            return g
    return "other"


def per_step(d):
    steps = None
    for probe in ONCE_PER_STEP:
        n = sum(v["dispatches"] for k, v in d.items() if k.startswith(probe))
        if n:
            steps = n
            break
    if not steps:
        raise SystemExit("no once-per-step kernel found: cannot tell how many steps the profiled run made")
    by, disp = {}, {}
    for k, v in d.items():
        if k.startswith("at::") or k.startswith("__amd") or "spin_kernel" in k:
            g = "torch_and_copies"
        else:
            g = group_of(k)
        b = v["dispatches"] * (v.get("read_bytes_per_launch", 0.0) + v.get("write_bytes_per_launch", 0.0))
        by[g] = by.get(g, 0.0) + b
        disp[g] = disp.get(g, 0) + v["dispatches"]
    out = {g: int(b / steps) for g, b in by.items()}
    out["step_total"] = int(sum(b for g, b in by.items() if g != "torch_and_copies") / steps)
    return steps, out, {g: round(n / steps, 1) for g, n in disp.items()}


def main():
    res = {}
    for f, key in zip(sys.argv[1::2], sys.argv[2::2]):
        steps, by, disp = per_step(json.load(open(f)))
        res[key] = {"bytes_per_step": by, "dispatches_per_step": disp, "step_equivalents": steps,
                    "source": "%s: sum of dispatches x (2*FETCH_SIZE + WRITE_SIZE)*1024 per kernel group / %d step-equivalents (separate --pmc passes, "
                              "gfx950 x2 FETCH correction; wgrad includes wgrad_reduce)" % (f, steps)}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

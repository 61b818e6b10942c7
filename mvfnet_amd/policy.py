"""MVF_POLICY -- the one environment variable of this package's switches.

    MVF_POLICY="name=value,name=value"      (integers; read once at import here, per launch plan / per call in libmvfnet_hip.so: csrc/common.h mvf_policy_int)

The defaults at the call sites ARE the measured policy; DESIGN.md section 4.5 lists every name with the measurement behind its default.  An override is for
A/B runs (tools/ab_env.py) and for the tests that drive both sides of a switch in a child process (tests/helpers.py policy_env)."""
import os


def parse(text):
    out = {}
    for item in (text or "").replace(";", ",").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip().lower()] = int(v)
    return out


_POLICY = parse(os.environ.get("MVF_POLICY", ""))
KNOWN = set()          # the names the Python side has asked for (the C side's are the mvf_policy_int call sites)


def policy(name, default):
    KNOWN.add(name)
    return _POLICY.get(name, default)

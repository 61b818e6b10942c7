"""Build libmvfnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mvfnet_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libmvfnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: left to itself hipcc SLP-packs adjacent scalar fp32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32; on gfx950 packed
# fp32 VALU is slower than the scalar pair wherever matrix instructions share the SIMD (MI355X_MICROARCH.md: "an anti-lever beside MFMAs, including when the
# compiler SLP-packs ... under plain -O3").  Measured on the whole library ([r4], alternating runs on one box): bf16 train step 20.53 -> 20.34 ms, bf16 inference
# +0.7 %, R101 16x4 +1.5 %, fp32 unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(HERE, "csrc")] + os.environ.get("MVF_HIPCC_EXTRA", "").split()


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    objs = []
    procs = []
    hdrs = glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h"))
    flag_file = os.path.join(HERE, "csrc", ".build_flags")
    flags_now = " ".join(FLAGS)
    same_flags = os.path.exists(flag_file) and open(flag_file).read() == flags_now
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        # per-object incremental build: an object is reused when it is newer than its source and every header, built with the same flags
        if not force and same_flags and os.path.exists(obj) and all(os.path.getmtime(d) < os.path.getmtime(obj) for d in [src] + hdrs):
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out.strip():
            print(out.decode())
    with open(flag_file, "w") as f:
        f.write(flags_now)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

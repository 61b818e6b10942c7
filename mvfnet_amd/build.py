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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function",
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
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

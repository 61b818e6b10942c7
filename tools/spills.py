#!/usr/bin/env python3
"""Register budget of every kernel in the library: carves the gfx950 code object out of each csrc/*.o (clang offload bundle inside the .hip_fatbin section) and
prints the kernels that spill registers or use scratch (llvm-readelf --notes: .vgpr_spill_count, .private_segment_fixed_size).

    python tools/spills.py [all]        # 'all' = every kernel with its VGPR / AGPR / LDS figures
"""
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "/usr/bin/c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    blob = open(path, "rb").read()
    at = blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + 1)


def main():
    show_all = len(sys.argv) > 1 and sys.argv[1] == "all"
    for obj in sorted(glob.glob(os.path.join(REPO, "mvfnet_amd", "csrc", "*.o"))):
        for co in code_objects(obj):
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(co)
                f.flush()
                notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
            for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1))  # noqa: E731
                name = re.search(r"\.name:\s+(\S+)", e).group(1)
                agpr, vgpr, spill, scratch, lds = int(re.match(r"\s*(\d+)", e).group(1)), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")
                if show_all or spill or scratch:
                    dem = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
                    print("%-22s vgpr %3d agpr %3d spill %3d scratch %5d B lds %6d  %s" % (os.path.basename(obj), vgpr, agpr, spill, scratch, lds, dem[:150]))


if __name__ == "__main__":
    main()

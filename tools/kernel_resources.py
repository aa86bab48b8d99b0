"""VGPR / SGPR / LDS / scratch of every kernel in the built library (llvm-readelf --notes on the
gfx950 code object), optionally filtered by substrings of the demangled name."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "boxer_amd",
                  "libboxattn_hip.so")
tmp = tempfile.mkdtemp()
fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so], check=True)
subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, k).group(1)
    rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                 g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
for (n, v, s, lds, sc), dn in zip(rows, names):
    if not sys.argv[1:] or any(t in dn for t in sys.argv[1:]):
        print("%-110s vgpr %3s sgpr %3s lds %6s scratch %s" % (dn[:110], v, s, lds, sc))

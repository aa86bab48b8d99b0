"""VGPR / SGPR / LDS / scratch of every kernel of the library: the per-translation-unit objects under
boxer_amd/_build (default), or the object files / libraries given as arguments (llvm-readelf --notes on the
gfx950 code object of each), optionally filtered by substrings of the demangled name.

    python tools/kernel_resources.py [file.o ...] [name substring ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(path, tmp):
    """The gfx950 code objects inside `path`: one per translation unit -- a linked library's .hip_fatbin section is
    the concatenation of its objects' offload bundles (each starts with the bundler's magic string; the unbundler
    itself only reads the first one)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] or [0]
    out = []
    for i, a in enumerate(starts):
        part = os.path.join(tmp, "fat%d.bin" % i)
        with open(part, "wb") as fh:
            fh.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, "gfx950_%d.co" % i)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        out.append(co)
    return out


def kernels(path):
    tmp = tempfile.mkdtemp()
    notes = "".join(subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                    for co in code_objects(path, tmp))
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, k).group(1)
        rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                     g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
    return [(dn,) + r[1:] for r, dn in zip(rows, names)]


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith((".o", ".so"))]
    pats = [a for a in args if a not in files]
    if not files:
        files = sorted(glob.glob(os.path.join(ROOT, "boxer_amd", "_build", "libboxattn_hip.so.*.o")))
    for f in files:
        for dn, v, s, lds, sc in kernels(f):
            if not pats or any(t in dn for t in pats):
                short = re.sub(r"\(.*", "", dn.replace("void boxattn::", ""))
                print("%-70s vgpr %3s sgpr %3s lds %6s scratch %s" % (short[:70], v, s, lds, sc))


if __name__ == "__main__":
    main()

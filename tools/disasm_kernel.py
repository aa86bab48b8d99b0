"""Disassemble one kernel of the built library:  python tools/disasm_kernel.py <substring of the mangled name> [library.so]
-> the kernel's ISA on stdout (llvm-objdump -d of the gfx950 code object that holds it)."""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_guard import LLVM, code_objects

want = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "boxer_amd", "libboxattn_hip.so")
with tempfile.TemporaryDirectory() as tmp:
    for co in code_objects(lib, tmp):
        asm = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co], capture_output=True,
                             text=True, check=True).stdout
        on = False
        for line in asm.splitlines():
            if line.endswith(">:"):
                on = want in line
            if on:
                print(line)

#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a hipcc object file, read from the code object's metadata notes.

    python scripts/kernel_regs.py generative-recommenders_amd/csrc/build/block.o [name-substring]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj = os.path.abspath(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)                   # llvm-objdump writes the extracted code objects next to its input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, capture_output=True)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not dev:
            raise SystemExit("no gfx950 code object found in " + obj)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, dev[0])], check=True, capture_output=True,
                               text=True).stdout
    print(f"{'vgpr':>5} {'agpr':>5} {'spill':>6} {'scratch B':>9} {'lds B':>7}  kernel")
    for blk in notes.split("  - .agpr_count")[1:]:
        def num(key):
            m = re.search(r"\.%s:\s*(\d+)" % key, blk)
            return int(m.group(1)) if m else -1
        name = re.search(r"\n\s*\.name:\s*(\S+)", blk).group(1)
        if want not in name:
            continue
        agpr = int(re.match(r":\s*(\d+)", blk).group(1))
        name = name.replace("_ZN12_GLOBAL__N_1", "").replace("ENS_11BlockParamsE", "")
        print(f"{num('vgpr_count'):5d} {agpr:5d} {num('vgpr_spill_count'):6d} {num('private_segment_fixed_size'):9d} "
              f"{num('group_segment_fixed_size'):7d}  {name[:90]}")


if __name__ == "__main__":
    main()

"""Register and LDS use of the gfx950 kernels in an object file or the built library, from the code objects' metadata notes.

    python tools/kernel_regs.py ust-run_amd/csrc/build/b/conv_halo_bf16.o [name filter]
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_store_hazard import OBJDUMP, code_objects  # noqa: E402

READELF = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
FILT = "c++filt"


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            t = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in t.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            rows.append((g("name"), blk.split()[0], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size")))
    names = subprocess.run([FILT], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("agpr vgpr spill sgpr  lds scratch  kernel")
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "")
        if pat in n:
            print(f"{r[1]:>4} {r[2]:>4} {r[3]:>5} {r[4]:>4} {r[5]:>5} {r[6]:>6}  {n[:150]}")


if __name__ == "__main__":
    main()

"""The built library must not contain the one wide-store form whose wait states hipcc does not insert on gfx950
(tools/check_store_hazard.py: a > 64-bit buffer store with a REGISTER soffset followed by a VALU write of its data registers --
found in round 4 as 6e-6 corrupted outputs of the streaming first convolution).  Runs on the CPU: the gfx950 code objects are
pulled out of libustrun.so and disassembled with llvm-objdump."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_unpadded_wide_store_with_register_soffset():
    lib = os.path.join(ROOT, "ust-run_amd", "ustrun", "libustrun.so")
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_store_hazard.py"), lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gfx950 code objects" in r.stdout and " 0 followed by" in r.stdout, r.stdout

#!/usr/bin/env python3
"""ISA check of the built library for a hazard hipcc does not cover on gfx950 (found in round 4, conv_first.hip):

    buffer_store_dwordx3/x4 vDATA, vOFF, sRSRC, sSOFF offen      <- > 64 bits of data AND a register soffset
    v_...  vDATA[i], ...                                          <- a VALU write of a data register 1-2 instructions later

LLVM's hazard recognizer inserts the wait states for wide stores only when soffset is NOT a register (an SI-era rule); with the
store pipe saturated the gfx950 store then ships the overwritten dword (lanes 12-15 of every 16: 6e-6 of the outputs of the
no-statistics build of the streaming first convolution, none once the row offset moved into the vector offset).  Sources avoid
the form (row offsets ride in voffset, soffset = 0, and LLVM pads by itself); this tool checks that no kernel of the built
library contains it: it pulls the gfx950 code objects out of libustrun.so (clang offload bundles in .hip_fatbin), disassembles
them with llvm-objdump and scans every wide buffer store.

    python tools/check_store_hazard.py [path/to/libustrun.so]          exit status 1 if a site is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WINDOW = 2          # wait states hipcc itself leaves behind the immediate-soffset form on gfx940+


def code_objects(path):
    blob = open(path, "rb").read()
    pos, out = 0, []
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(text):
    """-> (wide buffer stores with a register soffset, offending sites)"""
    lines = []
    kern = "?"
    for l in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            lines.append(("K", m.group(1)))
            continue
        m = re.match(r"^\s+(\S.*?)\s*//", l)
        if m:
            lines.append(("I", m.group(1).strip()))
    total, bad = 0, []
    for i, (kind, t) in enumerate(lines):
        if kind == "K":
            kern = t
            continue
        m = re.match(r"buffer_store_dwordx[34]\s+(.*)", t)
        if not m:
            continue
        ops = [o.strip() for o in m.group(1).split(",")]
        soff = ops[3].split()[0] if len(ops) > 3 else ""
        if not re.match(r"s\d+|s\[|m0|ttmp", soff):
            continue
        total += 1
        data, k, j = regs(ops[0]), 0, i + 1
        while k < WINDOW and j < len(lines):
            kind2, u = lines[j]
            j += 1
            if kind2 == "K":
                break
            if u.startswith("s_nop"):
                k += int(u.split()[1], 0) + 1
                continue
            k += 1
            if u.startswith("v_") and not u.startswith("v_mfma") and regs(u.split(None, 1)[1].split(",")[0].strip()) & data:
                bad.append((kern, t, u))
                break
    return total, bad


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "ust-run_amd", "ustrun", "libustrun.so")
    objs = code_objects(lib)
    if not objs:
        print("no gfx950 code objects found in", lib)
        return 2
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as d:
        for n, o in enumerate(objs):
            p = os.path.join(d, f"co{n}.o")
            open(p, "wb").write(o)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", p], capture_output=True, text=True).stdout
            t, b = scan(text)
            total += t
            bad += b
    print(f"{len(objs)} gfx950 code objects, {total} wide buffer stores with a register soffset, {len(bad)} followed by a VALU write of "
          f"their data within {WINDOW} wait states")
    for kern, st, ov in bad[:20]:
        print(f"  {kern[:100]}: {st}  <-  {ov}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

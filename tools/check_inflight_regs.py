#!/usr/bin/env python3
"""ISA check for the hand-counted register loads (conv_halo_bf16.hip / convT_bf16.hip load weight fragments with
inline-asm `buffer_load_dwordx4` and wait with hand-written `s_waitcnt vmcnt(N)`; hipcc does not know these loads, so nothing
but the exact tests would notice an instruction that reads a destination register while its load is still in flight -- for
instance a register copy the allocator chose to place ahead of the wait).

The tool replays every kernel of the built library's gfx950 code objects in program-text order with a model of the VMEM counter
(loads, stores, LDS-DMA and atomics retire in issue order; `s_waitcnt vmcnt(N)` leaves the N youngest in flight) and reports any
instruction that READS or WRITES a VGPR that a still-outstanding HAND-COUNTED register load will write (the inline-asm groups open
with `s_nop 4`, which is how the replay tells them from the loads hipcc issued and waits for itself).  Straight-line replay is exact inside
a basic block and along fall-through edges (state is carried along the text; a back edge re-enters with the state at the loop's
end, which the second pass over the kernel covers); it starts from an empty queue behind an unconditional branch, so a hazard
that exists only along a taken jump is not seen -- the exact-integer GPU tests remain the functional check.

    python tools/check_inflight_regs.py [path/to/libustrun.so]            exit status 1 if a violation is found
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_store_hazard import OBJDUMP, code_objects  # noqa: E402

VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_|tbuffer_)")


def vregs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip().split()[0] for o in parts[1].split(",") if o.strip()]
    return parts[0], ops


TRACKED = [0]


def replay(kernel, ins_list):
    """-> list of (instruction index, text, offending registers)"""
    out = []
    for _ in range(2):                       # second pass: enters loops with the state their back edge carries
        inflight = []                        # FIFO of (dest regs or empty set)
        marker = 0
        for idx, ins in enumerate(ins_list):
            op, ops = operands(ins)
            m = re.match(r"s_waitcnt\s+(.*)", ins)
            if m:
                vm = re.search(r"vmcnt\((\d+)\)", ins)
                if vm:
                    n = int(vm.group(1))
                    inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
                elif re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", ins):      # raw immediate: treat as a full wait
                    inflight = []
                continue
            if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
                # the next instruction in the text is not reached by falling through: what follows starts from an empty queue
                # (a limitation of the straight-line replay: state does not travel along jumps)
                inflight, marker = [], 0
                continue
            pending = set().union(*inflight) if inflight else set()
            if op == "s_nop" and ops and ops[0] == "4":
                marker = 5                   # the hand-written load groups open with `s_nop 4` and hold up to four loads
                continue
            if VMEM.match(op):
                is_load = "_load" in op and " lds" not in ins
                used = set()
                for o in (ops[1:] if is_load else ops):
                    used |= vregs(o)
                if used & pending:
                    out.append((idx, ins, sorted(used & pending)))
                # every VMEM operation takes a place in the counter; only the HAND-COUNTED register loads carry destination
                # registers here (hipcc waits correctly for the loads it knows about)
                tracked = is_load and marker > 0 and op == "buffer_load_dwordx4"
                inflight.append(vregs(ops[0]) if tracked else set())
                TRACKED[0] += int(tracked)
                if len(inflight) > 64:
                    inflight = inflight[-64:]
                marker -= 1
                continue
            marker = 0
            touched = set()
            for o in ops:
                touched |= vregs(o)
            if touched & pending:
                out.append((idx, ins, sorted(touched & pending)))
    # a violation found by both passes is reported once
    seen, uniq = set(), []
    for v in out:
        if (v[0], v[1]) not in seen:
            seen.add((v[0], v[1]))
            uniq.append(v)
    return uniq


def kernels(text):
    cur, name = [], None
    for l in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            if name:
                yield name, cur
            name, cur = m.group(1), []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//", l)
        if m and name:
            cur.append(m.group(1).strip())
    if name:
        yield name, cur


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "ust-run_amd", "ustrun", "libustrun.so")
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    objs = code_objects(lib)
    nk, bad = 0, []
    with tempfile.TemporaryDirectory() as d:
        for n, o in enumerate(objs):
            p = os.path.join(d, f"co{n}.o")
            open(p, "wb").write(o)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", p], capture_output=True, text=True).stdout
            for name, ins in kernels(text):
                if only and only not in name:
                    continue
                nk += 1
                for idx, t, r in replay(name, ins):
                    bad.append((name, idx, t, r))
    print(f"{len(objs)} gfx950 code objects, {nk} kernels replayed, {TRACKED[0] // 2} hand-counted register loads followed, "
          f"{len(bad)} instructions touch a register with its load still in flight")
    for name, idx, t, r in bad[:30]:
        print(f"  {name[:90]} +{idx}: {t}   [v{r}]")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

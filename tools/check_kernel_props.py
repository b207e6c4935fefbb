"""Two properties of the built gfx950 code that hipcc gives or takes silently (round 4 found both by their cost, not by a failure):

1. NO SCRATCH in the kernels that count their own `s_waitcnt vmcnt(N)` (halo-tiled convolution with register-fed weights, the
   64 -> 64 streaming kernel, ConvTranspose / 1x1 GEMM, the streaming first convolution): a spill reload is a VMEM instruction of its
   own -- it shifts every hand-written count behind it, and hipcc waits vmcnt(0) for it, which drains the weight loads in flight.
   The 16x16x32 build of the halo kernel carried 33 spilled registers (tile coordinates hoisted out of the chunk loop) with a reload
   and a full drain at the top of every chunk.
2. LOADS IN FLIGHT in the small fixed-order reductions (BatchNorm finalize kernels, reduce_rows): their row loop must issue a batch
   of loads before the first wait.  After an index generalisation the unroller left each load of bn_bwd_finalize_kernel waited for
   on its own: 8.6 -> 20 us per launch, 18 launches per step.

    python tools/check_kernel_props.py [libustrun.so]      exit 0 = both hold
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_store_hazard import OBJDUMP, code_objects  # noqa: E402

# (the eight-wave build of the streaming kernel -- conv3x3_ws64x8_kernel, debug flag bit 2, never the default -- spills 8-64
# registers and leaves its waits to the compiler: not in this list)
COUNTED = ("conv3x3_halo_bf16_kernel", "conv3x3_ws64cp_kernel", "conv3x3_ws64_kernel", "convT_bf16_kernel", "conv_first_fwd_stream_kernel",
           "wgradT2_bf16_kernel", "wgrad_halo4_bf16_kernel")
BATCHED = {"bn_bwd_finalize_kernel": 8, "bn_finalize_kernel": 8, "reduce_rows_kernel": 4, "bn_stat_fused_kernel": 8, "bn_stat_stage1_kernel": 8,
           "reduce_plain_kernel": 8, "reduce_groups_kernel": 8}


def kernels(text):
    name, body = None, []
    for l in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name:
            t = l.split("//")[0].strip()
            if t:
                body.append(t)
    if name:
        yield name, body


def findings(name, body):
    """-> (counted kernel?, batched reduction?, list of findings) for one disassembled kernel"""
    out, counted, batched = [], False, False
    if any(k in name for k in COUNTED):
        counted = True
        ns = sum(1 for t in body if t.startswith("scratch_"))
        if ns:
            out.append(f"{name[:110]}: {ns} scratch instructions in a kernel with hand-counted waits")
    for k, need in BATCHED.items():
        if k in name:
            batched = True
            best = run = 0          # longest run of vector loads not interrupted by a wait on the vector-memory counter
            for t in body:
                if t.startswith(("global_load", "buffer_load")):
                    run += 1
                    best = max(best, run)
                elif t.startswith("s_waitcnt") and "vmcnt" in t:
                    run = 0
            if best < need:
                out.append(f"{name[:110]}: at most {best} loads issued between waits (needs {need})")
    return counted, batched, out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "ust-run_amd", "ustrun", "libustrun.so")
    objs = code_objects(lib)
    if not objs:
        print("no gfx950 code objects found in", lib)
        return 2
    bad, ncounted, nbatched = [], 0, 0
    with tempfile.TemporaryDirectory() as d:
        for n, o in enumerate(objs):
            p = os.path.join(d, f"co{n}.o")
            open(p, "wb").write(o)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", p], capture_output=True, text=True).stdout
            for name, body in kernels(text):
                c, b_, f = findings(name, body)
                ncounted += c
                nbatched += b_
                bad += f
    print(f"{len(objs)} gfx950 code objects: {ncounted} kernels with hand-counted waits checked for scratch, {nbatched} fixed-order "
          f"reductions checked for loads in flight, {len(bad)} findings")
    for b in bad[:30]:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

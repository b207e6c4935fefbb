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


def test_hand_counted_register_loads_are_never_touched_in_flight():
    """the weight fragments of conv_halo_bf16.hip / convT_bf16.hip arrive by inline-asm buffer loads that hipcc does
    not know about and are waited for with hand-written vmcnt values; tools/check_inflight_regs.py replays every kernel of the
    built library and fails if any instruction reads or writes a destination register of such a load before the wait that
    retires it.  The replay itself is checked on two synthetic snippets first (one clean, one with a copy ahead of the wait)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_regs as T
    clean = ["s_nop 4", "buffer_load_dwordx4 v[10:13], v1, s[0:3], s4 offen", "buffer_load_dwordx4 v[14:17], v1, s[0:3], s4 offen offset:512",
             "global_load_dwordx4 v[20:23], v[2:3], off", "v_add_u32_e32 v5, v6, v7", "s_waitcnt vmcnt(1)",
             "v_mfma_f32_32x32x16_bf16 v[32:47], v[10:13], v[14:17], v[32:47]"]
    assert T.replay("k", clean) == []
    dirty = clean[:4] + ["v_mov_b32_e32 v30, v12"] + clean[4:]                  # a copy of an in-flight destination ahead of the wait
    hits = T.replay("k", dirty)
    assert len(hits) == 1 and hits[0][2] == [12], hits
    short = clean[:5] + ["s_waitcnt vmcnt(2)"] + clean[6:]                      # the wait leaves the second load in flight
    assert [h[2] for h in T.replay("k", short)] == [[14, 15, 16, 17]]
    lib = os.path.join(ROOT, "ust-run_amd", "ustrun", "libustrun.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py"), lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    m = re.search(r"(\d+) hand-counted register loads followed, 0 instructions", r.stdout)
    assert m and int(m.group(1)) > 1000, r.stdout          # the halo / ConvTranspose kernels really were replayed


def test_counted_kernels_do_not_spill_and_reductions_keep_loads_in_flight():
    """tools/check_kernel_props.py on the built library: no scratch instruction in any kernel that counts its own vmcnt waits (a
    spill reload shifts the counts and drains the queue), and the row loops of the small fixed-order reductions issue a batch of
    loads before their first wait (round 4: both were found by what they cost, 5 % of the input gradients and 0.2 ms per step)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_kernel_props as K
    # the checker itself, on synthetic kernels: a spill in a counted kernel, a reduction whose loads are waited for one by one
    assert K.findings("conv3x3_halo_bf16_kernelIx", ["v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]", "s_endpgm"]) == (True, False, [])
    c, b, f = K.findings("conv3x3_halo_bf16_kernelIx", ["scratch_store_dword off, v1, off", "scratch_load_dword v1, off, off", "s_waitcnt vmcnt(0)"])
    assert c and not b and len(f) == 1 and "2 scratch" in f[0]
    serial = ["global_load_dword v1, v[2:3], off", "s_waitcnt vmcnt(0)", "v_add_f64 v[4:5], v[4:5], v[6:7]"] * 8
    assert len(K.findings("bn_bwd_finalize_kernelILb0EE", serial)[2]) == 1
    batched = ["global_load_dword v1, v[2:3], off"] * 8 + ["s_waitcnt vmcnt(7)"] + ["v_add_f64 v[4:5], v[4:5], v[6:7]"] * 8
    assert K.findings("bn_bwd_finalize_kernelILb0EE", batched) == (False, True, [])
    lib = os.path.join(ROOT, "ust-run_amd", "ustrun", "libustrun.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernel_props.py"), lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    m = re.search(r"(\d+) kernels with hand-counted waits checked for scratch, (\d+) fixed-order reductions .* 0 findings", r.stdout)
    assert m and int(m.group(1)) > 100 and int(m.group(2)) >= 14, r.stdout

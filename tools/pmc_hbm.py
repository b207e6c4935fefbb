"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs of
the same command).  FETCH_SIZE and WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section), so HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE.

    python tools/pmc_hbm.py FETCH_DIR WRITE_DIR OUT_CSV [OUT_JSON [STEPS]]

OUT_JSON (optional) receives the per-launch bytes of the conv class that bench.py reports in roofline.traffic."""
import collections, csv, glob, json, os, re, sys


def load(d, counter):
    tot, cnt, dur = collections.Counter(), collections.Counter(), collections.Counter()
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"([A-Za-z_0-9]+_kernel(<[^>]*>)?)", r["Kernel_Name"])
            k = (m.group(1) if m else r["Kernel_Name"][:60]).replace(",", ";")
            tot[k] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"],)
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return tot, cnt, dur


def main():
    fdir, wdir, out = sys.argv[1:4]
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # steps the profiled command ran in all (warm-up included)
    ft, fc, fd = load(fdir, "FETCH_SIZE")
    wt, wc, wd = load(wdir, "WRITE_SIZE")
    rows = []
    for k in ft:
        n = fc[k]
        if n == 0 or wc.get(k, 0) != n:
            continue
        mb = (2 * ft[k] + wt[k]) * 1024 / n / 1e6
        us = fd[k] / n
        rows.append((2 * ft[k] + wt[k], k, n, ft[k], wt[k], mb, us, mb / us * 1e3 if us > 0 else 0))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,hbm_MB_per_launch(fetch_x2+write),avg_us_under_pmc,hbm_GBps\n")
        for _, k, n, a, b, mb, us, g in rows:
            f.write(f"{k},{n},{a:.1f},{b:.1f},{mb:.2f},{us:.1f},{g:.0f}\n")
    if len(sys.argv) > 4:
        conv = [r for r in rows if any(s in r[1] for s in ("conv3x3_halo_bf16", "conv3x3_ws64", "igemm_bf16", "conv_first_fwd", "convT_bf16"))]
        n = sum(r[2] for r in conv)
        by = sum(r[0] for r in conv) * 1024
        wg = [r for r in rows if any(s in r[1] for s in ("wgrad_halo", "wgradT_bf16", "wgradT2_bf16", "wgrad_bf16_kernel"))]
        nw = sum(r[2] for r in wg)
        bw = sum(r[0] for r in wg) * 1024
        ws = [r for r in rows if "conv3x3_ws64" in r[1]]
        import hashlib
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd", "ustrun", "libustrun.so")
        lib_sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]          # the build these counters were read from
        json.dump({"library_sha16": lib_sha, "counters": "FETCH_SIZE (KB, doubled per MI355X_MICROARCH HBM note) + WRITE_SIZE (KB), rocprofv3 --pmc, "
                               "separate passes, bench.py --steps 1 --warmup 1 --no-profile (2 steps in all)",
                   "kernel_class": "conv (conv3x3_halo_bf16 + conv3x3_ws64 + convT_bf16 + igemm_bf16 + conv_first_fwd)",
                   "launches": n, "hbm_bytes_per_launch": by / max(n, 1),
                   "steps": steps, "hbm_bytes_per_step": by / steps, "wgrad_hbm_bytes_per_step": bw / steps,
                   "wgrad_kernel_class": "weight gradients (wgrad_halo*_bf16 + wgradT_bf16 / wgradT2_bf16 + wgrad_bf16)",
                   "wgrad_launches": nw, "wgrad_hbm_bytes_per_launch": bw / max(nw, 1),
                   "ws64": [{"kernel": r[1], "launches": r[2], "hbm_MB_per_launch": round(r[5], 2)} for r in ws]},
                  open(sys.argv[4], "w"), indent=1)
    for r in rows[:12]:
        print(f"{r[1][:70]:70s} n={r[2]:4d} {r[5]:9.1f} MB/launch {r[6]:8.1f} us {r[7]:7.0f} GB/s")


if __name__ == "__main__":
    main()

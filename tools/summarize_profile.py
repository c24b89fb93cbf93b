#!/usr/bin/env python
"""tools/summarize_profile.py <gpurun_out/prof_TAG> <profiles/NAME> — condense a tools/profile.sh run into
tracked files: NAME_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels + totals) and
NAME_pmc.md (per-kernel PMC averages with the derived numbers DESIGN.md quotes)."""
import collections
import csv
import os
import sys


def main(src, dst, workload="fp64_int8_9, M=8192 N=8192 K=8192, op N/N"):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
    with open(dst + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
        w.writeheader()
        for r in stats:
            w.writerow(r)
    dur = {}
    for r in stats:
        dur[r["Name"]] = float(r["AverageNs"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(os.listdir(src)):
        p = os.path.join(src, d, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# PMC summary (%s)\n" % os.path.basename(src.rstrip("/")),
             "Averages per dispatch; each counter group collected in its own rocprofv3 --pmc pass "
             "(tools/profile.sh).  FETCH_SIZE/WRITE_SIZE are KiB as reported; on gfx950 FETCH_SIZE counts 64 B per "
             "128 B request for wide streaming reads (MI355X_MICROARCH.md §HBM): `fetch_GB_corrected` doubles it.\n"]
    for k in sorted(agg):
        if "ozhip" not in k:
            continue
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        lines.append("## `%s`\n" % k[:110])
        if k in dur:
            lines.append("- average duration (kernel-trace --stats run): %.3f ms" % (dur[k] / 1e6))
        for n in sorted(c):
            lines.append("- %s = %.4g" % (n, c[n]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 1024 SIMDs
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            lines.append("- derived: elapsed shader cycles per XCD = %.4g; MFMA pipe busy = %.1f %% of SIMD-cycles"
                         % (cyc, 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc))
        if "TCC_HIT_sum" in c:
            tot = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
            lines.append("- derived: L2 hit rate = %.1f %% of %.4g requests" % (100.0 * c["TCC_HIT_sum"] / max(tot, 1), tot))
        if "FETCH_SIZE" in c:
            lines.append("- derived: fetch_GB_corrected = %.3f GB (raw %.3f GB)" % (2 * c["FETCH_SIZE"] * 1024 / 1e9,
                                                                               c["FETCH_SIZE"] * 1024 / 1e9))
        if "WRITE_SIZE" in c:
            lines.append("- derived: write_GB = %.3f GB" % (c["WRITE_SIZE"] * 1024 / 1e9))
        lines.append("")
    open(dst + "_pmc.md", "w").write("\n".join(lines))
    # machine-readable HBM traffic of the dominant kernel (bench.py reports it as roofline.traffic)
    import json
    best = None
    for k in agg:
        if "slice_gemm" in k and "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            c = {n: sum(v) / len(v) for n, v in agg[k].items()}
            t = dict(kernel=k.split("(")[0], fetch_bytes_corrected=2 * c["FETCH_SIZE"] * 1024, write_bytes=c["WRITE_SIZE"] * 1024)
            t["hbm_bytes_per_launch"] = t["fetch_bytes_corrected"] + t["write_bytes"]
            t["l2_hit_rate"] = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
            t["workload"] = workload
            t["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh); FETCH_SIZE doubled (gfx950)"
            if best is None or t["hbm_bytes_per_launch"] > best["hbm_bytes_per_launch"]:
                best = t
    if best:
        json.dump(best, open(dst + "_traffic.json", "w"), indent=1)
    print("wrote", dst + "_kernel_stats.csv", dst + "_pmc.md")


if __name__ == "__main__":
    main(*sys.argv[1:4])

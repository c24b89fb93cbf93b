import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "ozhip" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"] or "memset" in r["Kernel_Name"].lower() or "fill" in r["Kernel_Name"].lower()]
# last 8 GEMM calls: print sequence of (name, dur, gap)
seq = rows[-60:]
prev_end = None
for r in seq[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e-s)/1e3:8.1f} us  gap {gap:7.1f} us  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')}")
    prev_end = e

"""tools/time_course.py M N K opa opb [calls] — per-call time of fp64_int8_9 over a long back-to-back run, 32x32x32 tile
(OZIMMU_HIP_K64_TILE=0) and k64 tile (=1) in blocks, with the shader clock / package power rocm-smi shows during each
block: separates the first ~0.3 s of a run (the part's power-averaging window) from the sustained state."""
import os; os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")
import sys, time, subprocess, threading, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ozimmu_amd as oz
m, n, k = (int(x) for x in sys.argv[1:4]); oa, ob = sys.argv[4:6]
calls = int(sys.argv[6]) if len(sys.argv) > 6 else 60
h = oz.create(); oz.set_cuda_stream(h, torch.cuda.current_stream())
a = torch.rand((k, m) if oa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
b = torch.rand((n, k) if ob == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
lda, ldb = a.shape[1], b.shape[1]

def smi_poll(stop, out):
    while not stop.is_set():
        t = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t); p = re.search(r"Package Power \(W\): ([\d.]+)", t)
        if s: out.append((int(s.group(1)), float(p.group(1)) if p else None))

for blk, val in enumerate(("0", "1", "0", "1")):
    os.environ["OZIMMU_HIP_K64_TILE"] = val
    if blk < 2: time.sleep(2.0)             # blocks 0, 1 start from an idle part; 2, 3 follow a loaded one
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
    stop, smp = threading.Event(), []
    th = threading.Thread(target=smi_poll, args=(stop, smp)); th.start()
    ev[0].record()
    for i in range(calls):
        oz.gemm(h, oa, ob, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, "fp64_int8_9")
        ev[i + 1].record()
    torch.cuda.synchronize(); stop.set(); th.join()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(calls)]
    sc = sorted(x[0] for x in smp); pw = [x[1] for x in smp if x[1]]
    print(f"block {blk} K64_TILE={val} {'from idle' if blk < 2 else 'after load'}: calls 1-5 " + " ".join(f"{x:.2f}" for x in ms[:5]) +
          f" | 6-10 mean {sum(ms[5:10]) / 5:.2f} | 11-20 {sum(ms[10:20]) / 10:.2f} | last 20 {sum(ms[-20:]) / 20:.2f} ms"
          f" | sclk median {sc[len(sc) // 2] if sc else None} MHz, power max {max(pw) if pw else None} W ({len(smp)} samples)", flush=True)
oz.destroy(h)

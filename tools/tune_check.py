import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["OZIMMU_HIP_ENV_PER_CALL"] = "1"
import torch
import ozimmu_amd as oz
SLOTS = ["k2", "classic", "wide", "x16", "k64", "k64_breg"]
def run(m, n, k, S, opa, opb, tune, reps, legs=5):
    os.environ["OZIMMU_HIP_AUTOTUNE"] = "1" if tune else "0"
    h = oz.create()
    a = torch.rand((k, m) if opa == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand((n, k) if opb == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
    c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    lda = m if opa == "N" else k
    ldb = k if opb == "N" else n
    mode = f"fp64_int8_{S}"
    ts = []
    for leg in range(legs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            assert oz.gemm(h, opa, opb, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, m, mode) == 0
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / reps * 1e3)
    st = oz.tuner_state(h, mode, m, n, k)
    pred, pick = oz.policy_predict(h, S, m, n, k)
    lk = oz.last_kernel(h)[0]
    oz.destroy(h)
    return ts, st, pick, lk, {kk: round(v) for kk, v in pred.items()}
for (m, n, k, S, opa, opb, reps) in [(32768, 32768, 1024, 9, "N", "T", 4), (4096, 4096, 1024, 9, "N", "N", 40), (2048, 2048, 2048, 9, "N", "N", 60),
                                     (16384, 16384, 128, 9, "N", "N", 10), (1024, 1024, 1024, 9, "N", "N", 200), (1536, 1536, 1536, 9, "N", "N", 100),
                                     (8192, 8192, 1024, 9, "N", "N", 10), (3000, 5000, 512, 6, "N", "N", 60)]:
    for tune in (0, 1, 0, 1):
        ts, st, pick, lk, pred = run(m, n, k, S, opa, opb, tune, reps)
        print(f"{m}x{n}x{k} S={S} {opa}{opb} tune={tune} legs_ms={[round(t,3) for t in ts]} state={st} decided={SLOTS[st[1]] if st[1]>=0 else None} model={pick} last={lk} pred={pred}", flush=True)

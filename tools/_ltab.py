import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import hashmergejoin_amd as H
os.environ["HMJ_LTABLE"] = "0"
e0 = H.Executor(0)
del os.environ["HMJ_LTABLE"]
e1 = H.Executor(0)
def timed(e, R, S, fl, reps=10):
    for _ in range(3): r = e.join_device(R, S, fl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = e.join_device(R, S, fl)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for P in (24, 26, 28):
    for k in (4, 8, 10, 11, 12):
        R, S = e0.gen_build(1 << k), e0.gen_uniform_domain(1 << P, 1 << k)
        row = []
        for fl, name in ((0, "count"), (H.HMJ_CHECKSUM | H.HMJ_SUM_PROBE, "checks"), (H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE, "first")):
            m0, r0 = timed(e0, R, S, fl); c0 = r0.checks()
            m1, r1 = timed(e1, R, S, fl)
            row.append("%s L2 table %.3f | LDS table %.3f ms%s" % (name, m0, m1, "" if c0 == r1.checks() and e1.last_timing()["path"] & H.HMJ_PATH_LDS_TABLE else " MISMATCH"))
        print("nb=2^%d np=2^%d | %s" % (k, P, " | ".join(row)), flush=True)

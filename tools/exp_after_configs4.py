import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hashmergejoin_amd as H
ex = H.Executor(0)
def timed(fn, reps=2):
    fn()
    best=None
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter(); r=fn(); torch.cuda.synchronize()
        dt=(time.perf_counter()-t0)*1e3; best=dt if best is None else min(best,dt)
    return best, r
n = 500000000
R, S = ex.gen_build(n), ex.gen_probe(n, n)
ms, r = timed(lambda: ex.join_device(R, S, 0)); print("fresh 5e8:", ms, hex(ex.last_timing()["path"]), flush=True)
del R, S; torch.cuda.empty_cache()
nb, npb, theta = 1 << 24, 1 << 30, 0.9
w = 1.0 / np.arange(1, nb + 1, dtype=np.float64) ** theta
cdf = np.cumsum(w) / w.sum()
thr = np.empty(nb, np.uint64); big = cdf >= 1.0 - 2.0 ** -53
thr[~big] = (cdf[~big] * 2.0 ** 64).astype(np.uint64); thr[big] = np.uint64((1 << 64) - 1); thr[-1] = np.uint64((1 << 64) - 1)
thr_dev = torch.from_numpy(thr.view(np.int64).copy()).cuda()
Rz = ex.gen_from_cdf(nb, thr_dev); Sz = ex.gen_uniform_domain(npb, nb)
ms, r = timed(lambda: ex.join_device(Rz, Sz, 0)); print("configs4 count:", ms, flush=True)
ms, r = timed(lambda: ex.join_device(Rz, Sz, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE)); print("configs4 first:", ms, flush=True)
del Rz, Sz; torch.cuda.empty_cache()
R, S = ex.gen_build(n), ex.gen_probe(n, n)
ex.set_profiling(True)
for i in range(4):
    torch.cuda.synchronize(); t0=time.perf_counter(); ex.join_device(R, S, 0); torch.cuda.synchronize()
    t = ex.last_timing()
    print("after configs4, 5e8 join %d: %.3f ms path %#x bits %d A %.3f B %.3f probe %.3f" % (i, (time.perf_counter()-t0)*1e3, t["path"], t["radix_bits"], t["ms_scatter_pass0"]/2, t["ms_scatter_pass1"]/2, t["ms_probe_count"]), flush=True)
print(ex.placement_info())

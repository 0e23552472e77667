import os, sys
sys.path.insert(0, os.getcwd())
import torch
import hashmergejoin_amd as H
ex = H.Executor(0)
g = torch.Generator(device="cuda"); g.manual_seed(6)
nd = 1 << 24
mk = lambda: torch.stack([torch.randint(0, nd // 8, (nd,), device="cuda", generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62), torch.arange(nd, device="cuda", dtype=torch.int64)], 1).contiguous()
Rd, Sd = mk(), mk()
ex.set_profiling(True)
for fl, name in ((0, "count"), (H.HMJ_MATERIALIZE, "rows"), (H.HMJ_ORDERED, "ordered")):
    for _ in range(3): r = ex.join_device(Rd, Sd, fl)
    t = ex.last_timing()
    print(name, int(r.n_matches), {k: round(v, 3) if isinstance(v, float) else v for k, v in t.items() if (isinstance(v, float) and v > 0) or k in ("path", "radix_bits", "n_probe_items")}, flush=True)

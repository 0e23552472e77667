#!/usr/bin/env python3
"""One workload per process, for rocprofv3 (tools/profile_cmd.sh <tag> tools/prof_paths.py <name> [reps]):
the join paths that only had wall-clock sweeps.  Prints one JSON line: path bits, ms per call
(wall clock, best of reps), the algorithmic minimum bytes of the call and the fraction of the 8 TB/s peak.

Names: small16_count small16_mat small16_ord small11_count mid20_count dup8_ord fk22_ord sort28 fk24_ord
       small16_ord64 (payloads spanning 64 bits) headline small12_ord (runs of 16384 rows: cut into 16 pieces)
       small10_ord28 (2^10 x 2^28: 256 pieces per run) configs1_26 (2^26 x 2^26 count)
Algorithmic bytes: every input row read once (16 B), every result row written once (24 B; 16 B for the sort)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hashmergejoin_amd as H  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ex = H.Executor(0)


def dup_rel(nd, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.stack([torch.randint(0, nd // 8, (nd,), device="cuda", generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62),
                        torch.arange(nd, device="cuda", dtype=torch.int64)], 1).contiguous()


n26, n28 = 1 << 26, 1 << 28
sort = False
if name.startswith("small16"):
    R, S = ex.gen_build(1 << 16), ex.gen_uniform_domain(n26, 1 << 16)
    if name == "small16_ord64":  # payloads over all 64 bits: the composite sort needs more passes
        S[:, 1] = S[:, 1] * 0x9E3779B97F4A7C15
    fl = {"count": 0, "mat": H.HMJ_MATERIALIZE, "ord": H.HMJ_ORDERED, "ord64": H.HMJ_ORDERED}[name.split("_")[1]]
elif name == "small12_ord":
    R, S, fl = ex.gen_build(1 << 12), ex.gen_uniform_domain(n26, 1 << 12), H.HMJ_ORDERED
elif name == "small10_ord28":
    R, S, fl = ex.gen_build(1 << 10), ex.gen_uniform_domain(n28, 1 << 10), H.HMJ_ORDERED
elif name == "small11_count":
    R, S, fl = ex.gen_build(1 << 11), ex.gen_uniform_domain(n26, 1 << 11), 0
elif name == "mid20_count":
    R, S, fl = ex.gen_build(1 << 20), ex.gen_uniform_domain(n26, 1 << 20), 0
elif name == "dup8_ord":
    R, S, fl = dup_rel(1 << 24, 6), dup_rel(1 << 24, 7), H.HMJ_ORDERED
elif name == "fk22_ord":
    R, S, fl = ex.gen_build(1 << 22), ex.gen_uniform_domain(n28, 1 << 22), H.HMJ_ORDERED
elif name == "fk24_ord":
    R, S, fl = ex.gen_build(1 << 24), ex.gen_uniform_domain(n28, 1 << 24), H.HMJ_ORDERED
elif name == "headline":
    R, S, fl = ex.gen_build(n28), ex.gen_probe(n28, n28), 0
elif name == "configs1_26":  # BASELINE configs[1]'s size under the planner's own bits
    R, S, fl = ex.gen_build(n26), ex.gen_probe(n26, n26), 0
elif name == "sort28":
    R, S, fl, sort = ex.gen_build(n28), None, 0, True
else:
    raise SystemExit("unknown workload " + name)


def call():
    if sort:
        return ex.sort_device(R)
    return ex.join_device(R, S, fl)


r = call()  # warm-up: workspace, plan
best = None
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    best = dt if best is None else min(best, dt)
t = ex.last_timing()
if sort:
    nbytes, rows = 32 * R.shape[0], R.shape[0]
else:
    rows = int(r.n_matches)
    nbytes = 16 * (R.shape[0] + S.shape[0]) + (24 * rows if fl & (H.HMJ_MATERIALIZE | H.HMJ_ORDERED) else 0)
print(json.dumps({"workload": name, "ms": round(best, 4), "result_rows": rows, "algorithmic_bytes": nbytes,
                  "frac_of_8TBps": round(nbytes / (best * 1e-3) / 8e12, 4), "path": "0x%x" % t["path"],
                  "radix_bits": t["radix_bits"], "radix_passes": t["radix_passes"]}), flush=True)
ex.close()

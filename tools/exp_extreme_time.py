#!/usr/bin/env python3
"""Timing of extreme shapes (second run of each mode): looking for performance cliffs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
from validate_random import dev, M63
g = torch.Generator(device=dev); g.manual_seed(7)
r = lambda hi, size: torch.randint(0, hi, (size,), dtype=torch.int64, device=dev, generator=g)
ex = H.Executor(0); ex.set_profiling(True)
u = lambda n: r(M63, n) * 2 + 1
def run(name, kb, kp, modes=("count", "checksum", "first", "materialize", "ordered")):
    B = torch.stack([kb, r(M63, kb.numel())], 1).contiguous(); P = torch.stack([kp, r(M63, kp.numel())], 1).contiguous()
    out = []
    for m in modes:
        fl = {"count": 0, "checksum": H.HMJ_CHECKSUM, "materialize": H.HMJ_MATERIALIZE, "ordered": H.HMJ_ORDERED, "first": H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE}[m]
        for _ in range(2):
            res = ex.join_device(B, P, fl); t = ex.last_timing()
        out.append("%s %.2f ms" % (m, t["ms_total"]))
        ex.release_result()
    print("%-46s nb=%10d np=%10d matches %12d | %s" % (name, kb.numel(), kp.numel(), int(res.n_matches), " | ".join(out)), flush=True)
kb = u(1000); run("tiny build, huge probe (fk)", kb, kb[r(1000, 1 << 27)])
kb = u(1 << 27); run("huge build, tiny probe", kb, kb[r(1 << 27, 1000)])
kb = u(1 << 20); run("all probes carry ONE key", kb, kb[:1].repeat(1 << 24))
kb = u(1 << 24); kb[: 1 << 20] = 77; run("2^20 build rows share one key, probes uniform", kb, u(1 << 24))
kb = u(1 << 24); kb[: 1 << 20] = 77; kp = u(1 << 24); kp[:16] = 77; run("... and 16 probe rows hit it", kb, kp, ("count", "first"))
kb = u(1 << 22); run("no probe row matches", kb, r(M63, 1 << 22) * 2)
kb = torch.arange(1 << 26, dtype=torch.int64, device=dev); run("dense sorted 0..2^26-1 both sides", kb, kb.flip(0))
kb = (torch.arange(1 << 24, dtype=torch.int64, device=dev) << 40); run("keys = i << 40 (low bits zero)", kb, kb[r(1 << 24, 1 << 25)])
kb = (torch.arange(1 << 24, dtype=torch.int64, device=dev) * 0x1111); run("keys = i * 0x1111", kb, kb[r(1 << 24, 1 << 24)])
kb = u(1 << 24); run("sorted probe side", kb, torch.sort(kb[r(1 << 24, 1 << 24)])[0])
kb = u(100000); run("small build (P=32), 2^26 probes on one key", kb, kb[:1].repeat(1 << 26), ("count", "checksum", "first", "materialize"))
kb = u(1 << 22); kp = kb[r(1 << 22, 1 << 26)]; kp[: 1 << 25] = kb[5]; run("2^22 build, half of 2^26 probes on one key", kb, kp)
kb = u(1 << 26); kb[: 1 << 16] = 99; kp = u(1 << 20); kp[:1000] = 99; run("65536 build copies x 1000 probe copies of a key", kb, kp)

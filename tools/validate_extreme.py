#!/usr/bin/env python3
"""Extreme shapes against the torch reference: tiny vs huge sides, all probes on one key, empty-ish sides."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
from validate_random import reference, s64, dev, M63

g = torch.Generator(device=dev); g.manual_seed(7)
r = lambda hi, size: torch.randint(0, hi, (size,), dtype=torch.int64, device=dev, generator=g)
ex = H.Executor(0)
bad = 0
def run(name, kb, kp):
    global bad
    B = torch.stack([kb, r(M63, kb.numel())], 1).contiguous(); P = torch.stack([kp, r(M63, kp.numel())], 1).contiguous()
    want = reference(B, P)
    res = ex.join_device(B, P, 0)
    ok = (int(res.n_matches), s64(int(res.sum_r)), s64(int(res.sum_s))) == (want["n"], want["sum_r"], want["sum_s"])
    f = ex.join_device(B, P, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE)
    ok2 = (int(f.n_matches), s64(int(f.sum_r)), s64(int(f.sum_s)), s64(int(f.sum_probe_all))) == (want["fw_n"], want["fw_sum_r"], want["fw_sum_s"], want["sum_p"])
    ok3 = True
    if want["n"] <= (1 << 27):
        c = ex.join_device(B, P, H.HMJ_CHECKSUM).checks()
        for fl in (H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM, H.HMJ_ORDERED | H.HMJ_CHECKSUM):
            m = ex.join_device(B, P, fl)
            ok3 = ok3 and m.checks() == c and int(m.n_matches) == want["n"]
            ex.release_result()
    print("%-44s nb=%10d np=%10d matches %14d : count %s first-wins %s rows %s" % (name, kb.numel(), kp.numel(), want["n"],
          "OK" if ok else "WRONG", "OK" if ok2 else "WRONG", "OK" if ok3 else "WRONG"), flush=True)
    bad += (not ok) + (not ok2) + (not ok3)

u = lambda n: r(M63, n) * 2 + 1
kb = u(1000); run("tiny build, huge probe (fk)", kb, kb[r(1000, 1 << 27)])
kb = u(1 << 27); run("huge build, tiny probe", kb, kb[r(1 << 27, 1000)])
kb = u(1 << 20); run("all probes carry ONE key", kb, kb[:1].repeat(1 << 24))
kb = u(1 << 24); kb[: 1 << 20] = 77; run("2^20 build rows share one key, probes uniform", kb, u(1 << 24))
kb = u(1 << 22); run("no probe row matches", kb, r(M63, 1 << 22) * 2)
kb = u(1); run("single build row", kb, kb.repeat(1 << 20))
kb = u(1 << 26); run("2^26 x 2^28 fk", kb, kb[r(1 << 26, 1 << 28)])
kb = torch.arange(1 << 26, dtype=torch.int64, device=dev); run("dense sorted 0..2^26-1 both sides", kb, kb.flip(0))
kb = (torch.arange(1 << 24, dtype=torch.int64, device=dev) << 40); run("keys = i << 40 (low bits zero)", kb, kb[r(1 << 24, 1 << 25)])
kb = u(1 << 24) | -(1 << 63); run("all keys with the top bit set", kb, kb[r(1 << 24, 1 << 24)])
print("ALL OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)

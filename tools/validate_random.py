#!/usr/bin/env python3
"""Randomized large-scale cross-check against an independent torch implementation (sort / unique / searchsorted):
random sizes (2^16 .. 2^25 rows per side, independently), key distributions, fan-outs and flags.
Usage: tools/validate_random.py [iters] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import hashmergejoin_amd as H
from hashmergejoin_amd.join import _memcpy_d2d

dev = torch.device("cuda:0")
M63 = (1 << 63) - 1
SIGN = torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)

def reference(B, P):
    kb, vb = B[:, 0] ^ SIGN, B[:, 1]
    kp, vp = P[:, 0] ^ SIGN, P[:, 1]
    sk, order = torch.sort(kb, stable=True)
    sv = vb[order]
    uk, inv, cnt = torch.unique_consecutive(sk, return_inverse=True, return_counts=True)
    sumv = torch.zeros_like(uk).scatter_add_(0, inv, sv)
    firstv = sv[torch.cumsum(cnt, 0) - cnt]
    pos = torch.searchsorted(uk, kp).clamp_(max=uk.numel() - 1)
    hit = uk[pos] == kp
    c = cnt[pos] * hit
    return {"n": int(c.sum().item()), "sum_r": int((sumv[pos] * hit).sum().item()), "sum_s": int((vp * c).sum().item()),
            "fw_n": int(hit.sum().item()), "fw_sum_r": int((firstv[pos] * hit).sum().item()),
            "fw_sum_s": int((vp * hit).sum().item()), "sum_p": int(vp.sum().item())}

def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    g = torch.Generator(device=dev); g.manual_seed(seed)
    cpu = torch.Generator(); cpu.manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=cpu).item())
    r = lambda hi, size: torch.randint(0, hi, (size,), dtype=torch.int64, device=dev, generator=g)
    ex = H.Executor(0)
    bad = 0
    slow = []
    _join = ex.join_device
    ex.set_profiling(True)
    PH = ("ms_hist", "ms_scan", "ms_scatter", "ms_offsets", "ms_probe_count", "ms_out_scan", "ms_probe_write", "ms_order")
    def timed_join(B, P, fl):  # GPU time of the join's kernels (workspace growth on the host is not a cliff)
        r_ = _join(B, P, fl)
        t = ex.last_timing()
        dt = sum(t[k] for k in PH) * 1e-3
        if dt > float(os.environ.get("VR_SLOW", "1.0")):
            slow.append((dt, fl))
        return r_
    ex.join_device = timed_join
    for it in range(iters):
        lo_, hi_ = int(os.environ.get("VR_MINLOG", "16")), int(os.environ.get("VR_MAXLOG", "25")) + 1
        nb, npb = 1 << ri(lo_, hi_), 1 << ri(lo_, hi_)
        nb += ri(0, 5000); npb += ri(0, 5000)
        kind = ["uniform", "dups", "dense", "tagged", "hotprobe", "hotbuild", "fk", "densepow2", "fewkeys", "strided"][ri(0, 10)]
        if kind == "uniform":
            kb = r(M63, nb) * 2 + r(2, nb); kp = r(M63, npb) * 2
            m = npb // 2; kp[:m] = kb[r(nb, m)]
        elif kind == "dups":
            d = max(2, nb // ri(1, 5)); kb = (r(d, nb) * 0x9E3779B97F4A7C15) & -1; kp = (r(2 * d, npb) * 0x9E3779B97F4A7C15) & -1
        elif kind == "dense":
            kb = r(2 * nb, nb); kp = r(2 * nb, npb)
        elif kind == "tagged":
            idb = ri(18, 41); kb = (r(3, nb) << 61) | r(1 << idb, nb); kp = (r(3, npb) << 61) | r(1 << idb, npb)
            m = npb // 2; kp[:m] = kb[r(nb, m)]
        elif kind == "hotprobe":
            kb = r(M63, nb) * 2 + 1; kp = kb[r(nb, npb)]; kp[r(npb, npb // 10)] = kb[0]
        elif kind == "hotbuild":
            kb = r(M63, nb) * 2 + 1; kb[r(nb, min(nb // 20, 200000))] = 0x1234567890ABCDEF; kp = kb[r(nb, npb)]
            kp[: npb // 2] = r(M63, npb // 2) * 2  # half of the probes miss (keeps the cross product finite)
        elif kind == "densepow2":  # dense 0..N-1, N just above a power of two: a few keys beyond the sampled prefix
            N = (1 << (nb.bit_length() - 1)) + ri(1, 2000)
            kb = torch.randperm(N, device=dev, generator=g)[: min(nb, N)]; nb = kb.numel()
            kp = r(N, npb)
        elif kind == "fewkeys":    # a handful of keys on both sides: giant cross products / giant equal-key groups
            pool = r(M63, ri(1, 40)) * 2 + 1
            kb = pool[r(pool.numel(), min(nb, 1 << 18))]; nb = kb.numel()
            kp = r(M63, npb) * 2; m = ri(0, 3000); kp[:m] = pool[r(pool.numel(), m)] if m else kp[:0]
        elif kind == "strided":    # keys = i * stride: varying bits spread out
            st = [3, 0x1111, 1 << 20, (1 << 33) + 1][ri(0, 4)]
            kb = (torch.randperm(nb, device=dev, generator=g) * st) & M63; kp = (r(2 * nb, npb) * st) & M63
        else:  # fk: every probe row references a build key
            kb = r(M63, nb) * 2 + r(2, nb); kp = kb[r(nb, npb)]
        B = torch.stack([kb, r(M63, nb)], 1).contiguous(); P = torch.stack([kp, r(M63, npb)], 1).contiguous()
        if it < int(os.environ.get("VR_START", "0")):
            continue
        if os.environ.get("VR_VERBOSE"):
            print("   config", it, kind, nb, npb, flush=True)
        want = reference(B, P)
        if os.environ.get("VR_VERBOSE"):
            print("   reference done: matches", want["n"], flush=True)
        res = ex.join_device(B, P, 0)
        ok = (int(res.n_matches), s64(int(res.sum_r)), s64(int(res.sum_s))) == (want["n"], want["sum_r"], want["sum_s"])
        f = ex.join_device(B, P, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE)
        ok2 = (int(f.n_matches), s64(int(f.sum_r)), s64(int(f.sum_s)), s64(int(f.sum_probe_all))) == (want["fw_n"], want["fw_sum_r"], want["fw_sum_s"], want["sum_p"])
        ok3 = True
        if want["n"] <= (1 << 26):
            c = ex.join_device(B, P, H.HMJ_CHECKSUM).checks()
            for fl in (H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM, H.HMJ_ORDERED | H.HMJ_CHECKSUM, H.HMJ_FIRST_WINS | H.HMJ_ORDERED):
                if os.environ.get("VR_VERBOSE"):
                    print("   flags", fl, flush=True)
                m = ex.join_device(B, P, fl)
                if fl & H.HMJ_FIRST_WINS:
                    ok3 = ok3 and int(m.n_matches) == want["fw_n"] and s64(int(m.sum_r)) == want["fw_sum_r"]
                else:
                    ok3 = ok3 and m.checks() == c and int(m.n_matches) == want["n"]
                n = int(m.n_matches)
                if n:
                    cols = []
                    for ptr in (m.key, m.rval, m.sval):
                        t = torch.empty(n, dtype=torch.int64, device=dev); _memcpy_d2d(torch, t, ptr, n * 8); cols.append(t)
                    if fl & H.HMJ_ORDERED:
                        ks = cols[0] ^ SIGN
                        ok3 = ok3 and bool((ks[1:] >= ks[:-1]).all())
                    # sums over the materialised rows equal the reported sums
                    ok3 = ok3 and s64(int(cols[1].sum().item())) == s64(int(m.sum_r)) and s64(int(cols[2].sum().item())) == s64(int(m.sum_s))
                ex.release_result()
        print("it %3d %-9s nb=%9d np=%9d matches %12d : count %s first-wins %s rows %s%s" % (
            it, kind, nb, npb, want["n"], "OK" if ok else "WRONG", "OK" if ok2 else "WRONG", "OK" if ok3 else "WRONG",
            "  SLOW " + ", ".join("%.3fs(flags %d)" % x for x in slow) if slow else ""), flush=True)
        del slow[:]
        bad += (not ok) + (not ok2) + (not ok3)
        del B, P
    print("ALL OK" if bad == 0 else "%d MISMATCHES" % bad)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()

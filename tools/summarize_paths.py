#!/usr/bin/env python3
"""One table per workload of tools/profile_paths.sh: per kernel the mean duration (rocprofv3 --kernel-trace --stats), the HBM
bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, both in KiB; the factor 2 is the guide's gfx950 correction for
wide streaming reads), the rate those bytes give, the L2 hit rate, the share of LDS-active cycles lost to bank conflicts and the
share of wave cycles spent waiting (any / on LDS).   Usage: tools/summarize_paths.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import json
import os
import re
import sys

root = sys.argv[1]
SKIP = ("gen_", "rocclr", "elementwise", "vectorized", "CatArray", "at::native", "fill_probe", "key_sample")


def parse(path):
    stats, pmc, line = {}, {}, None
    for ln in open(path):
        ln = ln.rstrip("\n")
        if ln.startswith("{"):
            line = json.loads(ln)
        elif "calls=" in ln and "avg_us=" in ln:
            name = ln[:62].strip()
            m = re.search(r"calls=(\d+)\s+total_ms=\s*([\d.]+)\s+avg_us=\s*([\d.]+)", ln)
            stats[name] = (int(m.group(1)), float(m.group(2)), float(m.group(3)))
        elif ln.startswith("  ") and "=" in ln:
            name = ln[2:60].strip()
            d = pmc.setdefault(name, {})
            for k, v in re.findall(r"(\w+)=([\d.e+-]+)\(n=\d+\)", ln):
                d[k] = float(v)
    return stats, pmc, line


for wl in sorted(os.listdir(root)):
    f = os.path.join(root, wl, "summary.txt")
    if not os.path.exists(f):
        continue
    stats, pmc, line = parse(f)
    print("=" * 150)
    print("workload %s: %s" % (wl, json.dumps(line)))
    print("%-58s %5s %9s %9s %9s %7s %6s %8s %8s" % ("kernel", "calls", "avg_us", "HBM_MB", "TB/s", "L2hit", "bankc", "wait_any", "wait_lds"))
    tot = sum(s[1] for k, s in stats.items() if not any(x in k for x in SKIP))
    for k, (calls, total_ms, avg_us) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if any(x in k for x in SKIP):
            continue
        p = pmc.get(k[:58].strip(), pmc.get(k, {}))
        if not p:  # long template names are cut differently in the two sections
            for kk, vv in pmc.items():
                if kk.startswith(k[:40]):
                    p = vv
        hbm = (2 * p.get("FETCH_SIZE", 0) + p.get("WRITE_SIZE", 0)) * 1024
        hit = p.get("TCC_HIT_sum", 0) / max(1.0, p.get("TCC_HIT_sum", 0) + p.get("TCC_MISS_sum", 0))
        bank = p.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, p.get("SQ_LDS_IDX_ACTIVE", 0))
        wc = max(1.0, p.get("SQ_WAVE_CYCLES", 0))
        print("%-58s %5d %9.1f %9.1f %9.2f %7.3f %6.2f %8.2f %8.2f" % (
            k[:58], calls, avg_us, hbm / 1e6, hbm / (avg_us * 1e-6) / 1e12 if avg_us else 0, hit, bank,
            p.get("SQ_WAIT_ANY", 0) / wc, p.get("SQ_WAIT_INST_LDS", 0) / wc))
    print("kernels of the workload (all calls): %.3f ms" % tot)

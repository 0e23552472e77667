"""CPU-only checks of the boundary: the C-ABI library loads without a GPU, exports every symbol
include/hmj.h declares, fails loudly (no fallback) when no device exists, and the planner /
host-side helpers behave."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "hmj.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmj_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import hashmergejoin_amd as H

    L = H.load_library()
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), "libhmj_hip.so does not export %s" % s


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    import hashmergejoin_amd as H

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(H.HmjError) as ei:
        H.Executor(0)
    assert ei.value.code == -2  # HMJ_E_NODEV
    L = H.load_library()
    assert L.hmj_strerror(-2) == b"no usable HIP device"
    assert L.hmj_join_u64_device(None, None, 0, None, 0, 0, None) == -1  # NULL ctx -> HMJ_E_ARG


def test_planner():
    import hashmergejoin_amd as H

    # average build partition <= 4096 rows (LDS table of probe.hip), <= 9 bits per LSD pass
    assert H.plan(0) == (0, [])
    assert H.plan(4096) == (0, [])
    assert H.plan(4097) == (0, [])  # up to 4096 + 256 rows per partition are tolerated
    assert H.plan(4353) == (1, [1])
    assert H.plan((1 << 28) + 100000) == (16, [8, 8])  # a shard slightly over 2^28 keeps the 2 x 8-bit plan
    assert H.plan(10 ** 6) == (8, [8])
    assert H.plan(1 << 26) == (14, [7, 7])
    assert H.plan(1 << 28) == (16, [8, 8])
    tb, pb = H.plan(1 << 31)
    assert tb == 19 and sum(pb) == 19 and max(pb) <= 9


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hashmergejoin_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                for needle in ("pyoracle", "import oracle", "from oracle", "libhmj_oracle", "libhmj_ref",
                               '#include "hmj_oracle', "orc_"):
                    hits = [ln for ln in txt.splitlines() if needle in ln and not ln.lstrip().startswith(("//", "#", "*", "/*"))
                            and "// " + needle not in ln]
                    hits = [ln for ln in hits if not re.search(r"//.*" + re.escape(needle), ln)]
                    assert not hits, (f, needle, hits)


def test_path_bits_are_distinct_and_mirrored_by_the_binding():
    # hmj_timing.path is a bit set the tests assert on: two names on one bit (round 4 put HMJ_PATH_GLOBAL_TABLE on
    # HMJ_PATH_SORTED_FK_WIDE's for a few hours) make those assertions mean something else
    from hashmergejoin_amd import _lib

    src = open(os.path.join(ROOT, "include", "hmj.h")).read()
    seen = {}
    for name, v in re.findall(r"#define (HMJ_PATH_\w+) (0x[0-9a-fA-F]+)u", src):
        bit = int(v, 16)
        assert bit and bit & (bit - 1) == 0, (name, v)
        assert bit not in seen, (name, seen[bit])
        seen[bit] = name
        assert getattr(_lib, name) == bit, name
    assert len(seen) >= 21

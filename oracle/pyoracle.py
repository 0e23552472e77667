"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU oracle.

Two libraries live beside this file:
  * libhmj_oracle.so      -- our plain-C restatement (hmj_oracle.c); travels to the GPU box.
  * _ref/libhmj_ref.so    -- the real reference, compiled from /root/reference by
                             oracle/Makefile (ref_driver.cc); git-ignored, travels prebuilt.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (hashmergejoin_amd/) never imports it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_U64P = C.POINTER(C.c_uint64)

SEED_B = 0x243F6A8885A308D3
PI_A = 0x9E3779B1
PI_B = 12345
VAL_XOR = 0x9E3779B97F4A7C15


M64 = (1 << 64) - 1


class Checks(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_matches", "sum_r", "sum_s", "xor_fold", "mix_sum")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _p(a):
    return a.ctypes.data_as(_U64P) if a is not None else None


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


class Oracle:
    """The C restatement (always available once `make -C oracle` has run)."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "libhmj_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library missing: run `make -C oracle` (%s)" % path)
        L = self.lib = C.CDLL(path)
        u, i, vp = C.c_uint64, C.c_int, _U64P
        L.orc_mix64.restype = u
        L.orc_mix64.argtypes = [u]
        L.orc_unmix64.restype = u
        L.orc_unmix64.argtypes = [u]
        L.orc_tmix.restype = u
        L.orc_tmix.argtypes = [u, u, u]
        L.orc_gen_build.argtypes = [vp, u, u, u]
        L.orc_gen_probe.argtypes = [vp, u, u, u, u, u]
        L.orc_gen_from_cdf.argtypes = [vp, u, u, vp, u, u, u]
        L.orc_gen_uniform_domain.argtypes = [vp, u, u, u, u, u]
        L.orc_checks_of_triples.argtypes = [vp, u, C.POINTER(Checks)]
        L.orc_fnv1a_triples.restype = u
        L.orc_fnv1a_triples.argtypes = [vp, u]
        L.orc_optimal_partition.restype = i
        L.orc_optimal_partition.argtypes = [u]
        L.orc_radix_non_inplace_par.argtypes = [vp, u, i, i, vp]
        L.orc_stable_partition.argtypes = [vp, u, i, i, i, vp, vp]
        L.orc_radix_inplace_seq.argtypes = [vp, u, i]
        L.orc_radix_inplace_par_t1.argtypes = [vp, u, i]
        L.orc_radix_int_non_inplace.argtypes = [vp, u, i, i, vp]
        L.orc_radix_int_inplace_t1.argtypes = [vp, u, i]
        L.orc_merge_iterate.restype = u
        L.orc_merge_iterate.argtypes = [vp, u, vp, u, vp, u, vp]
        L.orc_hashmergejoin.restype = u
        L.orc_hashmergejoin.argtypes = [vp, u, vp, u, i, vp, u, vp]
        L.orc_hashmergejoin2.restype = u
        L.orc_hashmergejoin2.argtypes = [vp, u, vp, u, vp, u, vp]
        L.orc_partition_sizes.argtypes = [vp, u, i, vp]
        L.orc_partitioned_table_sizes.argtypes = [vp, u, i, vp]
        L.orc_partitioned_join_sum.restype = u
        L.orc_partitioned_join_sum.argtypes = [vp, u, vp, u, i, vp]
        L.orc_equijoin.restype = u
        L.orc_equijoin.argtypes = [vp, u, vp, u, i, vp, u, C.POINTER(Checks)]

    # ---- generators: return (n,2) uint64 arrays {key,val} -------------------------------
    def mix64(self, x):
        return int(self.lib.orc_mix64(x & (2**64 - 1)))

    def unmix64(self, x):
        return int(self.lib.orc_unmix64(x & (2**64 - 1)))

    def tmix(self, k, r, s):
        return int(self.lib.orc_tmix(k, r, s))

    def gen_build(self, n, start=0, seed=SEED_B):
        a = np.empty((n, 2), np.uint64)
        self.lib.orc_gen_build(_p(a), n, start, seed)
        return a

    def gen_probe(self, n, n_build, start=0, seed=SEED_B, miss_mod=0):
        a = np.empty((n, 2), np.uint64)
        self.lib.orc_gen_probe(_p(a), n, start, n_build, seed, miss_mod)
        return a

    def gen_from_cdf(self, n, thr, start=0, seed=SEED_B, zseed=0x1234567):
        thr = _u64(thr)
        a = np.empty((n, 2), np.uint64)
        self.lib.orc_gen_from_cdf(_p(a), n, start, _p(thr), len(thr), seed, zseed)
        return a

    def gen_uniform_domain(self, n, domain, start=0, seed=SEED_B, zseed=0x7654321):
        a = np.empty((n, 2), np.uint64)
        self.lib.orc_gen_uniform_domain(_p(a), n, start, domain, seed, zseed)
        return a

    # ---- checksums ----------------------------------------------------------------------
    def checks_of_triples(self, t):
        t = _u64(t).reshape(-1, 3)
        c = Checks()
        self.lib.orc_checks_of_triples(_p(t), len(t), C.byref(c))
        return c.as_dict()

    def fnv1a_triples(self, t):
        t = _u64(t).reshape(-1, 3)
        return int(self.lib.orc_fnv1a_triples(_p(t), len(t)))

    # ---- restated reference functions -----------------------------------------------------
    def optimal_partition(self, n):
        return int(self.lib.orc_optimal_partition(n))

    def radix_non_inplace_par(self, aos, threads=1, bits=-1):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty((len(aos), 3), np.uint64)
        self.lib.orc_radix_non_inplace_par(_p(aos), len(aos), threads, bits, _p(out))
        return out

    def stable_partition(self, aos, shift, bits, threads=1):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty_like(aos)
        off = np.empty((1 << bits) + 1, np.uint64)
        self.lib.orc_stable_partition(_p(aos), len(aos), threads, shift, bits, _p(out), _p(off))
        return out, off

    def radix_inplace_seq(self, hkv, bits=-1):
        hkv = _u64(hkv).reshape(-1, 3).copy()
        self.lib.orc_radix_inplace_seq(_p(hkv), len(hkv), bits)
        return hkv

    def radix_inplace_par_t1(self, hkv, bits=-1):
        hkv = _u64(hkv).reshape(-1, 3).copy()
        self.lib.orc_radix_inplace_par_t1(_p(hkv), len(hkv), bits)
        return hkv

    def radix_int_non_inplace(self, aos, threads=1, bits=-1):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty_like(aos)
        self.lib.orc_radix_int_non_inplace(_p(aos), len(aos), threads, bits, _p(out))
        return out

    def radix_int_inplace_t1(self, aos, bits=-1):
        aos = _u64(aos).reshape(-1, 2).copy()
        self.lib.orc_radix_int_inplace_t1(_p(aos), len(aos), bits)
        return aos

    def hashmergejoin(self, r, s, threads=1, cap=None):
        r = _u64(r).reshape(-1, 2)
        s = _u64(s).reshape(-1, 2)
        cap = (len(r) + len(s) + 1) if cap is None else cap
        t = np.zeros((max(cap, 1), 3), np.uint64)
        sm = C.c_uint64(0)
        n = int(self.lib.orc_hashmergejoin(_p(r), len(r), _p(s), len(s), threads, _p(t), cap,
                                           C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value), t[: min(n, cap)]

    def hashmergejoin2(self, r_hkv, s_hkv, cap=None):
        r = _u64(r_hkv).reshape(-1, 3).copy()
        s = _u64(s_hkv).reshape(-1, 3).copy()
        cap = (len(r) + len(s) + 1) if cap is None else cap
        t = np.zeros((max(cap, 1), 3), np.uint64)
        sm = C.c_uint64(0)
        n = int(self.lib.orc_hashmergejoin2(_p(r), len(r), _p(s), len(s), _p(t), cap,
                                            C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value), t[: min(n, cap)]

    def partition_sizes(self, aos, bits):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty(1 << bits, np.uint64)
        self.lib.orc_partition_sizes(_p(aos), len(aos), bits, _p(out))
        return out

    def partitioned_table_sizes(self, aos, bits):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty(1 << bits, np.uint64)
        self.lib.orc_partitioned_table_sizes(_p(aos), len(aos), bits, _p(out))
        return out

    def partitioned_join_sum(self, probe, build, bits=10):
        probe = _u64(probe).reshape(-1, 2)
        build = _u64(build).reshape(-1, 2)
        f = C.c_uint64(0)
        s = int(self.lib.orc_partitioned_join_sum(_p(probe), len(probe), _p(build), len(build),
                                                  bits, C.cast(C.byref(f), _U64P)))
        return s, int(f.value)

    def equijoin(self, r, s, first_wins=False, cap=None):
        """Relational equi-join, R=build S=probe.  Returns (checks dict, sorted triples)."""
        r = _u64(r).reshape(-1, 2)
        s = _u64(s).reshape(-1, 2)
        c = Checks()
        if cap is None:  # count first, then materialise everything
            n = int(self.lib.orc_equijoin(_p(r), len(r), _p(s), len(s), int(first_wins), None, 0,
                                          C.byref(c)))
            cap = n
        t = np.zeros((max(cap, 1), 3), np.uint64)
        n = int(self.lib.orc_equijoin(_p(r), len(r), _p(s), len(s), int(first_wins), _p(t), cap,
                                      C.byref(c)))
        return c.as_dict(), t[: min(n, cap)]


def strgen_mix64(x):
    x &= M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def create_strvec(number, words, seed):
    """Python restatement of oracle/strgen_restated.h create_strvec (itself the restated strgen.cc:27-61):
    list of (key, value).  Pure-Python loops: small and medium sizes only."""
    import math

    sq = int(math.ceil(math.sqrt(float(number))))
    assert sq <= len(words), "word list too short"
    pairs = [(words[i], i) for i in range(sq)]
    done = len(pairs) >= number
    i = 0
    while i < sq and not done:
        wi = pairs[i][0]
        for j in range(sq):
            if len(pairs) == number:
                done = True
                break
            pairs.append((wi + "-" + pairs[j][0], i + j))
        i += 1
    del pairs[number:]
    for k in range(len(pairs), 1, -1):
        j = strgen_mix64(seed + k) % k
        pairs[k - 1], pairs[j] = pairs[j], pairs[k - 1]
    return pairs


def fnv_relation(pairs):
    h = 0xCBF29CE484222325
    for key, val in pairs:
        for ch in key.encode() + int(val).to_bytes(8, "little"):
            h = ((h ^ ch) * 0x100000001B3) & M64
    return h


class Reference:
    """The real reference compiled into oracle/_ref/ (None-like if the .so is absent)."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "libhmj_ref.so")
        self.available = os.path.exists(path)
        if not self.available:
            return
        L = self.lib = C.CDLL(path)
        u, i, ui, vp, vd = C.c_uint64, C.c_int, C.c_uint, _U64P, C.c_void_p
        L.ref_optimal_partition.restype = i
        L.ref_optimal_partition.argtypes = [u]
        L.ref_hashmergejoin_u64.restype = u
        L.ref_hashmergejoin_u64.argtypes = [vp, u, vp, u, ui, vp, u, vp]
        L.ref_pairs_new.restype = vd
        L.ref_pairs_new.argtypes = [vp, u]
        L.ref_pairs_free.argtypes = [vd]
        L.ref_hashmergejoin_pairs.restype = u
        L.ref_hashmergejoin_pairs.argtypes = [vd, vd, ui, vp]
        L.ref_radix_non_inplace_par_u64.argtypes = [vp, u, i, i, vp]
        L.ref_radix_inplace_seq_u64.argtypes = [vp, u, i]
        L.ref_radix_inplace_par_u64.argtypes = [vp, u, i, i]
        L.ref_radix_int_non_inplace_u64.argtypes = [vp, u, i, i, vp]
        L.ref_radix_int_non_inplace_pairs.argtypes = [vd, vd, i]
        L.ref_radix_int_inplace_u64.argtypes = [vp, u, i, i]
        L.ref_partition_only_u64.argtypes = [vp, u, i, i, vp, vp]
        L.ref_partitioned_hash_table_sizes_u64.argtypes = [vp, u, i, i, vp]
        L.ref_partitioned_join_sum_u64.restype = u
        L.ref_partitioned_join_sum_u64.argtypes = [vp, u, vp, u, i, i, vp]
        L.ref_hashmergejoin_str.restype = u
        L.ref_hashmergejoin_str.argtypes = [u, u, u, ui, vp, u, vp]
        L.ref_hashmergejoin2_u64.restype = u
        L.ref_hashmergejoin2_u64.argtypes = [vp, u, vp, u, ui, vp, u, vp]

    def optimal_partition(self, n):
        return int(self.lib.ref_optimal_partition(n))

    def hashmergejoin(self, r, s, threads=1, cap=None):
        r = _u64(r).reshape(-1, 2)
        s = _u64(s).reshape(-1, 2)
        cap = (len(r) + len(s) + 1) if cap is None else cap
        t = np.zeros((max(cap, 1), 3), np.uint64)
        sm = C.c_uint64(0)
        n = int(self.lib.ref_hashmergejoin_u64(_p(r), len(r), _p(s), len(s), threads, _p(t), cap,
                                               C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value), t[: min(n, cap)]

    def hashmergejoin2(self, r_hkv, s_hkv, threads=1, cap=None):
        r = _u64(r_hkv).reshape(-1, 3).copy()
        s = _u64(s_hkv).reshape(-1, 3).copy()
        cap = (len(r) + len(s) + 1) if cap is None else cap
        t = np.zeros((max(cap, 1), 3), np.uint64)
        sm = C.c_uint64(0)
        n = int(self.lib.ref_hashmergejoin2_u64(_p(r), len(r), _p(s), len(s), threads, _p(t),
                                                cap, C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value), t[: min(n, cap)]

    def radix_non_inplace_par(self, aos, threads=1, bits=-1):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty((len(aos), 3), np.uint64)
        self.lib.ref_radix_non_inplace_par_u64(_p(aos), len(aos), threads, bits, _p(out))
        return out

    def radix_inplace_seq(self, hkv, bits=-1):
        hkv = _u64(hkv).reshape(-1, 3).copy()
        self.lib.ref_radix_inplace_seq_u64(_p(hkv), len(hkv), bits)
        return hkv

    def radix_inplace_par(self, hkv, threads=1, bits=-1):
        hkv = _u64(hkv).reshape(-1, 3).copy()
        self.lib.ref_radix_inplace_par_u64(_p(hkv), len(hkv), threads, bits)
        return hkv

    def radix_int_non_inplace(self, aos, threads=1, bits=-1):
        aos = _u64(aos).reshape(-1, 2)
        out = np.empty_like(aos)
        self.lib.ref_radix_int_non_inplace_u64(_p(aos), len(aos), threads, bits, _p(out))
        return out

    def radix_int_inplace(self, aos, threads=1, bits=-1):
        aos = _u64(aos).reshape(-1, 2).copy()
        self.lib.ref_radix_int_inplace_u64(_p(aos), len(aos), threads, bits)
        return aos

    def partition_only(self, aos, threads, bits, content=False):
        aos = _u64(aos).reshape(-1, 2)
        sizes = np.empty(1 << bits, np.uint64)
        cont = np.empty_like(aos) if content else None
        self.lib.ref_partition_only_u64(_p(aos), len(aos), threads, bits, _p(sizes), _p(cont))
        return (sizes, cont) if content else sizes

    def partitioned_table_sizes(self, aos, threads, bits):
        aos = _u64(aos).reshape(-1, 2)
        sizes = np.empty(1 << bits, np.uint64)
        self.lib.ref_partitioned_hash_table_sizes_u64(_p(aos), len(aos), threads, bits, _p(sizes))
        return sizes

    def partitioned_join_sum(self, probe, build, threads=1, bits=10):
        probe = _u64(probe).reshape(-1, 2)
        build = _u64(build).reshape(-1, 2)
        f = C.c_uint64(0)
        s = int(self.lib.ref_partitioned_join_sum_u64(_p(probe), len(probe), _p(build),
                                                      len(build), threads, bits,
                                                      C.cast(C.byref(f), _U64P)))
        return s, int(f.value)

    def hashmergejoin_str(self, nr, ns, seed, threads=1):
        """Reference HashMergeJoin over synthetic std::string-keyed relations (ref_driver.cc)."""
        cap = nr + ns + 1
        pairs = np.zeros((cap, 2), np.uint64)
        sm = C.c_uint64(0)
        n = int(self.lib.ref_hashmergejoin_str(nr, ns, seed, threads, _p(pairs), cap, C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value), pairs[:n]

    def hashmergejoin_strgen(self, words_path, n, seed_r=1, seed_s=2, threads=1):
        """Reference HashMergeJoin<KeyValVec::iterator,...> over two restated create_strvec(n) relations
        (ref_driver.cc ref_hashmergejoin_strgen).  Returns dict(n, sum, pairs, fnv_r, fnv_s, distinct)."""
        pairs = np.zeros((n + 1, 2), np.uint64)
        sm, dist = C.c_uint64(0), C.c_uint64(0)
        fnv = (C.c_uint64 * 2)()
        self.lib.ref_hashmergejoin_strgen.restype = C.c_uint64
        self.lib.ref_hashmergejoin_strgen.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, _U64P, C.c_uint64,
                                                      _U64P, _U64P, _U64P]
        cnt = int(self.lib.ref_hashmergejoin_strgen(words_path.encode(), n, seed_r, seed_s, threads, _p(pairs), n + 1,
                                                    C.cast(C.byref(sm), _U64P), fnv, C.cast(C.byref(dist), _U64P)))
        return {"n": cnt, "sum": int(sm.value), "pairs": pairs[:cnt], "fnv_r": int(fnv[0]), "fnv_s": int(fnv[1]),
                "distinct": int(dist.value)}

    def hashmergejoin_strgen_timed(self, words_path, n, threads, reps=3, seed_r=1, seed_s=2):
        """ref_hashmergejoin_strgen_timed: best seconds of construct + iterate over two create_strvec(n) relations (generation
        outside the clock, as hashjoin_bench.cc:115-134) and (count, sum, ordered FNV of the pairs)."""
        out3 = (C.c_uint64 * 3)()
        self.lib.ref_hashmergejoin_strgen_timed.restype = C.c_double
        self.lib.ref_hashmergejoin_strgen_timed.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_int, _U64P]
        sec = float(self.lib.ref_hashmergejoin_strgen_timed(words_path.encode(), n, seed_r, seed_s, threads, reps, out3))
        return sec, (int(out3[0]), int(out3[1]), int(out3[2]))

    # opaque PairVec handles so a timed region excludes the AoS->vector conversion
    def pairs_new(self, aos):
        aos = _u64(aos).reshape(-1, 2)
        return self.lib.ref_pairs_new(_p(aos), len(aos))

    def pairs_free(self, h):
        self.lib.ref_pairs_free(h)

    def hashmergejoin_pairs(self, rh, sh, threads):
        sm = C.c_uint64(0)
        n = int(self.lib.ref_hashmergejoin_pairs(rh, sh, threads, C.cast(C.byref(sm), _U64P)))
        return n, int(sm.value)

    def radix_int_non_inplace_pairs(self, inh, outh, threads):
        self.lib.ref_radix_int_non_inplace_pairs(inh, outh, threads)

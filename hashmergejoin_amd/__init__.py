"""hashmergejoin_amd -- MI355X-native radix-partitioned hash join (host-side Python binding).

The product is libhmj_hip.so (hand-written HIP for gfx950 behind the C ABI of include/hmj.h).  This
package is a thin ctypes binding over that ABI plus a mirror of the reference's operator surface
(`HashMergeJoin`, reference hashjoin.h:33-199).  There is no CPU implementation here: if the
library or a GPU is missing, calls raise.
"""
from ._lib import (HMJ_CHECKSUM, HMJ_FIRST_WINS, HMJ_MATERIALIZE, HMJ_ORDERED, HMJ_PATH_CHUNKED_BUILD, HMJ_PATH_DENSE_BUILD,
                   HMJ_PATH_EXACT, HMJ_PATH_GLOBAL_TABLE, HMJ_PATH_HOST_PIPELINE, HMJ_PATH_HOT_KEY_HINT, HMJ_PATH_LOOKBACK_TIMEOUT, HMJ_PATH_ORDER_BY_KEY, HMJ_PATH_ORDER_BY_RANK_SORT, HMJ_PATH_ORDER_DEFERRED, HMJ_PATH_ORDERED_EXPANSION, HMJ_PATH_LDS_TABLE,
                   HMJ_PATH_KEY_RANGES, HMJ_PATH_PREPARED, HMJ_PATH_PRESORTED, HMJ_PATH_RANK_RUNS, HMJ_PATH_SORT_MSD, HMJ_PATH_SLAB, HMJ_PATH_SLAB_ONE_PASS, HMJ_PATH_SLAB_PROBE, HMJ_PATH_SORTED_FK, HMJ_PATH_SORTED_FK_HALF, HMJ_PATH_SORTED_FK_WIDE, HMJ_PATH_SORTED_WRITE, HMJ_PATH_SPLIT, HMJ_PATH_UNIQ_WRITE, HMJ_PATH_WINDOW,
                   HMJ_SUM_PROBE, HmjError, JoinResult, Timing, lib_path, load_library)
from .join import Executor, HashMergeJoin, plan

__all__ = ["Executor", "HashMergeJoin", "plan", "HmjError", "JoinResult", "Timing", "load_library",
           "lib_path", "HMJ_MATERIALIZE", "HMJ_ORDERED", "HMJ_FIRST_WINS", "HMJ_CHECKSUM",
           "HMJ_SUM_PROBE"]

"""tools/checksum_rate.py -- what the content checks of the MEX cache cost on this host: sdm_mexcache_checksum (the checksum residency is decided on)
over arrays of several sizes, cache-hot and after the caches have been swept by another array (what a gateway sees of an array the DMA
engine has just written)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from sedumi_amd import capi  # noqa: E402

lib = ctypes.CDLL(capi.DEFAULT_LIB)
lib.sdm_mexcache_checksum.restype = ctypes.c_uint64
sweep = np.ones(96 << 20 >> 3)
print("cpus", os.cpu_count())
for n in (65536, 222111, 443556, 4_000_000, 16_000_000):
    a = np.random.default_rng(0).standard_normal(n)
    p = a.ctypes.data_as(ctypes.c_void_p)
    lib.sdm_mexcache_checksum(p, ctypes.c_int64(n))
    k = max(3, int(2e7 // n))
    t0 = time.perf_counter()
    for _ in range(k):
        lib.sdm_mexcache_checksum(p, ctypes.c_int64(n))
    hot = (time.perf_counter() - t0) / k
    cold = 0.0
    for _ in range(3):
        sweep += 1.0
        t0 = time.perf_counter()
        lib.sdm_mexcache_checksum(p, ctypes.c_int64(n))
        cold += (time.perf_counter() - t0) / 3
    print("%9d words: hot %.3f ns/word (%.1f GB/s, %.0f us)   swept %.3f ns/word (%.1f GB/s, %.0f us)" %
          (n, 1e9 * hot / n, 8 * n / hot / 1e9, 1e6 * hot, 1e9 * cold / n, 8 * n / cold / 1e9, 1e6 * cold))

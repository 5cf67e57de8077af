#!/usr/bin/env python3
"""k_corr_gemm2 against the number of 256 x 128 tiles: the same hi rows against lo sets of 7 936 ... 9 216 rows (496 ... 576 tiles on
512 resident workgroups), timed through the library's timers.   python tools/probe_gemm_sizes.py [n_hi n_lo ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_amd import _lib      # noqa: E402


def rows(n, seed):
    rng = np.random.default_rng(seed)
    d = np.zeros((n, 1024), np.int16)
    z = rng.integers(0, 16, size=(n, 64, 60))
    for s in range(64):
        np.add.at(d, (np.arange(n)[:, None], s * 16 + z[:, s, :]), 1)
    return d


def main():
    lib = _lib.Lib(0)
    n_hi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    sizes = [int(a) for a in sys.argv[2:]] or [7936, 8192, 8448, 9088, 9216, 16384, 16512]
    hi = rows(n_hi, 1)
    for n_lo in sizes:
        lo = rows(n_lo, 2)
        lib.correlate(hi, lo, 0.9)
        lib.synchronize()
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            lib.correlate(hi, lo, 0.9)
            lib.synchronize()
            t.append(time.perf_counter() - t0)
        tm = lib.timers() if hasattr(lib, "timers") else {}
        print("n_lo %5d  tiles %4d  call %.3f ms  timers %s" % (n_lo, ((n_hi + 255) // 256) * ((n_lo + 127) // 128), 1e3 * min(t), {k: round(v, 4) for k, v in tm.items()} if tm else ""), flush=True)


if __name__ == "__main__":
    main()

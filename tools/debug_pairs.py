#!/usr/bin/env python3
"""Which pairs differ between mad_correlate and the oracle on the ragged-size case of tests/test_gpu_stages.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mad_amd import _lib
from oracle import oracle as O
import test_gpu_stages as T

lib = _lib.Lib(0)
n_hi, n_lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1100, int(sys.argv[2]) if len(sys.argv) > 2 else 16001
cc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
lo = T._random_descriptors(n_lo, 31)
hi = T._random_descriptors(n_hi, 32, base=lo[n_lo // 3:])
rh, rl, rs, _ = O.correlate(hi, lo, cc)
for rep in range(3):
    gh, gl, gs = lib.correlate(hi, lo, cc)
    ref = set(zip(rh.tolist(), rl.tolist())); got = set(zip(gh.tolist(), gl.tolist()))
    miss, extra = sorted(ref - got), sorted(got - ref)
    print("run", rep, "oracle", len(ref), "device", len(got), "missing", miss[:10], "extra", extra[:10])
    nh, nl = np.linalg.norm(hi.astype(np.float64), axis=1), np.linalg.norm(lo.astype(np.float64), axis=1)
    for (i, j) in miss[:5] + extra[:5]:
        d = int(hi[i].astype(np.int64) @ lo[j].astype(np.int64))
        sc = d / (nh[i] * nl[j])
        row = np.array(sorted(c for (r, c) in ref if r == i))
        print("  pair", (i, j), "dot", d, "score", repr(sc), "score - cc", sc - cc, "| row", i, "has", len(row), "pairs; in the pass of column", j, ":",
              int(np.sum((row // 8192) == j // 8192)), "tile row", i // 256, "tile col", j // 128, "word", j // 32, "bit", j % 32)

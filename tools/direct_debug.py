"""Where does MODE_FAST on the direct stages leave the legacy PCM?  (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from direct_probe import run
from tests.test_gpu_parity import random_batch

rng = np.random.default_rng(31)
batch = random_batch(rng, 700, quiet_fraction=0.15, wild=False)
legacy = run(batch, 0, 0, 1)
direct = run(batch, 0, 1, 1)
start = legacy[1]
fs = batch["frame_start"]
shown = 0
for u in range(len(start) - 1):
    a, b = legacy[0][start[u]:start[u + 1]].astype(np.int32), direct[0][start[u]:start[u + 1]].astype(np.int32)
    d = np.nonzero(a != b)[0]
    if len(d) == 0:
        continue
    m, f, nul = batch["min"][fs[u]:fs[u + 1]], np.maximum(batch["fade"][fs[u]:fs[u + 1]], 1), batch["isnull"][fs[u]:fs[u + 1]]
    span = np.maximum(m.astype(np.int64), f.astype(np.int64) + 1) + 1
    T = np.concatenate([[0], np.cumsum(span)])
    k = int(np.searchsorted(T, d[0], side="right") - 1)
    print("utt %d: %d differ, first at sample %d (max |d| %d) = frame %d + %d; frames (min, fade, null): %s" % (
        u, len(d), d[0], np.abs(a - b).max(), k, d[0] - T[k], list(zip(m.tolist(), f.tolist(), nul.tolist()))))
    fr = batch["frames"][fs[u] + k]
    print("    frame %d: cf %s cb %s cfN0 %.1f" % (k, np.round(fr[7:13], 1), np.round(fr[15:21], 1), fr[13]))
    shown += 1
    if shown >= 12:
        break

"""Diagnostic (KLATT_STAMPS build): where do the eight stage waves of the direct kernel (klatt_direct.h) spend their cycles?

    python tools/ab_probe.py build dst=-DKLATT_STAMPS          (here)
    SPEECHPLAYER_LIB=nvspeechplayer_amd/lib/variants/libspeechPlayer_dst.so python tools/direct_stamps.py all_different 65536 1     (GPU box)
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, _native, workloads
from mixed_probe import jitter, distinct

wl, n, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
base = workloads.make("cfg2", n)
batch = {"cfg2": lambda: base, "jittered": lambda: jitter(base), "distinct": lambda: distinct(base), "all_different": lambda: jitter(distinct(base))}[wl]()
bp = BatchPlayer(batch["sr"], mode=mode)
bp.setOption("tracks", 0)
bp.setOption("direct", 2)
bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
info = bp.kernelInfo()
bp.synthesize(); bp.synthesize()
L = _native.load()
L.speechPlayer_batch_debugStamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
groups = (info["direct_utterances"] + 63) // 64
buf = np.zeros(groups * 64, dtype=np.uint64)
got = L.speechPlayer_batch_debugStamps(bp._h, buf.ctypes.data, len(buf))
st = buf.reshape(-1, 8, 8).astype(np.float64)
ms = float(np.mean(bp.time(4)))
print("%s n=%d mode=%d (%d direct utterances): per stage mean cycles  work / barrier-wait  (over %d workgroups); a launch of this build takes %.3f ms" % (
    wl, n, mode, info["direct_utterances"], st.shape[0], ms))
names = ["T0 source", "T1 N0 NP", "T2 r6 r5", "T3 r4 r3", "T4 r2 r1", "T5 fric p1 p2", "T6 p3 p4", "T7 p5 p6 pcm"]
for s in range(8):
    m = st[:, s, :].mean(axis=0)
    print("  %-14s work %.3e wait %.3e | chunks steady/mixed %6.0f %6.0f | cycles per chunk %7.0f %7.0f" % (
        names[s], m[0], m[1], m[2], m[4], m[5] / max(m[2], 1), m[7] / max(m[4], 1)))

"""Diagnostic (KLATT_STAMPS build): where do the four stage waves spend their cycles?  Not a timing run."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, _native, workloads

wl, n, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 1     # 2: lane-pipelined workgroups (20 utterances each)
per = 16 if layout == 2 else 64
if wl == "staggered":          # cfg2 with per-utterance leading silence: lanes of a wave do not fade together
    from mixed_probe import stagger
    batch = stagger(workloads.make("cfg2", n))
elif wl == "jittered":
    from mixed_probe import jitter
    batch = jitter(workloads.make("cfg2", n))
elif wl == "rotated":
    from mixed_probe import rotate
    batch = rotate(workloads.make("cfg2", n))
else:
    batch = workloads.make(wl, n)
bp = BatchPlayer(batch["sr"], mode=mode, layout=layout)
bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
if layout == -1 and bp.kernelInfo()["lane_pipelined_utterances"] == n:
    per = 16                   # the engine chose the lane-pipelined kernel for the whole batch
bp.synthesize(); bp.synthesize()
L = _native.load()
L.speechPlayer_batch_debugStamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((n // per + 2) * 32, dtype=np.uint64)
got = L.speechPlayer_batch_debugStamps(bp._h, buf.ctypes.data, len(buf))
st = buf[:(n // per) * 32].reshape(-1, 4, 8).astype(np.float64)
ms = float(np.mean(bp.time(4)))
print("%s n=%d mode=%d: per stage mean cycles  work / barrier-wait   (over %d workgroups); a launch of this build takes %.3f ms: %.2f ticks per ns" % (
    wl, n, mode, st.shape[0], ms, (st[:, 0, 0] + st[:, 0, 1]).mean() / (ms * 1e6)))
for s in range(4):
    m = st[:, s, :].mean(axis=0)
    print("  stage %d: work %.3e wait %.3e | chunks steady/fade/general %5.0f %5.0f %5.0f | cycles per chunk %7.0f %7.0f %7.0f" % (
        s, m[0], m[1], m[2], m[3], m[4], m[5] / max(m[2], 1), m[6] / max(m[3], 1), m[7] / max(m[4], 1)))

#!/usr/bin/env python3
"""tools/stamps.py on the steady noisy phoneme of tools/steady_probe.py (diagnostic -DKLATT_STAMPS build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, _native, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
z = np.load(workloads.INPUTS)
names = [b.decode("utf8") for b in z["phoneme_names"]]
b = workloads.cfg1_steady_vowels(n, seconds=1.0)
i = names.index("z"); mask = z["phoneme_mask"][i].astype(bool)
fr = b["frames"]; fr[0::2][:, mask] = z["phoneme_frames"][i][mask]
bp = BatchPlayer(b["sr"], layout=1)
bp.setUtterances(b["frame_start"], fr, b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
bp.synthesize(); bp.synthesize()
L = _native.load()
L.speechPlayer_batch_debugStamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((n // 64 + 2) * 32, dtype=np.uint64)
L.speechPlayer_batch_debugStamps(bp._h, buf.ctypes.data, len(buf))
st = buf[:(n // 64) * 32].reshape(-1, 4, 8).astype(np.float64)
for s in range(4):
    m = st[:, s, :].mean(axis=0)
    print("  stage %d: work %.3e wait %.3e | chunks steady/fade/general %5.0f %5.0f %5.0f | cycles per chunk %7.0f %7.0f %7.0f" % (
        s, m[0], m[1], m[2], m[3], m[4], m[5] / max(m[2], 1), m[6] / max(m[3], 1), m[7] / max(m[4], 1)))

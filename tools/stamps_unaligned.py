"""KLATT_STAMPS build: ONE workgroup of 64 live handles that are not aligned (tools/live_unaligned.py's skewed case), pulled 8192 samples at a
time -- per stage: work against barrier wait, chunks by kind and ticks per chunk.
    python tools/ab_probe.py build stamps=-DKLATT_STAMPS=1
    SPEECHPLAYER_LIB=nvspeechplayer_amd/lib/variants/libspeechPlayer_stamps.so python tools/stamps_unaligned.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, _native
L = _native.load()
L.speechPlayer_debugStreamStamps.argtypes = [ctypes.c_void_p]; L.speechPlayer_debugStreamStamps.restype = None
d = np.load(os.path.join(os.path.dirname(eng.__file__), "data", "workload_inputs.npz"), allow_pickle=True)
lines = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in d["ipa_lines"]]
rng = np.random.default_rng(3)
n = 64
players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
for k, p in enumerate(players):
    fr = list(ipa.generateFramesAndTiming(lines[k % 8], basePitch=90 + 2 * (k // 8), clauseType="."))
    for _ in range(4):
        for f, dd, fd in fr:
            p.queueFrame(f, dd, fd)
    p.synthesize(int(rng.integers(1, 4000)))
group = eng.LiveGroup(players)
group.pullDevice(64)
buf = np.zeros(32, dtype=np.uint64)
L.speechPlayer_debugStreamStamps(buf.ctypes.data)       # (clears what the set-up pulls left)
kms = []
for _ in range(4):
    group.pullDevice(8192)
    kms.append(L.speechPlayer_lastLiveKernelMs(0))
L.speechPlayer_debugStreamStamps(buf.ctypes.data)
print("64 unaligned live handles, 4 pulls of 8192 samples: kernel ms", " ".join("%.2f" % x for x in kms))
st = buf.reshape(4, 8).astype(np.float64)
for s in range(4):
    m = st[s]; nn = np.maximum(m[2:5], 1)
    print("stage slot %d: work %.3e wait %.3e ticks | chunks steady/fade/general %6.0f %6.0f %6.0f | ticks per chunk: steady %6.1f fade %6.1f general %6.1f" % (s, m[0], m[1], m[2], m[3], m[4], m[5] / nn[0], m[6] / nn[1], m[7] / nn[2]))

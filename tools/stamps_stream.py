"""KLATT_STAMPS build (its stream hook: speechPlayer_debugStreamStamps, klatt_engine.hip): where the four stages of ONE live handle
spend their cycles over the pulls of bench.py's single_stream extra -- per stage: chunks decided steady / fade / sample by sample and the
cycles (s_memtime, 100 MHz) spent in each kind, work against barrier wait.

    python tools/ab_probe.py build stamps=-DKLATT_STAMPS=1
    SPEECHPLAYER_LIB=nvspeechplayer_amd/lib/variants/libspeechPlayer_stamps.so python tools/stamps_stream.py [pulls]
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from nvspeechplayer_amd import _native
pulls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = _native.load()
buf = np.zeros(32, dtype=np.uint64)
L.speechPlayer_debugStreamStamps.argtypes = [ctypes.c_void_p]
L.speechPlayer_debugStreamStamps.restype = None
r = bench.single_stream_extra(pulls=pulls)
L.speechPlayer_debugStreamStamps(buf.ctypes.data)
print(json.dumps({k: r[k] for k in ("ms_per_pull_median", "pcm_equal")}))
st = buf.reshape(4, 8).astype(np.float64)
for s in range(4):
    m = st[s]
    n = np.maximum(m[2:5], 1)
    print("stage slot %d: work %.3e wait %.3e ticks | chunks steady/fade/general %6.0f %6.0f %6.0f | ticks per chunk: steady %6.1f fade %6.1f general %6.1f | share of work: %.2f %.2f %.2f"
          % (s, m[0], m[1], m[2], m[3], m[4], m[5] / n[0], m[6] / n[1], m[7] / n[2], m[5] / m[0], m[6] / m[0], m[7] / m[0]))

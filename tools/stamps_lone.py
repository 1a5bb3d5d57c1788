"""KLATT_STAMPS build: one cfg2 utterance alone (64 replicas in its wavefront) on the stages with the frame state machine -- what a live pull runs."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, _native, workloads
for line in range(8):
    b = workloads.make("cfg2", 8).slice(line, 1)
    bp = BatchPlayer(22050, layout=1)
    bp.setOption("tracks", 0); bp.setOption("direct", 0)
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    bp.synthesize(); bp.synthesize()
    L = _native.load()
    L.speechPlayer_batch_debugStamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    buf = np.zeros(3 * 32, dtype=np.uint64)
    L.speechPlayer_batch_debugStamps(bp._h, buf.ctypes.data, len(buf))
    st = buf[:32].reshape(4, 8).astype(np.float64)
    ms = float(np.mean(bp.time(4)))
    print("line %d: %d samples, %.3f ms = %.0f ns/sample" % (line, bp.totalSamples, ms, ms * 1e6 / bp.totalSamples))
    for s in range(4):
        m = st[s]
        print("  stage %d: work %.3e wait %.3e | decisions steady/fade/general %5.0f %5.0f %5.0f | cycles in steady %9.0f fade %9.0f general %9.0f" % (s, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]))
    bp.close()

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads
b = workloads.make("cfg2", 8).slice(1, 1)
for same_seed in (False, True):
    for n in (1, 2, 4, 8, 16, 32, 63, 64, 65, 128):
        bp = eng.BatchPlayer(22050)
        seeds = np.zeros(n, np.uint32) if same_seed else np.arange(n, dtype=np.uint32)
        bp.setUtterancesShared(b["frame_start"], b["frames"], b["min"], b["fade"], np.zeros(n, np.uint32), b["index"], b["isnull"], seeds)
        bp.time(3)
        ms = bp.time(15)
        print("same_seed=%d n=%-4d median %.3f min %.3f max %.3f ms" % (same_seed, n, float(np.median(ms)), float(ms.min()), float(ms.max())))
        bp.close()

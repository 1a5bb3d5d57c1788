"""Does a lone workgroup run faster while the rest of the chip is busy?  (DVFS / idle-state hypothesis)"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads
b = workloads.make("cfg2", 8).slice(1, 1)
def lone(n):
    bp = eng.BatchPlayer(22050)
    bp.setUtterancesShared(b["frame_start"], b["frames"], b["min"], b["fade"], np.zeros(n, np.uint32), b["index"], b["isnull"], np.arange(n, dtype=np.uint32))
    return bp
for n_bg in (0, 64 * 32, 64 * 200, 65536):
    stop = [False]
    th = None
    if n_bg:
        bg = eng.BatchPlayer(22050)
        bg.setIpa(**workloads.cfg2_spec(n_bg))
        def spin():
            while not stop[0]:
                bg.synthesize()
        th = threading.Thread(target=spin); th.start()
        time.sleep(0.2)
    for n in (1, 64):
        bp = lone(n)
        bp.time(3)
        ms = bp.time(15)
        print("background %6d utterances, lone n=%-3d median %.3f min %.3f max %.3f ms" % (n_bg, n, float(np.median(ms)), float(ms.min()), float(ms.max())))
        bp.close()
    if th:
        stop[0] = True; th.join(); bg.close()

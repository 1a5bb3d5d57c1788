"""MODE_FAST headline (cfg2, flat stages) -- kernel ms of this library, both modes, plus cfg3 and the rotated / jittered batches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads
for name, spec in (("cfg2", workloads.cfg2_spec(65536)),):
    for mode in (1, 0):
        bp = eng.BatchPlayer(22050, mode=mode)
        bp.setIpa(**spec)
        bp.time(3)
        ms = bp.time(30)
        print("%s mode %d: mean %.3f median %.3f min %.3f ms  digest %016x" % (name, mode, float(ms.mean()), float(np.median(ms)), float(ms.min()), bp.digest()))
        bp.close()
lists, list_of, seeds = workloads.shared("cfg3", 125000)
for mode in (1, 0):
    bp = eng.BatchPlayer(22050, mode=mode)
    bp.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
    bp.time(3); ms = bp.time(20)
    print("cfg3 mode %d: mean %.3f median %.3f ms digest %016x" % (mode, float(ms.mean()), float(np.median(ms)), bp.digest()))
    bp.close()

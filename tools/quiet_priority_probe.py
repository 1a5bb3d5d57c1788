"""The quiet groups' kernels queued in front of or behind the noisy groups' (option "quiet_last"): ms per launch, same player, alternating.
(Round 6 also measured low-PRIORITY streams for the quiet groups: cfg2 8.45 -> 8.9 ms; not kept.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads
def player(kind, mode):
    bp = eng.BatchPlayer(22050, mode=mode)
    if kind == "cfg2":
        bp.setIpa(**workloads.cfg2_spec(65536))
    elif kind == "cfg3":
        lists, list_of, seeds = workloads.shared("cfg3", 125000)
        bp.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
    else:
        bp.setIpa(**workloads.cfg4_spec(32 * 16384))
    return bp
for kind, reps in (("cfg2", 30), ("cfg3", 30), ("cfg4", 5)):
    for mode in (0, 1):
        bp = player(kind, mode)
        out = {}
        for rnd in range(2):
            for key, last in (("quiet groups first", 0), ("quiet groups last (shipped)", 1)):
                bp.setOption("quiet_last", last)
                bp.time(2)
                out.setdefault(key, []).append(float(np.mean(bp.time(reps))))
        print("%s mode %d: %s   digest %016x" % (kind, mode, " | ".join("%s %s" % (k, " ".join("%.3f" % x for x in v)) for k, v in out.items()), bp.digest()))
        bp.close()

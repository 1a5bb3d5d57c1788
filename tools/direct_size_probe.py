import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads
for n in [int(a) for a in sys.argv[1:]] or (8192, 16384, 20480, 24576, 32768):
    b = workloads.all_different(workloads.make("cfg2", n))
    row = []
    for mode in (0, 1):
        for lean in (0, 1, -1):
            bp = BatchPlayer(b["sr"], mode=mode)
            bp.setOption("tracks", 0); bp.setOption("direct", 2); bp.setOption("direct_lean", lean)
            bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
            bp.time(1); ms = float(np.median(bp.time(5)))
            row.append("m%d lean %2d %6.2f" % (mode, lean, ms))
            bp.close()
    print("n=%6d: %s" % (n, " | ".join(row)), flush=True)

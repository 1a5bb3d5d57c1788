import sys, os
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads
from mixed_probe import rotate, jitter
b = workloads.make("cfg2", 65536)
for name, bb in (("jittered", jitter(b)), ("rotated", rotate(b))):
    bp = BatchPlayer(bb["sr"])
    bp.setUtterances(bb["frame_start"], bb["frames"], bb["min"], bb["fade"], bb["index"], bb["isnull"], bb["seeds"])
    print(name, bp.kernelInfo())
    bp.time(1); print(name, np.mean(bp.time(3)))
    bp.close()

"""Kernel time of the cfg1 recipe against utterance length: the slope is the cost per steady sample, the intercept
what fades, events and the launch cost.  usage: len_probe.py [layout] [n_utt]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads

layout = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
xs, ys = [], []
for seconds in (0.25, 0.5, 1.0, 2.0):
    batch = workloads.cfg1_steady_vowels(n, seconds=seconds)
    bp = BatchPlayer(batch["sr"], layout=layout)
    bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
    bp.synthesize(); bp.wait()
    ms = float(np.mean(bp.time(8)))
    samples = bp.totalSamples // n
    xs.append(samples); ys.append(ms)
    print("layout %d, %d utterances x %.2f s (%d samples each): %.4f ms" % (layout, n, seconds, samples, ms))
    bp.close()
b, a = np.polyfit(xs, ys, 1)
print("fit: %.1f us fixed + %.2f ns per sample" % (a * 1e3, b * 1e6))

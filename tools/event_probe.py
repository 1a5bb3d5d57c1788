"""Kernel time of the cfg1 recipe with its steady frame cut into k identical frames (fade of 1 sample between them):
what one frame boundary (dequeue event + one-sample fade + fade-end event) costs a launch."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads

layout = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
xs, ys = [], []
for k in (1, 5, 9, 17, 33):
    base = workloads.cfg1_steady_vowels(n, seconds=1.0)
    fr = base["frames"].reshape(n, 2, 47)
    M, F = int(base["min"][0]), int(base["fade"][0])
    frames = np.concatenate([np.repeat(fr[:, :1], k, axis=1), fr[:, 1:]], axis=1).reshape(-1, 47)
    mins = np.tile(np.array([M // k] * k + [F], np.uint32), n)
    fades = np.tile(np.array([F] + [1] * (k - 1) + [F], np.uint32), n)
    isnull = np.tile(np.array([0] * k + [1], np.uint8), n)
    start = np.arange(n + 1, dtype=np.int64) * (k + 1)
    bp = BatchPlayer(22050, layout=layout)
    bp.setUtterances(start, frames, mins, fades, np.full(len(mins), -1, np.int32), isnull, base["seeds"])
    bp.synthesize(); bp.wait()
    ms = float(np.mean(bp.time(8)))
    xs.append(k - 1); ys.append(ms)
    print("layout %d, %d frames per utterance (%d samples each): %.4f ms" % (layout, k, bp.totalSamples // n, ms))
    bp.close()
b, a = np.polyfit(xs, ys, 1)
print("fit: %.1f us + %.2f us per extra frame boundary" % (a * 1e3, b * 1e3))

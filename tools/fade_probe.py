"""Kernel time of the cfg1 recipe against the length of its two fades (in and out of silence): per-fade-sample cost."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads

layout = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
xs, ys = [], []
for fade_ms in (50, 200, 400, 800):
    batch = workloads.cfg1_steady_vowels(n, seconds=1.0)
    F = workloads.ms(fade_ms)
    batch["fade"][:] = F
    batch["min"][1::2] = F
    bp = BatchPlayer(batch["sr"], layout=layout)
    bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
    bp.synthesize(); bp.wait()
    ms = float(np.mean(bp.time(8)))
    samples = bp.totalSamples // n
    xs.append(2 * F); ys.append(ms - 31.3e-6 * (samples - 2 * F))
    print("layout %d, fades of %d ms (%d fade samples of %d): %.4f ms" % (layout, fade_ms, 2 * F, samples, ms))
    bp.close()
b, a = np.polyfit(xs, ys, 1)
print("fit (steady samples charged at 31.3 ns): %.1f us fixed + %.2f ns per fade sample" % (a * 1e3, b * 1e6))

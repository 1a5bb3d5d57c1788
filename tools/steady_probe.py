"""How fast is the noisy kernel when nothing fades?  65 536 utterances of one steady phoneme with noise gains (1 s + the fades
into and out of silence), against the speech mix of cfg2: separates the cost of steady samples from the cost of fades."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
z = np.load(workloads.INPUTS)
names = [b.decode("utf8") for b in z["phoneme_names"]]
for ph, seconds in (("z", 1.0), ("s", 1.0), ("a", 1.0)):
    b = workloads.cfg1_steady_vowels(n, seconds=seconds)
    i = names.index(ph)
    mask = z["phoneme_mask"][i].astype(bool)
    fr = b["frames"]
    fr[0::2][:, mask] = z["phoneme_frames"][i][mask]
    if ph == "a":
        fr[0::2, 6] = 0.1            # a vowel with a little aspiration: noisy launch, every resonator steady
    bp = BatchPlayer(b["sr"])
    bp.setUtterances(b["frame_start"], fr, b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    info = bp.kernelInfo()
    bp.time(2)
    ms = float(np.mean(bp.time(5)))
    print("steady /%s/ x %d: %.2f ms  %.3g samples/s  noisy=%s chunk=%d" % (ph, n, ms, bp.totalSamples / ms * 1e3, info["noisy_group"], info["stage_parallel_chunk"]), flush=True)
    bp.close()

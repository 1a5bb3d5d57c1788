"""120 000 random ragged utterances (NaN holds, vibrato, NULL frames, half of them quiet) through every kernel layout:
the PCM pools must agree byte for byte (the lane kernel, layout 0, is the independent implementation)."""
import sys, os, hashlib, time
sys.path.insert(0, os.getcwd())
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import nvspeechplayer_amd as eng
from tests.test_gpu_parity import random_batch
rng = np.random.default_rng(2024)
batch = random_batch(rng, 120000, quiet_fraction=0.5, wild=True, nasal_fraction=0.15)
dig = {}
for layout in (0, 1, 2, -1):
    bp = eng.BatchPlayer(22050, layout=layout)
    bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
    t = time.time(); bp.synthesize(); bp.wait(); dt = time.time() - t
    a, st = bp.readAll()
    dig[layout] = hashlib.sha1(a.tobytes()).hexdigest()
    info = bp.kernelInfo()
    print("layout %2d: %d samples, %.1f ms, sha1 %s, lane-pipelined %d nasal-free %d" % (layout, len(a), dt * 1e3, dig[layout][:12], info["lane_pipelined_utterances"], info["nasal_free_utterances"]))
    bp.close()
assert len(set(dig.values())) == 1, dig
print("all four layouts agree byte for byte")

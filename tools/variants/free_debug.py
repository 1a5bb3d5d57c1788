#!/usr/bin/env python3
"""Smallest batches on the tracked flat stages (KLATT_FLAT_FREE builds: do the hand-over counters make progress?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads
for n in [int(a) for a in sys.argv[1:]] or [64, 256, 4096]:
    b = workloads.make("cfg2", n)
    bp = BatchPlayer(b["sr"])
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    t = time.time(); bp.synthesize(); dg = bp.digest(); dt = time.time() - t
    info = bp.kernelInfo()
    print("n=%d: first launch %.3f s, digest %016x, tracked %d, kernel ms %s" % (n, dt, dg, info["tracked_utterances"], [round(float(x), 3) for x in bp.time(3)]), flush=True)
    bp.close()

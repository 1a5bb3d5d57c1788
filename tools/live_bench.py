#!/usr/bin/env python3
"""How fast do N LIVE handles advance when pulled together (speechPlayer_synthesizeMany) vs one by one?"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = 8192
text = "mɑɪ næɪm ɪz mɑɪkʊl dæɪmɪən kɑɹən"
frames = list(ipa.generateFramesAndTiming(text, clauseType="."))
players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
for p in players:
    for _ in range(4):
        for fr, d, f in frames:
            p.queueFrame(fr, d, f)
eng.SpeechPlayer.synthesizeMany(players, 64)      # warm-up
t0 = time.perf_counter()
total = 0
for _ in range(8):
    bufs = eng.SpeechPlayer.synthesizeMany(players, chunk)
    total += sum(b.length for b in bufs if b is not None)
dt = time.perf_counter() - t0
print("%d live handles, %d-sample pulls together: %.3g samples/s (%.0f x real time per stream), %.1f ms per pull" % (
    n, chunk, total / dt, total / dt / n / 22050, dt / 8 * 1e3))
few = players[:8]
t0 = time.perf_counter()
tot2 = 0
for _ in range(4):
    for p in few:
        b = p.synthesize(chunk)
        tot2 += b.length if b is not None else 0
dt2 = time.perf_counter() - t0
print("one handle per call: %.3g samples/s (%.0f x real time), %.2f ms per 8192-sample pull" % (tot2 / dt2, tot2 / dt2 / 22050, dt2 / (4 * len(few)) * 1e3))

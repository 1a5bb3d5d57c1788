"""Live handles that are NOT aligned: the eight sampleIpa sentences at different pitches, every handle skewed by a pull of its own length
first, so that no two lanes of a wavefront dequeue or fade on the same sample -- against the same handles aligned (tools/live_bench.py's case).
Kernel ms per 8192-sample pull of all handles together, with 64 handles per wavefront ("live_alone" 1) and, up to 1536 handles, with a
wavefront per handle (the default).
    python tools/live_unaligned.py [handles]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
chunk = 8192
L = _native.load()
d = np.load(os.path.join(os.path.dirname(eng.__file__), "data", "workload_inputs.npz"), allow_pickle=True)
lines = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in d["ipa_lines"]]
rng = np.random.default_rng(3)
cases = [(label, skew, varied, alone) for label, skew, varied in (("in step, one sentence", False, False), ("eight sentences x pitches, aligned starts", False, True), ("eight sentences x pitches, skewed starts", True, True))
         for alone in ((1, 1536) if n <= 1536 else (1,))]
for label, skew, varied, alone in cases:
    assert L.speechPlayer_setGlobalOption(b"live_alone", alone) == 0
    players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
    streams = {}
    for k, p in enumerate(players):
        key = (k % len(lines) if varied else 0, 90 + 2 * ((k // 8) % 64) if varied else 120)
        if key not in streams:
            streams[key] = list(ipa.generateFramesAndTiming(lines[key[0]], basePitch=key[1], clauseType="."))
        for _ in range(4 if varied else 12):
            for fr, dd, f in streams[key]:
                p.queueFrame(fr, dd, f)
    if skew:
        for k, p in enumerate(players):      # a pull of its own length per handle: 1 .. 4000 samples
            p.synthesize(int(rng.integers(1, 4000)))
    group = eng.LiveGroup(players)
    group.pullDevice(64)
    kms = []
    for _ in range(4):
        _, _, produced = group.pullDevice(chunk)
        kms.append(L.speechPlayer_lastLiveKernelMs(0))
    print("%6d live handles, %-42s %-22s kernel %s ms per 8192-sample pull (produced %d)" % (n, label + ":", "a wavefront each" if alone > 1 else "64 per wavefront", " ".join("%.2f" % x for x in kms), int(produced.sum())), flush=True)
    for p in players:
        p.close()
    del group, players

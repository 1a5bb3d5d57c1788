"""Unrelated live handles beyond 1024: a wavefront per handle ("live_alone" large) against 64 handles per wavefront ("live_alone" 1), kernel ms of three
8192-sample pulls -- where the two policies meet (profiles/r6_live_alone.txt: near 1850 handles; the default is 1536)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, _native
L = _native.load()
d = np.load(os.path.join(os.path.dirname(eng.__file__), "data", "workload_inputs.npz"), allow_pickle=True)
lines = [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in d["ipa_lines"]]
for n in (1280, 1536, 1792, 2048):
    for alone in (1, 4096):
        assert L.speechPlayer_setGlobalOption(b"live_alone", alone) == 0
        rng = np.random.default_rng(3)
        players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
        streams = {}
        for k, p in enumerate(players):
            key = (k % 8, 90 + 2 * ((k // 8) % 64))
            if key not in streams:
                streams[key] = list(ipa.generateFramesAndTiming(lines[key[0]], basePitch=key[1], clauseType="."))
            for _ in range(4):
                for fr, dd, f in streams[key]:
                    p.queueFrame(fr, dd, f)
            p.synthesize(int(rng.integers(1, 4000)))
        group = eng.LiveGroup(players); group.pullDevice(64)
        kms = []
        for _ in range(3):
            group.pullDevice(8192); kms.append(L.speechPlayer_lastLiveKernelMs(0))
        print(n, "alone" if alone > 1 else "shared", " ".join("%.2f" % x for x in kms), flush=True)
        for p in players: p.close()
        del group, players

"""Large pulls of live handles: the two-workgroups-per-CU stream kernel (what a pull of more than CUs x 64 handles launches) against
the one-per-CU instantiation run over the same handles (speechPlayer_setGlobalOption("live_cus", huge)): kernel ms per 8192-sample pull.
    python tools/live_large.py [handles ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, _native

L = _native.load()
text = "mɑɪ næɪm ɪz mɑɪkʊl dæɪmɪən kɑɹən"
frames = list(ipa.generateFramesAndTiming(text, clauseType="."))
chunk = 8192
for n in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]:
    row = []
    for label, cus in (("auto (two per CU beyond %d handles)" % (256 * 64), 0), ("one per CU always", 1 << 20)):
        assert L.speechPlayer_setGlobalOption(b"live_cus", cus) == 0
        t0 = time.perf_counter()
        players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
        for p in players:
            for _ in range(3):
                for fr, d, f in frames:
                    p.queueFrame(fr, d, f)
        tq = time.perf_counter() - t0
        group = eng.LiveGroup(players)
        group.pullDevice(64)
        kms, total = [], 0
        t0 = time.perf_counter()
        for _ in range(4):
            _, _, produced = group.pullDevice(chunk)
            total += int(produced.sum())
            kms.append(L.speechPlayer_lastLiveKernelMs(0))
        dt = time.perf_counter() - t0
        row.append("%s: kernel %.2f ms, call %.2f ms per pull (%.3g samples/s)" % (label, float(np.median(kms)), dt / 4 * 1e3, total / dt))
        for p in players:
            p.close()
        del group, players
    print("%6d live handles x %d samples: %s" % (n, chunk, " | ".join(row)), flush=True)
L.speechPlayer_setGlobalOption(b"live_cus", 0)

"""How fast do N LIVE handles advance when pulled together (speechPlayer_synthesizeMany)?
Three figures per run: the kernel alone (HIP events around the launch), the call with the PCM left in HBM
(speechPlayer_synthesizeManyDevice: upload of the queued frames + kernel + results), and the call that hands every handle
its samples in a host buffer (PCIe copy of n x 8192 x 2 bytes + a memcpy per handle)."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = 8192
pulls = 6
text = "mɑɪ næɪm ɪz mɑɪkʊl dæɪmɪən kɑɹən"
frames = list(ipa.generateFramesAndTiming(text, clauseType="."))
L = _native.load()
L.speechPlayer_setGlobalOption(b"live_alone", 1)      # 64 handles per wavefront, as measured since round 3 (default since round 6: up to 1536 handles a wavefront each)


def fresh():
    players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
    for p in players:
        for _ in range(3):
            for fr, d, f in frames:
                p.queueFrame(fr, d, f)
    return players


t0 = time.perf_counter()
players = fresh()
tq = time.perf_counter() - t0
nq = 3 * len(frames) * n
group = eng.LiveGroup(players)                           # handle array built once
group.pullDevice(64)                                     # warm-up
t0 = time.perf_counter()
total, kms = 0, 0.0
for _ in range(pulls):
    _, _, produced = group.pullDevice(chunk)
    total += int(produced.sum())
    kms += L.speechPlayer_lastLiveKernelMs(0)
dt = time.perf_counter() - t0
print("%d handles created, %d frames queued (through ctypes): %.2f s = %.2f us per frame" % (n, nq, tq, tq / nq * 1e6), flush=True)
print("%d live handles, %d-sample pulls, layout %s: kernel %.2f ms per pull = %.3g samples/s; call with PCM left in HBM %.2f ms = %.3g samples/s" % (
    n, chunk, os.environ.get("SPEECHPLAYER_LIVE_LAYOUT", "1"), kms / pulls, total / (kms * 1e-3), dt / pulls * 1e3, total / dt), flush=True)
for p in players:
    p.close()
players = fresh()
group = eng.LiveGroup(players)
out = np.zeros((n, chunk), dtype=np.int16)
group.pull(64, out)
t0 = time.perf_counter()
total = 0
for _ in range(pulls):
    total += int(group.pull(chunk, out).sum())
dt = time.perf_counter() - t0
print("    PCM to per-handle host buffers: %.1f ms per pull = %.3g samples/s (%.0f x real time per stream)" % (dt / pulls * 1e3, total / dt, total / dt / n / 22050), flush=True)
few = players[:8]
for p in few:
    for fr, d, f in frames:
        p.queueFrame(fr, d, f)
t0 = time.perf_counter()
tot2 = 0
for _ in range(2):
    for p in few:
        b = p.synthesize(chunk)
        tot2 += b.length if b is not None else 0
dt2 = time.perf_counter() - t0
print("    one handle per call: %.2f ms per 8192-sample pull (%.0f x real time)" % (dt2 / (2 * len(few)) * 1e3, tot2 / dt2 / 22050), flush=True)

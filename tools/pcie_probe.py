"""Host-buffer (PCIe-inclusive) rate of the batch path: upload of the frame streams, kernel, read-back of all PCM.
bench.py's `value` is the HBM-resident rate; this is the note DESIGN.md section 7 quotes beside it."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, host_array, workloads

pinned = "pinned" in sys.argv[1:]      # frames and PCM in page-locked memory (speechPlayer_hostAlloc): one DMA each way
for wl in [a for a in sys.argv[1:] if a != "pinned"] or ["cfg1", "cfg2"]:
    b = workloads.make(wl)
    bp = BatchPlayer(b["sr"])
    best = None
    if pinned:
        fr = host_array(b["frames"].shape, np.float64); fr[...] = b["frames"]; b["frames"] = fr
        out = host_array(int(b.sample_counts().sum()), np.int16); out[...] = 0
    else:
        out = np.zeros(int(b.sample_counts().sum()), np.int16)      # touched once, reused: no page faults in the timed copies
    for rep in range(3):
        t0 = time.perf_counter()
        bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
        t1 = time.perf_counter()
        bp.synthesize(); bp.wait()
        t2 = time.perf_counter()
        pcm, starts = bp.readAll(out)
        t3 = time.perf_counter()
        cur = (t1 - t0, t2 - t1, t3 - t2)
        if best is None or sum(cur) < sum(best):
            best = cur
    n = bp.totalSamples
    print("%s%s: %d samples, %d frames: upload+plan %.2f ms, kernel (synchronous) %.2f ms, read all PCM to host (%.0f MB) %.2f ms"
          " -> %.3g samples/s host buffer to host buffer, %.3g samples/s kernel only" % (
              wl, " (page-locked host buffers)" if pinned else "", n, bp.totalFrames, best[0] * 1e3, best[1] * 1e3, pcm.nbytes / 1e6, best[2] * 1e3, n / sum(best), n / best[1]), flush=True)
    bp.close()

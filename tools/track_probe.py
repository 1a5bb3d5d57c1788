"""Tracks (and the flat stages they feed) on / off: kernel time and PCM digest of cfg2 as benchmarked, with rotated frame lists, with jittered
durations and without the sort by length (tools/mixed_probe.py), plus cfg3 / cfg4 on request.  The digests of a row must agree.

    python tools/track_probe.py [utterances] [+cfg3] [+cfg4] [+unsorted] [+distinct] [only=NAME]
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads
from mixed_probe import rotate, jitter, distinct


def run(b, tracks, sort=1, mode=0, launches=4, flat=0):
    bp = BatchPlayer(b["sr"], mode=mode)
    bp.setOption("sort", sort)
    bp.setOption("tracks", tracks)
    t0 = time.time()
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    host = time.time() - t0
    bp.time(1)
    ms = float(np.mean(bp.time(launches)))
    dg = bp.digest()
    info = bp.kernelInfo()
    bp.close()
    return ms, dg, host, info


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
    extra = [a[1:] for a in sys.argv[1:] if a.startswith("+")]
    base = workloads.make("cfg2", n)
    cases = [("cfg2", base, 1), ("rotated", rotate(base), 1), ("jittered", jitter(base), 1)]
    if "unsorted" in extra:
        cases.append(("unsorted", base, 0))
    if "distinct" in extra:
        cases.append(("distinct", distinct(base), 1))
        cases.append(("dist+rot", rotate(distinct(base)), 1))
    if "cfg3" in extra:
        cases.append(("cfg3", workloads.make("cfg3", 125000), 1))
    if "cfg4" in extra:
        cases.append(("cfg4", workloads.make("cfg4", 32768), 1))
    only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
    for name, b, sort in cases:
        if only and name not in only:
            continue
        off = run(b, 0, sort)
        on = run(b, 1, sort)
        same = "same PCM" if off[1] == on[1] else "PCM DIFFERS (%016x vs %016x)" % (off[1], on[1])
        print("%-9s %6d utt  untracked %7.2f ms  tracked %7.2f ms  (%.2fx)  %s  setUtterances %.2f -> %.2f s  tracked %d utt, %d tracks, %d MB" % (
            name, b.n_utt, off[0], on[0], off[0] / on[0], same, off[2], on[2], on[3].get("tracked_utterances"), on[3].get("tracks"), on[3].get("track_mbytes")), flush=True)

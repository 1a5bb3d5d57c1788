"""Per-sample latency of every kernel family for ONE utterance (a lone lane) and for 64 copies of it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads

def run(b, n, **opt):
    lists = b.slice(0, 1)
    bp = eng.BatchPlayer(22050, mode=opt.pop("mode", 0), layout=opt.pop("layout", None))
    for k, v in opt.items():
        bp.setOption(k, v)
    bp.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], np.zeros(n, np.uint32), lists["index"], lists["isnull"], np.arange(n, dtype=np.uint32))
    bp.time(3)
    ms = float(np.median(bp.time(15)))
    info = bp.kernelInfo()
    s = bp.utteranceSamples(0)
    bp.close()
    return ms, s, info

for name, b in (("cfg2 line 1 (noisy speech)", workloads.make("cfg2", 8).slice(1, 1)), ("cfg1 vowel (quiet)", workloads.make("cfg1", 1))):
    for n in (1, 64):
        for label, opt in (("tracks (flat stages)", dict()), ("no tracks, frame state machine", dict(tracks=0, direct=0)), ("direct stages", dict(tracks=0, direct=2)),
                           ("lane kernel", dict(layout=0, tracks=0, direct=0)), ("MODE_FAST no tracks", dict(tracks=0, direct=0, mode=1))):
            ms, s, info = run(b, n, **opt)
            print("%-28s n=%-3d %-34s %.3f ms for %d samples = %.0f ns/sample  [chunk %d, tracked %d, direct %d, lanepipe %d]" %
                  (name, n, label, ms, s, ms * 1e6 / s, info["stage_parallel_chunk"], info["tracked_utterances"], info["direct_utterances"], info["lane_pipelined_utterances"]))

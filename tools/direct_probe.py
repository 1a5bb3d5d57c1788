"""Direct stages (klatt_direct.h) against the stages with the frame state machine and the tracked flat stages: same PCM (MODE_EXACT: the
same bytes), and how long each takes on the batches in which nothing is shared and / or nothing is aligned.

    python tools/direct_probe.py check              # random ragged batches: legacy = direct = tracked, both modes
    python tools/direct_probe.py time [utterances]  # cfg2 / jittered / distinct / all_different: tracked, direct, legacy, both modes
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads
from mixed_probe import jitter, distinct


def run(b, tracks, direct, mode=0, launches=0, sort=1, lean=-1):
    bp = BatchPlayer(b["sr"] if "sr" in b else 22050, mode=mode)
    bp.setOption("sort", sort)
    bp.setOption("tracks", tracks)
    bp.setOption("direct", direct)
    bp.setOption("direct_lean", lean)
    t0 = time.time()
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    host = time.time() - t0
    if launches:
        bp.time(1)
        ms = float(np.median(bp.time(launches)))
        out = (ms, bp.digest(), host, bp.kernelInfo())
    else:
        bp.synthesize()
        pcm, start = bp.readAll()
        out = (pcm.copy(), start.copy(), bp.kernelInfo())
    bp.close()
    return out


def check():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from tests.test_gpu_parity import random_batch
    from tests import oracle
    for seed, wild in ((31, False), (32, True)):
        rng = np.random.default_rng(seed)
        batch = random_batch(rng, 700, quiet_fraction=0.15, wild=wild)
        exp, exp_start, total = oracle.batch_synthesize(22050, batch, threads=8)
        for mode in (0, 1):
            legacy = run(batch, 0, 0, mode)
            direct = run(batch, 0, 2, mode, lean=0)
            lean = run(batch, 0, 2, mode, lean=1)
            both = run(batch, 1, 2, mode)
            for name, r in (("legacy", legacy), ("direct", direct), ("direct lean", lean), ("tracked+direct", both)):
                d = r[0].astype(np.int32) - exp.astype(np.int32)
                print("seed %d wild %s mode %d %-15s: %d samples, %d differ from the oracle (max %d), direct utterances %d, tracked %d; equal to legacy: %s" % (
                    seed, wild, mode, name, total, int(np.count_nonzero(d)), int(np.abs(d).max()), r[2]["direct_utterances"], r[2]["tracked_utterances"],
                    np.array_equal(r[0], legacy[0])), flush=True)


def timing(n):
    base = workloads.make("cfg2", n)
    cases = [("cfg2", base), ("jittered", jitter(base)), ("distinct", distinct(base)), ("all_different", jitter(distinct(base)))]
    only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
    for name, b in cases:
        if only and name not in only:
            continue
        for mode in (0, 1):
            row = []
            for label, tracks, direct, lean in (("tracked", 1, 2, -1), ("direct", 0, 2, 0), ("lean", 0, 2, 1), ("legacy", 0, 0, -1), ("auto", 1, 1, -1)):
                if label in ("legacy", "auto") and "+legacy" not in sys.argv:
                    continue
                if label == "tracked" and "-tracked" in sys.argv:
                    continue
                ms, dg, host, info = run(b, tracks, direct, mode, launches=5, lean=lean)
                row.append("%s %7.2f ms (%016x; direct %d, tracked %d utt; set %.2f s)" % (label, ms, dg, info["direct_utterances"], info["tracked_utterances"], host))
            print("%-14s mode %d: %s" % (name, mode, " | ".join(row)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 65536)
    else:
        check()

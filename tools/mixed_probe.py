"""How much slower is a batch whose lanes do NOT fade at the same time?  (The BASELINE recipes repeat eight
sentences, so after the sort by length every wavefront holds 64 copies of one sentence: all lanes fade together.)
  sorted     : cfg2 as benchmarked
  unsorted   : the same batch with the sort by length off (ragged lengths + de-aligned fades)
  staggered  : sorted, but every utterance starts with 0..200 ms of silence of its own (the sort by length puts
               similar delays side by side again: lanes end up skewed by ~100 samples only)
  rotated    : sorted, every utterance's frame list rotated by a random amount (same lengths, fully de-aligned
               fades) -- the stand-in for 64 different sentences per wave
  jittered   : every frame's duration and fade scaled by its own random factor in [0.7, 1.3]: no two utterances of the batch
               share a timing or a length (what a batch of unrelated sentences looks like to the kernel)
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nvspeechplayer_amd import BatchPlayer, workloads


def stagger(b, seed=1):
    rng = np.random.default_rng(seed)
    n = b.n_utt
    fs = b["frame_start"]
    counts = np.diff(fs) + 1
    new_fs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    nF = int(new_fs[-1])
    head = new_fs[:-1]
    keep = np.ones(nF, bool); keep[head] = False
    out = {}
    for k, fill in (("frames", 0.0), ("min", 0), ("fade", 0), ("index", -1), ("isnull", 1)):
        a = np.zeros((nF,) + b[k].shape[1:], dtype=b[k].dtype)
        a[keep] = b[k]
        a[head] = fill
        out[k] = a
    out["min"][head] = rng.integers(1, 4410, n).astype(np.uint32)
    out["fade"][head] = 1
    return workloads.Batch(frame_start=new_fs, seeds=b["seeds"], name=b["name"] + " staggered", sr=b["sr"], **out)


rotate = workloads.rotated


jitter = workloads.jittered


distinct = workloads.distinct


def run(name, b, sort, mode=0):
    bp = BatchPlayer(b["sr"], mode=mode)
    bp.setOption("sort", sort)
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    bp.time(1)
    ms = float(np.mean(bp.time(3)))
    print("%-10s %d utterances: %7.2f ms  %.3g samples/s" % (name, b.n_utt, ms, bp.totalSamples / ms * 1e3), flush=True)
    bp.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    b = workloads.make("cfg2", n)
    run("sorted", b, 1)
    run("unsorted", b, 0)
    run("staggered", stagger(b), 1)
    run("rotated", rotate(b), 1)
    run("jittered", jitter(b), 1)

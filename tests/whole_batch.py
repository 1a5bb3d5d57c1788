"""Every utterance of a full-size batch against the oracle (a strided sample of 40 of 65 536 utterances leaves the lane
packing, the track sharing and the six groups of a real batch unchecked in the other 99.94 %).

The engine's PCM stays in HBM: speechPlayer_batch_digest gives 8 bytes per utterance.  The oracle synthesises the same utterances on
the host's cores, piece by piece (the whole of configs[2] is 3 GB of PCM), tests/native/pcm_digest.c computes the same digest of the
oracle's PCM, and the two arrays are compared.  MODE_EXACT's exp / cos are within 1 ulp of glibc's, so a sample on a truncation
boundary may differ by one LSB (tests/test_gpu_parity.py, compare): an utterance whose digests differ is read back and held to that
bar, and the number of such utterances is bounded.  Test infrastructure: loads the oracle.
"""
import ctypes
import os
import subprocess

import numpy as np

from tests import oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _digest_lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "native", "pcm_digest.c")
        out = os.path.join(_HERE, "native", "libpcm_digest.so")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            tmp = out + ".tmp.%d" % os.getpid()
            subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", tmp, src])
            os.replace(tmp, out)
        _lib = ctypes.CDLL(out)
        _lib.pcm_digest_many.restype = None
        _lib.pcm_digest_many.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    return _lib


def host_digests(pcm, starts):
    """Per-utterance digests of host PCM (int16, utterance u = pcm[starts[u]:starts[u + 1]])."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    st = np.ascontiguousarray(starts, dtype=np.int64)
    out = np.zeros(max(len(st) - 1, 1), dtype=np.uint64)
    _digest_lib().pcm_digest_many(pcm.ctypes.data, st.ctypes.data, len(st) - 1, out.ctypes.data)
    return out[:len(st) - 1]


def subset(batch, sel):
    """The utterances `sel` (ascending indices) of a batch as a batch of their own (vectorised gather of their frames)."""
    fs = np.asarray(batch["frame_start"], dtype=np.int64)
    sel = np.asarray(sel, dtype=np.int64)
    counts = fs[sel + 1] - fs[sel]
    new_fs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rows = np.repeat(fs[sel] - new_fs[:-1], counts) + np.arange(int(new_fs[-1]), dtype=np.int64)
    return dict(frames=batch["frames"][rows], min=batch["min"][rows], fade=batch["fade"][rows], index=batch["index"][rows],
                isnull=batch["isnull"][rows], frame_start=new_fs, seeds=np.asarray(batch["seeds"])[sel])


def check_against_oracle(bp, batch, device_digests, compare, name, stride=1, piece=8192, threads=None, max_differing=None):
    """Utterances 0, stride, 2 stride, ... of `batch` (already synthesised on `bp`, whose per-utterance digests are `device_digests`):
    digest and index mark of every one against the oracle's.  `compare(got, exp, name)`: the sample-level bar for an utterance whose
    digests differ.  Returns (utterances checked, utterances whose digests differed)."""
    sr = bp.sampleRate
    threads = threads or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            threads = max(1, min(threads, int(float(quota) / float(period))))
    except Exception:
        pass
    sel_all = np.arange(0, batch.n_utt if hasattr(batch, "n_utt") else len(batch["frame_start"]) - 1, stride, dtype=np.int64)
    differing = []
    for a in range(0, len(sel_all), piece):
        sel = sel_all[a:a + piece]
        sub = subset(batch, sel)
        exp, exp_start, _ = oracle.batch_synthesize(sr, sub, threads=threads)
        want = host_digests(exp, exp_start)
        got = np.asarray(device_digests)[sel]
        marks = oracle.batch_last_index(sr, sub, threads=threads)
        for j in np.flatnonzero(got != want):
            u = int(sel[j])
            compare(bp.read(u), exp[exp_start[j]:exp_start[j + 1]], "%s utt %d (digests differ)" % (name, u))
            differing.append(u)
        for j, u in enumerate(sel):
            assert bp.getLastIndex(int(u)) == int(marks[j]), (name, int(u))      # reference src/frame.cpp:69, :117-119
    if max_differing is None:
        max_differing = max(3, len(sel_all) // 2000)
    assert len(differing) <= max_differing, "%s: %d of %d utterances differ from the oracle by one-LSB flips: %s" % (name, len(differing), len(sel_all), differing[:20])
    return len(sel_all), len(differing)

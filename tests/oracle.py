"""ctypes binding of oracle/libklatt_oracle.so -- the CPU checker.

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libklatt_oracle.so")

NOISE_LIBC = 0
NOISE_COUNTER = 1
NP = 47

_lib = None


def build(force=False):
    src = os.path.join(ORACLE_DIR, "klatt_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        vp, u32, i32, i64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_longlong
        L.oracle_initialize.restype = vp
        L.oracle_initialize.argtypes = [i32]
        L.oracle_setNoise.argtypes = [vp, i32, u32]
        L.oracle_setNoise.restype = None
        L.oracle_queueFrame.argtypes = [vp, vp, u32, u32, i32, i32]
        L.oracle_queueFrame.restype = None
        L.oracle_synthesize.argtypes = [vp, u32, vp]
        L.oracle_synthesize.restype = i32
        L.oracle_getLastIndex.argtypes = [vp]
        L.oracle_getLastIndex.restype = i32
        L.oracle_terminate.argtypes = [vp]
        L.oracle_terminate.restype = None
        L.oracle_utteranceLength.argtypes = [vp, vp, u32]
        L.oracle_utteranceLength.restype = i64
        L.klatt_noise31.argtypes = [u32, u32]
        L.klatt_noise31.restype = u32
        L.oracle_batchSynthesize.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32]
        L.oracle_batchSynthesize.restype = i64
        _lib = L
    return _lib


class OraclePlayer:
    """Same call surface as the reference's speechPlayer.SpeechPlayer, but durations are
    given in SAMPLES (the C-ABI unit) to keep tests independent of ms rounding."""

    def __init__(self, sample_rate, noise=NOISE_COUNTER, seed=0):
        self.L = lib()
        self.h = self.L.oracle_initialize(sample_rate)
        self.L.oracle_setNoise(self.h, noise, seed)

    def queue(self, frame, min_samples, fade_samples, user_index=-1, purge=False):
        if frame is None:
            ptr = None
        else:
            buf = np.ascontiguousarray(frame, dtype=np.float64)
            assert buf.shape == (NP,)
            ptr = buf.ctypes.data
        self.L.oracle_queueFrame(self.h, ptr, int(min_samples), int(fade_samples), int(user_index), int(bool(purge)))

    def synthesize(self, n):
        out = np.zeros(n, dtype=np.int16)
        got = self.L.oracle_synthesize(self.h, n, out.ctypes.data)
        return out[:got]

    def drain(self, chunk=8192):
        parts = []
        while True:
            p = self.synthesize(chunk)
            parts.append(p)
            if len(p) < chunk:
                break
        return np.concatenate(parts) if parts else np.zeros(0, np.int16)

    def last_index(self):
        return self.L.oracle_getLastIndex(self.h)

    def close(self):
        if self.h:
            self.L.oracle_terminate(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def utterance_length(min_samples, fade_samples):
    m = np.ascontiguousarray(min_samples, dtype=np.uint32)
    f = np.ascontiguousarray(fade_samples, dtype=np.uint32)
    return lib().oracle_utteranceLength(m.ctypes.data, f.ctypes.data, len(m))


def batch_synthesize(sample_rate, batch, first=0, count=None, threads=1):
    """batch: dict with frames[nF,47] f64, min[nF] u32, fade[nF] u32, index[nF] i32, isnull[nF] u8,
    frame_start[nU+1] i64, seeds[nU] u32.  Returns (pcm int16 concatenated, out_start[nU+1])."""
    fs = np.ascontiguousarray(batch["frame_start"], dtype=np.int64)
    n_utt = len(fs) - 1
    if count is None:
        count = n_utt - first
    m = np.ascontiguousarray(batch["min"], dtype=np.uint32)
    f = np.ascontiguousarray(batch["fade"], dtype=np.uint32)
    fe = np.maximum(f, 1).astype(np.int64)
    per_frame = np.maximum(m.astype(np.int64), fe + 1) + 1
    csum = np.concatenate([[0], np.cumsum(per_frame)])
    lens = csum[fs[1:]] - csum[fs[:-1]]
    out_start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.zeros(int(out_start[-1]), dtype=np.int16)
    frames = np.ascontiguousarray(batch["frames"], dtype=np.float64)
    idx = np.ascontiguousarray(batch["index"], dtype=np.int32)
    isn = np.ascontiguousarray(batch["isnull"], dtype=np.uint8)
    seeds = np.ascontiguousarray(batch["seeds"], dtype=np.uint32)
    total = lib().oracle_batchSynthesize(sample_rate, frames.ctypes.data, m.ctypes.data, f.ctypes.data,
                                         idx.ctypes.data, isn.ctypes.data, fs.ctypes.data, seeds.ctypes.data,
                                         out_start.ctypes.data, pcm.ctypes.data, first, count, threads)
    return pcm, out_start, total


def batch_last_index(sample_rate, batch, first=0, count=None, threads=1):
    """What getLastIndex() answers for each utterance of `batch` once it has been pulled to its end (the oracle's own frame
    state machine, reference src/frame.cpp:69, :117-119)."""
    fs = np.ascontiguousarray(batch["frame_start"], dtype=np.int64)
    n_utt = len(fs) - 1
    if count is None:
        count = n_utt - first
    m = np.ascontiguousarray(batch["min"], dtype=np.uint32)
    f = np.ascontiguousarray(batch["fade"], dtype=np.uint32)
    frames = np.ascontiguousarray(batch["frames"], dtype=np.float64)
    idx = np.ascontiguousarray(batch["index"], dtype=np.int32)
    isn = np.ascontiguousarray(batch["isnull"], dtype=np.uint8)
    seeds = np.ascontiguousarray(batch["seeds"], dtype=np.uint32)
    out = np.full(count, -2, dtype=np.int32)
    L = lib()
    L.oracle_batchLastIndex.restype = ctypes.c_longlong
    L.oracle_batchLastIndex.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
    L.oracle_batchLastIndex(sample_rate, frames.ctypes.data, m.ctypes.data, f.ctypes.data, idx.ctypes.data, isn.ctypes.data,
                            fs.ctypes.data, seeds.ctypes.data, out.ctypes.data, first, count, threads)
    return out

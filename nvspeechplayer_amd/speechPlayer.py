"""Host-side mirror of the reference's ctypes wrapper, over the MI355X engine.

`Frame` and `SpeechPlayer` keep the reference wrapper's names, arguments and
behaviour (reference speechPlayer.py:20-68): durations in milliseconds converted
with the same truncation, `synthesize` returning a ctypes short array carrying
`.length`, or None when nothing was produced.  `BatchPlayer` is the additive batch
interface (include/speechPlayer_batch.h): many frame streams, one kernel launch.

Everything here calls the HIP library through its C-ABI; there is no CPU path.
"""
from ctypes import POINTER, Structure, byref, c_double, c_int, c_short, c_void_p, cast
from ctypes import c_longlong as ctypes_longlong

import numpy as np

from . import _native

speechPlayer_frameParam_t = c_double

FRAME_FIELDS = [
    'voicePitch',
    'vibratoPitchOffset',
    'vibratoSpeed',
    'voiceTurbulenceAmplitude',
    'glottalOpenQuotient',
    'voiceAmplitude',
    'aspirationAmplitude',
    'cf1', 'cf2', 'cf3', 'cf4', 'cf5', 'cf6', 'cfN0', 'cfNP',
    'cb1', 'cb2', 'cb3', 'cb4', 'cb5', 'cb6', 'cbN0', 'cbNP',
    'caNP',
    'fricationAmplitude',
    'pf1', 'pf2', 'pf3', 'pf4', 'pf5', 'pf6',
    'pb1', 'pb2', 'pb3', 'pb4', 'pb5', 'pb6',
    'pa1', 'pa2', 'pa3', 'pa4', 'pa5', 'pa6',
    'parallelBypass',
    'preFormantGain',
    'outputGain',
    'endVoicePitch',
]


class Frame(Structure):
    """47 doubles, field order = the C struct (include/speechPlayer.h; reference speechPlayer.py:20-40)."""
    _fields_ = [(name, speechPlayer_frameParam_t) for name in FRAME_FIELDS]

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float64).copy()

    @classmethod
    def from_array(cls, values):
        f = cls()
        for name, v in zip(FRAME_FIELDS, values):
            setattr(f, name, float(v))
        return f


class SpeechPlayer(object):
    """One live stream (reference speechPlayer.py:44-68)."""

    def __init__(self, sampleRate, noiseSeed=None):
        self.sampleRate = sampleRate
        self._dll = _native.load()
        self._speechHandle = self._dll.speechPlayer_initialize(sampleRate)
        if not self._speechHandle:
            raise RuntimeError("speechPlayer_initialize failed: %s" % _native.last_error())
        if noiseSeed is not None:
            self._dll.speechPlayer_setNoiseSeed(self._speechHandle, int(noiseSeed))

    def queueFrame(self, frame, minFrameDuration, fadeDuration, userIndex=-1, purgeQueue=False):
        frame = byref(frame) if frame else None
        self._dll.speechPlayer_queueFrame(self._speechHandle, frame,
                                          int(minFrameDuration * (self.sampleRate / 1000.0)),
                                          int(fadeDuration * (self.sampleRate / 1000.0)), userIndex, purgeQueue)

    def queueFrameSamples(self, frame, minSamples, fadeSamples, userIndex=-1, purgeQueue=False):
        """Same call with durations already in samples (the C-ABI unit)."""
        frame = byref(frame) if frame else None
        self._dll.speechPlayer_queueFrame(self._speechHandle, frame, int(minSamples), int(fadeSamples), userIndex, purgeQueue)

    def synthesize(self, numSamples):
        buf = (c_short * numSamples)()
        res = self._dll.speechPlayer_synthesize(self._speechHandle, numSamples, buf)
        if res > 0:
            buf.length = min(res, len(buf))
            return buf
        else:
            return None

    def getLastIndex(self):
        return self._dll.speechPlayer_getLastIndex(self._speechHandle)

    @staticmethod
    def synthesizeMany(players, numSamples, out=None):
        """Advance many live players together in one kernel launch (speechPlayer_synthesizeMany).
        Returns a list with what each player's synthesize(numSamples) would have returned.  With `out` (a C-contiguous int16
        array [len(players), >= numSamples]) the samples land in its rows instead and the return value is the array of
        per-player sample counts (no buffer is allocated per player and pull)."""
        n = len(players)
        if n == 0:
            return []
        dll = players[0]._dll
        handles = (c_void_p * n)(*[p._speechHandle for p in players])
        produced = (c_int * n)()
        if out is not None:
            if out.dtype != np.int16 or out.ndim != 2 or out.shape[0] < n or out.shape[1] < numSamples or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous int16 array [n, >= numSamples]")
            ptrs = (out.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(out.strides[0])).astype(np.uint64)
            rc = dll.speechPlayer_synthesizeMany(handles, n, numSamples, ptrs.ctypes.data, produced)
            if rc != 0:
                raise RuntimeError("speechPlayer_synthesizeMany failed: %s" % _native.last_error())
            return np.frombuffer(produced, dtype=np.int32).copy()
        bufs = [(c_short * numSamples)() for _ in range(n)]
        ptrs = (c_void_p * n)(*[cast(b, c_void_p) for b in bufs])
        rc = dll.speechPlayer_synthesizeMany(handles, n, numSamples, ptrs, produced)
        if rc != 0:
            raise RuntimeError("speechPlayer_synthesizeMany failed: %s" % _native.last_error())
        res = []
        for b, got in zip(bufs, produced):
            if got > 0:
                b.length = min(got, len(b))
                res.append(b)
            else:
                res.append(None)
        return res

    @staticmethod
    def synthesizeManyDevice(players, numSamples):
        """The same, PCM left in HBM (speechPlayer_synthesizeManyDevice): -> (device pointer, row stride in samples,
        produced[n]).  Row i holds produced[i] samples of players[i]."""
        n = len(players)
        dll = players[0]._dll
        handles = (c_void_p * n)(*[p._speechHandle for p in players])
        produced = (c_int * n)()
        ptr = c_void_p()
        stride = ctypes_longlong()
        rc = dll.speechPlayer_synthesizeManyDevice(handles, n, numSamples, byref(ptr), byref(stride), produced)
        if rc != 0:
            raise RuntimeError("speechPlayer_synthesizeManyDevice failed: %s" % _native.last_error())
        return ptr.value, stride.value, np.frombuffer(produced, dtype=np.int32).copy()

    def close(self):
        if getattr(self, "_speechHandle", None):
            self._dll.speechPlayer_terminate(self._speechHandle)
            self._speechHandle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LiveGroup(object):
    """A fixed set of live players pulled together again and again: the handle array and the result arrays are built once
    (SpeechPlayer.synthesizeMany builds them per call -- about a millisecond for 8192 players, as much as the engine's own host
    work per pull)."""

    def __init__(self, players):
        self.players = list(players)
        n = len(self.players)
        if n == 0:
            raise ValueError("no players")
        self._dll = self.players[0]._dll
        self._handles = (c_void_p * n)(*[p._speechHandle for p in self.players])
        self._produced = (c_int * n)()
        self.produced = np.frombuffer(self._produced, dtype=np.int32)      # a view: overwritten by the next pull
        self._ptr = c_void_p()
        self._stride = ctypes_longlong()
        self._out = None
        self._rows = None

    def pullDevice(self, numSamples):
        """-> (device pointer, row stride in samples, produced[n]); the PCM stays in HBM (speechPlayer_synthesizeManyDevice)."""
        rc = self._dll.speechPlayer_synthesizeManyDevice(self._handles, len(self.players), numSamples, byref(self._ptr), byref(self._stride),
                                                         self._produced)
        if rc != 0:
            raise RuntimeError("speechPlayer_synthesizeManyDevice failed: %s" % _native.last_error())
        return self._ptr.value, self._stride.value, self.produced

    def pull(self, numSamples, out):
        """Samples into the rows of `out` (C-contiguous int16 [n, >= numSamples]); -> produced[n]."""
        n = len(self.players)
        if self._out is not out:
            if out.dtype != np.int16 or out.ndim != 2 or out.shape[0] < n or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous int16 array [n, >= numSamples]")
            self._rows = (out.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(out.strides[0])).astype(np.uint64)
            self._out = out
        if out.shape[1] < numSamples:
            raise ValueError("out has fewer than numSamples columns")
        rc = self._dll.speechPlayer_synthesizeMany(self._handles, n, numSamples, self._rows.ctypes.data, self._produced)
        if rc != 0:
            raise RuntimeError("speechPlayer_synthesizeMany failed: %s" % _native.last_error())
        return self.produced


def setGlobalOption(name, value):
    """speechPlayer_setGlobalOption (include/speechPlayer_batch.h): process-wide options of the live handles -- "live_mode" (arithmetic mode
    of handles created from now on), "live_alone" (up to how many handles of a pull get a wavefront each), "live_layout", "live_cus",
    "live_replicate", "live_trim"."""
    if _native.load().speechPlayer_setGlobalOption(name.encode() if isinstance(name, str) else name, int(value)) != 0:
        raise ValueError("speechPlayer_setGlobalOption(%r, %r) refused" % (name, value))


def pcm_digest(pcm):
    """speechPlayer_batch_digest's per-utterance value for a host int16 array (uint64 arithmetic wraps)."""
    v = np.asarray(pcm, dtype=np.int16).view(np.uint16).astype(np.uint64)
    pos = np.arange(len(v), dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (pos + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) ^ (v + np.uint64(1)) * np.uint64(0xC2B2AE3D27D4EB4F)
        x ^= x >> np.uint64(29); x *= np.uint64(0xBF58476D1CE4E5B9); x ^= x >> np.uint64(32)
        return int(x.sum(dtype=np.uint64))


class _HostBlock(object):
    """Owner of one speechPlayer_hostAlloc block: frees it when the last array view is gone."""
    def __init__(self, dll, nbytes):
        self._dll, self.ptr = dll, dll.speechPlayer_hostAlloc(nbytes)
        if not self.ptr:
            raise MemoryError("speechPlayer_hostAlloc(%d): %s" % (nbytes, _native.last_error()))

    def __del__(self):
        try:
            if self.ptr:
                self._dll.speechPlayer_hostFree(self.ptr)
                self.ptr = None
        except Exception:
            pass


def host_array(shape, dtype):
    """A numpy array in page-locked host memory (speechPlayer_hostAlloc): frames handed to setUtterances from such an array, and PCM
    read into one (readAll / readAllAsync), cross the link as one DMA at its full rate.  Freed with the array."""
    import ctypes
    dt = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    block = _HostBlock(_native.load(), max(n * dt.itemsize, 1))
    raw = (ctypes.c_char * max(n * dt.itemsize, 1)).from_address(block.ptr)
    raw._block = block      # the array's base is `raw`: the block lives as long as any view of the array
    return np.frombuffer(raw, dtype=dt, count=n).reshape(shape)


class BatchPlayer(object):
    """N independent utterances per launch (include/speechPlayer_batch.h)."""

    def __init__(self, sampleRate, device=-1, mode=0, layout=None):
        """mode: SPEECHPLAYER_MODE_EXACT (0) or _FAST (1); layout: None / -1 = chosen per batch (default),
        1 = stage-parallel workgroups, 0 = one wavefront per 64 utterances."""
        self.sampleRate = sampleRate
        self._dll = _native.load()
        self._h = self._dll.speechPlayer_batch_create(sampleRate, device)
        if not self._h:
            raise RuntimeError("speechPlayer_batch_create failed: %s" % _native.last_error())
        self._check(self._dll.speechPlayer_batch_setOption(self._h, b"mode", mode))
        if layout is not None:
            self._check(self._dll.speechPlayer_batch_setOption(self._h, b"layout", layout))
        self.nUtterances = 0

    def _check(self, rc):
        if rc is None or rc < 0:
            raise RuntimeError("speechPlayer batch call failed: %s" % _native.last_error())
        return rc

    def setOption(self, name, value):
        self._check(self._dll.speechPlayer_batch_setOption(self._h, name.encode(), int(value)))

    def setUtterances(self, frameStart, frames, minSamples, fadeSamples, userIndex=None, isNull=None, noiseSeed=None):
        fs = np.ascontiguousarray(frameStart, dtype=np.int64)
        fr = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 47)
        m = np.ascontiguousarray(minSamples, dtype=np.uint32)
        f = np.ascontiguousarray(fadeSamples, dtype=np.uint32)
        n_utt = len(fs) - 1
        assert fs[-1] == len(fr) == len(m) == len(f)
        ix = None if userIndex is None else np.ascontiguousarray(userIndex, dtype=np.int32)
        nu = None if isNull is None else np.ascontiguousarray(isNull, dtype=np.uint8)
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        p = lambda a: None if a is None else a.ctypes.data
        self._check(self._dll.speechPlayer_batch_setUtterances(self._h, n_utt, p(fs), p(fr), p(m), p(f), p(ix), p(nu), p(sd)))
        self.nUtterances = n_utt

    def setUtterancesShared(self, listStart, frames, minSamples, fadeSamples, listOf, userIndex=None, isNull=None, noiseSeed=None):
        """Frame lists that utterances share (speechPlayer_batch_setUtterancesShared): `listStart`/frames/... describe the lists as
        setUtterances describes utterances; utterance u speaks list listOf[u] with noise seed noiseSeed[u]."""
        ls = np.ascontiguousarray(listStart, dtype=np.int64)
        fr = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 47)
        m = np.ascontiguousarray(minSamples, dtype=np.uint32)
        f = np.ascontiguousarray(fadeSamples, dtype=np.uint32)
        lo = np.ascontiguousarray(listOf, dtype=np.uint32)
        assert ls[-1] == len(fr) == len(m) == len(f)
        ix = None if userIndex is None else np.ascontiguousarray(userIndex, dtype=np.int32)
        nu = None if isNull is None else np.ascontiguousarray(isNull, dtype=np.uint8)
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        p = lambda a: None if a is None else a.ctypes.data
        self._check(self._dll.speechPlayer_batch_setUtterancesShared(self._h, len(ls) - 1, p(ls), p(fr), p(m), p(f), p(ix), p(nu), len(lo), p(lo), p(sd)))
        self.nUtterances = len(lo)

    def setRecords(self, shapes, listStart, records, listOf=None, noiseSeed=None):
        """The batch in compact form (speechPlayer_batch_setRecords): `shapes` [nShapes, 47] f64, `records` a structured array of
        nvspeechplayer_amd.ipa.RECORD_DTYPE (32 bytes per frame), lists and listOf as in setUtterancesShared (None: utterance u = list u)."""
        from .ipa import RECORD_DTYPE
        sh = np.ascontiguousarray(shapes, dtype=np.float64).reshape(-1, 47)
        ls = np.ascontiguousarray(listStart, dtype=np.int64)
        rc = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        assert ls[-1] == len(rc)
        lo = None if listOf is None else np.ascontiguousarray(listOf, dtype=np.uint32)
        n_utt = len(ls) - 1 if lo is None else len(lo)
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        p = lambda a: None if a is None else a.ctypes.data
        self._check(self._dll.speechPlayer_batch_setRecords(self._h, len(sh), p(sh), len(ls) - 1, p(ls), p(rc), n_utt, p(lo), p(sd)))
        self.nUtterances = n_utt

    def frames(self, u):
        """The frames of utterance u as they are resident in HBM (speechPlayer_batch_frames): -> (frames[n, 47], min[n], fade[n], index[n], isnull[n])."""
        n = self._check(self._dll.speechPlayer_batch_frames(self._h, u, None, None, None, None, None, 0))
        fr = np.zeros((n, 47)); m = np.zeros(n, np.uint32); f = np.zeros(n, np.uint32); ix = np.zeros(n, np.int32); nu = np.zeros(n, np.uint8)
        if n:
            self._check(self._dll.speechPlayer_batch_frames(self._h, u, fr.ctypes.data, m.ctypes.data, f.ctypes.data, ix.ctypes.data, nu.ctypes.data, n))
        return fr, m, f, ix, nu

    def setIpa(self, texts, speed=1, basePitch=100, inflection=0.5, clauseType=None, noiseSeed=None, voice=None,
               trailing_silence_ms=150.0, textOf=None):
        """Text in, batch ready (speechPlayer_batch_setIpa / _setIpaVoices): every IPA string becomes one utterance through the native
        frame producer followed by 150 ms of silence, as reference test_speakIpa.py:24-27 queues them.  basePitch and
        clauseType may be sequences (one per utterance); voice: one of nvspeechplayer_amd.ipa.voices() (or a voice defined with
        ipa.defineVoice), or a sequence of voice INDICES, one per utterance (-1: none).  textOf: utterance u speaks texts[textOf[u]]
        (a batch that repeats few sentences hands them over once)."""
        from .ipa import _text_pointers
        ptrs, n, keep = _text_pointers(texts, textOf)
        pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
        code = lambda c: 0 if not c else ord(c[0])
        clauses = bytes([code(clauseType)]) * n if (clauseType is None or isinstance(clauseType, str)) else bytes(code(c) for c in clauseType)
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        tail = -1.0 if trailing_silence_ms is None else float(trailing_silence_ms)
        if voice is None or isinstance(voice, str):
            self._check(self._dll.speechPlayer_batch_setIpa(self._h, n, ptrs.ctypes.data, float(speed), pitch.ctypes.data, float(inflection), clauses + b"\0",
                                                            None if not voice else voice.encode("utf8"), tail, None if sd is None else sd.ctypes.data))
        else:
            vo = np.ascontiguousarray(np.broadcast_to(np.asarray(voice, dtype=np.int32), (n,)))
            self._check(self._dll.speechPlayer_batch_setIpaVoices(self._h, n, ptrs.ctypes.data, float(speed), pitch.ctypes.data, float(inflection), clauses + b"\0",
                                                                  vo.ctypes.data, tail, None if sd is None else sd.ctypes.data))
        del keep
        self.nUtterances = n

    def setText(self, texts, speed=1, basePitch=100, inflection=0.5, noiseSeed=None, voice=None, espeakVoice="en"):
        """Plain text in (speechPlayer_batch_setText): each text becomes one utterance the way the NVDA driver speaks it -- clauses
        through eSpeak NG's text-to-IPA, the frame producer per clause, the pause after the last clause.  Needs libespeak-ng at
        run time (nvspeechplayer_amd.ipa.textAvailable()); raises RuntimeError with the reason when it is not there."""
        import ctypes
        n = len(texts)
        enc = [t.encode("utf8") for t in texts]
        ptrs = (ctypes.c_char_p * max(n, 1))(*enc)
        pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        self._check(self._dll.speechPlayer_batch_setText(self._h, n, ptrs, None if not espeakVoice else espeakVoice.encode("utf8"), float(speed),
                                                         pitch.ctypes.data, float(inflection), None if not voice else voice.encode("utf8"),
                                                         None if sd is None else sd.ctypes.data))
        self.nUtterances = n

    @property
    def totalSamples(self):
        return self._dll.speechPlayer_batch_totalSamples(self._h)

    @property
    def totalFrames(self):
        return self._dll.speechPlayer_batch_totalFrames(self._h)

    def utteranceSamples(self, u):
        return self._dll.speechPlayer_batch_utteranceSamples(self._h, u)

    def synthesize(self, wait=True):
        self._check(self._dll.speechPlayer_batch_synthesize(self._h))
        if wait:
            self._check(self._dll.speechPlayer_batch_wait(self._h))

    def wait(self):
        self._check(self._dll.speechPlayer_batch_wait(self._h))

    def read(self, u):
        n = self.utteranceSamples(u)
        buf = np.zeros(max(n, 1), dtype=np.int16)
        got = self._check(self._dll.speechPlayer_batch_read(self._h, u, buf.ctypes.data, n))
        return buf[:got]

    def readFloat(self, u):
        """Utterance u as float32 in [-1, 1] (int16 / 32767, converted on the device)."""
        n = self.utteranceSamples(u)
        buf = np.zeros(max(n, 1), dtype=np.float32)
        got = self._check(self._dll.speechPlayer_batch_readFloat(self._h, u, buf.ctypes.data, n))
        return buf[:got]

    def writeWav(self, u, path):
        """Utterance u as a 16-bit mono WAV file at the batch's sample rate."""
        import wave
        pcm = self.read(u)
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(self.sampleRate)
            w.writeframes(pcm.astype("<i2").tobytes())

    def readAll(self, out=None):
        """All utterances' PCM, concatenated, and the nUtterances + 1 start offsets.  `out`: an int16 array of at
        least totalSamples to fill instead of a new one (a reused buffer avoids the page faults of a fresh one)."""
        total = self.totalSamples
        if out is not None:
            if out.dtype != np.int16 or not out.flags["C_CONTIGUOUS"] or out.size < total:
                raise ValueError("readAll: out must be a contiguous int16 array of at least %d samples" % total)
            buf = out
        else:
            buf = np.empty(max(total, 1), dtype=np.int16)
        starts = np.zeros(self.nUtterances + 1, dtype=np.int64)
        got = self._check(self._dll.speechPlayer_batch_readAll(self._h, buf.ctypes.data, total, starts.ctypes.data))
        return buf[:got], starts

    def readAllAsync(self, out):
        """readAll without waiting (speechPlayer_batch_readAllAsync): `out` must be page-locked (nvspeechplayer_amd.host_array); the
        compaction and the copy are queued behind the synthesis and run beside whatever is launched next.  Returns (view of `out`, starts);
        the samples are there after readWait()."""
        total = self.totalSamples
        if out.dtype != np.int16 or not out.flags["C_CONTIGUOUS"] or out.size < total:
            raise ValueError("readAllAsync: out must be a contiguous int16 array of at least %d samples" % total)
        starts = np.zeros(self.nUtterances + 1, dtype=np.int64)
        got = self._check(self._dll.speechPlayer_batch_readAllAsync(self._h, out.ctypes.data, out.size, starts.ctypes.data))
        return out[:got], starts

    def readWait(self):
        self._check(self._dll.speechPlayer_batch_readWait(self._h))

    def digest(self, per_utterance=False):
        """Digest of the whole PCM pool, computed on the device (speechPlayer_batch_digest); with per_utterance also the
        array of per-utterance digests.  `pcm_digest` below is the same formula on a host array."""
        import ctypes
        whole = ctypes.c_ulonglong(0)
        per = np.zeros(max(self.nUtterances, 1), dtype=np.uint64) if per_utterance else None
        self._check(self._dll.speechPlayer_batch_digest(self._h, None if per is None else per.ctypes.data, ctypes.byref(whole)))
        return (int(whole.value), per[:self.nUtterances]) if per_utterance else int(whole.value)

    def getLastIndex(self, u):
        return self._dll.speechPlayer_batch_getLastIndex(self._h, u)

    def time(self, launches):
        ms = np.zeros(launches, dtype=np.float32)
        self._check(self._dll.speechPlayer_batch_time(self._h, launches, ms.ctypes.data))
        return ms

    def kernelInfo(self):
        info = np.zeros(20, dtype=np.int32)
        self._check(self._dll.speechPlayer_batch_kernelInfo(self._h, info.ctypes.data, len(info)))
        return dict(vgprs=int(info[0]), lds_bytes=int(info[1]), wavefronts=int(info[2]), cus=int(info[3]),
                    workgroups_per_cu_by_lds=int(info[4]), scratch_bytes=int(info[5]),
                    stage_parallel_chunk=int(info[6]), noisy_group=bool(info[7]),
                    lane_pipelined=bool(info[8]), lane_pipelined_utterances=int(info[9]),
                    nasal_free=bool(info[10]), nasal_free_utterances=int(info[11]),
                    tracked_utterances=int(info[12]), tracks=int(info[13]), track_mbytes=int(info[14]), tracked=bool(info[15]),
                    direct_utterances=int(info[16]), direct=bool(info[17]), direct_mbytes=int(info[18]))

    def devicePcm(self):
        return self._dll.speechPlayer_batch_devicePcm(self._h)

    def deviceOffset(self, u):
        return self._dll.speechPlayer_batch_deviceOffset(self._h, u)

    def close(self):
        if getattr(self, "_h", None):
            self._dll.speechPlayer_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodePlayer(object):
    """One batch over several GPUs of a node (speechPlayer_node_*): contiguous shards of near-equal sample count, one
    per device, synthesised side by side; no exchange between devices.  `devices`: HIP device per shard."""

    def __init__(self, sampleRate, devices, mode=0, layout=None):
        self.sampleRate = sampleRate
        self._dll = _native.load()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self._h = self._dll.speechPlayer_node_create(sampleRate, len(dev), dev.ctypes.data)
        if not self._h:
            raise RuntimeError("speechPlayer_node_create failed: %s" % _native.last_error())
        self._check(self._dll.speechPlayer_node_setOption(self._h, b"mode", mode))
        if layout is not None:
            self._check(self._dll.speechPlayer_node_setOption(self._h, b"layout", layout))
        self.nUtterances = 0
        self._lens = None

    def _check(self, rc):
        if rc is None or rc < 0:
            raise RuntimeError("speechPlayer node call failed: %s" % _native.last_error())
        return rc

    def setUtterances(self, frameStart, frames, minSamples, fadeSamples, userIndex=None, isNull=None, noiseSeed=None):
        fs = np.ascontiguousarray(frameStart, dtype=np.int64)
        fr = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 47)
        m = np.ascontiguousarray(minSamples, dtype=np.uint32)
        f = np.ascontiguousarray(fadeSamples, dtype=np.uint32)
        assert fs[-1] == len(fr) == len(m) == len(f)
        ix = None if userIndex is None else np.ascontiguousarray(userIndex, dtype=np.int32)
        nu = None if isNull is None else np.ascontiguousarray(isNull, dtype=np.uint8)
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        p = lambda a: None if a is None else a.ctypes.data
        self._check(self._dll.speechPlayer_node_setUtterances(self._h, len(fs) - 1, p(fs), p(fr), p(m), p(f), p(ix), p(nu), p(sd)))
        self.nUtterances = len(fs) - 1
        per = np.maximum(m.astype(np.int64), np.maximum(f.astype(np.int64), 1) + 1) + 1
        c = np.concatenate([[0], np.cumsum(per)])
        self._lens = c[fs[1:]] - c[fs[:-1]]

    def setIpa(self, texts, speed=1, basePitch=100, inflection=0.5, clauseType=None, noiseSeed=None, voice=None,
               trailing_silence_ms=150.0, textOf=None):
        """The node's batch from IPA text (speechPlayer_node_setIpa): arguments as BatchPlayer.setIpa; the producer's compact form goes to
        every shard, the deal decides which utterances each shard speaks."""
        from .ipa import _text_pointers, _clauses
        ptrs, n, keep = _text_pointers(texts, textOf)
        pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
        sd = None if noiseSeed is None else np.ascontiguousarray(noiseSeed, dtype=np.uint32)
        by_name = voice is None or isinstance(voice, str)
        vo = None if by_name else np.ascontiguousarray(np.broadcast_to(np.asarray(voice, dtype=np.int32), (n,)))
        self._check(self._dll.speechPlayer_node_setIpa(self._h, int(self.sampleRate), n, ptrs.ctypes.data, float(speed), pitch.ctypes.data, float(inflection),
                                                       _clauses(clauseType, n), None if vo is None else vo.ctypes.data,
                                                       (voice.encode("utf8") if (by_name and voice) else None),
                                                       -1.0 if trailing_silence_ms is None else float(trailing_silence_ms), None if sd is None else sd.ctypes.data))
        del keep
        self.nUtterances = n
        self._lens = None

    def utteranceSamples(self, u):
        """Samples utterance u produces (from the shard that holds it)."""
        for d in range(self._dll.speechPlayer_node_devices(self._h)):
            mem = self.shardUtterances(d)
            at = np.flatnonzero(mem == u)
            if len(at):
                return self._dll.speechPlayer_batch_utteranceSamples(self._dll.speechPlayer_node_part(self._h, d), int(at[0]))
        raise RuntimeError("NodePlayer.utteranceSamples: utterance %d out of range" % u)

    @property
    def totalSamples(self):
        return self._dll.speechPlayer_node_totalSamples(self._h)

    def shards(self):
        """[(first utterance, utterances, samples, device)] per shard."""
        import ctypes
        out = []
        for d in range(self._dll.speechPlayer_node_devices(self._h)):
            a, n, s, dev = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
            self._check(self._dll.speechPlayer_node_shardInfo(self._h, d, ctypes.byref(a), ctypes.byref(n), ctypes.byref(s), ctypes.byref(dev)))
            out.append((a.value, n.value, s.value, dev.value))
        return out

    def synthesize(self, wait=True):
        self._check(self._dll.speechPlayer_node_synthesize(self._h))
        if wait:
            self._check(self._dll.speechPlayer_node_wait(self._h))

    def read(self, u):
        if not 0 <= u < self.nUtterances:
            raise RuntimeError("NodePlayer.read: utterance %d out of range" % u)
        n = int(self._lens[u]) if self._lens is not None else int(self.utteranceSamples(u))
        buf = np.zeros(max(n, 1), dtype=np.int16)
        got = self._check(self._dll.speechPlayer_node_read(self._h, u, buf.ctypes.data, n))
        return buf[:got]

    def getLastIndex(self, u):
        return self._dll.speechPlayer_node_getLastIndex(self._h, u)

    def setOption(self, name, value):
        """Batch options go to every shard; "deal": 0 contiguous shards (default), 1 the sorted deal (blocks of 64 length-sorted
        utterances dealt round-robin) -- set it before setUtterances."""
        self._check(self._dll.speechPlayer_node_setOption(self._h, name.encode(), int(value)))

    def shardUtterances(self, d):
        """The utterances of shard d (numbers in the node batch), in the shard's own order."""
        n = self._check(self._dll.speechPlayer_node_shardUtterances(self._h, d, None, 0))
        out = np.zeros(max(n, 1), dtype=np.int64)
        self._check(self._dll.speechPlayer_node_shardUtterances(self._h, d, out.ctypes.data, n))
        return out[:n]

    def digests(self):
        """Per-utterance digests of the PCM in the node batch's utterance order, computed where each shard's PCM lives
        (speechPlayer_batch_digest on the shards: nothing is copied but 8 bytes per utterance)."""
        out = np.zeros(max(self.nUtterances, 1), dtype=np.uint64)
        for d in range(self._dll.speechPlayer_node_devices(self._h)):
            mem = self.shardUtterances(d)
            part = self._dll.speechPlayer_node_part(self._h, d)
            if not part:
                raise RuntimeError("speechPlayer_node_part(%d) failed: %s" % (d, _native.last_error()))
            per = np.zeros(max(len(mem), 1), dtype=np.uint64)
            self._check(self._dll.speechPlayer_batch_digest(part, per.ctypes.data, None))
            out[mem] = per[:len(mem)]
        return out[:self.nUtterances]

    def time(self, launches):
        ms = np.zeros(launches, dtype=np.float32)
        self._check(self._dll.speechPlayer_node_time(self._h, launches, ms.ctypes.data))
        return ms

    def close(self):
        if getattr(self, "_h", None):
            self._dll.speechPlayer_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

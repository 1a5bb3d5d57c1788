"""IPA text -> frame streams: Python face of the native frame producer.

The producer itself is C++ inside the engine library (nvspeechplayer_amd/csrc/frame_producer.cpp, exported through
include/speechPlayer_batch.h): table-driven, built for whole batches, and value for value equal to the reference's
`ipa.generateFramesAndTiming` (reference ipa.py:336-353) and the NVDA driver's voice presets (reference
nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:86-125) on every captured case (tests/test_ipa_producer.py).
This module only marshals arguments; it keeps the reference's call surface (`generateFramesAndTiming`) and adds the
batch packer `frames_for_batch`.  It needs the library but no GPU.
"""
import ctypes

import numpy as np

from . import _native
from .speechPlayer import Frame


def _clause_code(clauseType):
    return 0 if not clauseType else ord(clauseType[0])


def _voice_arg(voice):
    return None if not voice else voice.encode("utf8")


def voices(defined=False):
    """Names of the voice presets (reference __init__.py:86-116), in the driver's sorted order; defined=True: followed by the voices
    defined with defineVoice (a voice's position in that list is its index)."""
    L = _native.load()
    return [L.speechPlayer_voiceName(i).decode("utf8") for i in range(L.speechPlayer_voiceCount() if defined else L.speechPlayer_voicePresetCount())]


def applyVoiceToFrame(frame, voiceName):
    """reference __init__.py:118-125, on a Frame (in place)."""
    if _native.load().speechPlayer_applyVoiceToFrame(ctypes.byref(frame), _voice_arg(voiceName)) != 0:
        raise KeyError(voiceName)


_CLASS_FLAGS = ("_isVowel", "_isVoiced", "_isNasal", "_isStop", "_isLiquid", "_isSemivowel", "_isAfricate", "_copyAdjacent")


def _load_table():
    """The producer's phoneme table as the reference's `data` dict (reference ipa.py:22, data.py): symbol -> {field: value, '_is...': True}."""
    L = _native.load()
    names = [n for n, _ in Frame._fields_]
    table = {}
    for i in range(L.speechPlayer_ipa_phonemeCount()):
        sym = ctypes.create_string_buffer(16)
        vals = (ctypes.c_double * 47)()
        mask = ctypes.c_ulonglong(0)
        cls = ctypes.c_uint(0)
        if L.speechPlayer_ipa_phoneme(i, sym, 16, vals, ctypes.byref(mask), ctypes.byref(cls)) != 0:
            raise RuntimeError("phoneme table entry %d" % i)
        entry = {names[k]: vals[k] for k in range(47) if (mask.value >> k) & 1}
        for b, flag in enumerate(_CLASS_FLAGS):
            if (cls.value >> b) & 1:
                entry[flag] = True
        table[sym.value.decode("utf8")] = entry
    return table


class _LazyTable(dict):
    """`data` is filled from the library on first use (importing this module must not need the library)."""
    _loaded = False

    def _fill(self):
        if not self._loaded:
            self._loaded = True
            self.update(_load_table())

    def __getitem__(self, k):
        self._fill(); return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill(); return dict.__iter__(self)

    def __len__(self):
        self._fill(); return dict.__len__(self)

    def __contains__(self, k):
        self._fill(); return dict.__contains__(self, k)

    def items(self):
        self._fill(); return dict.items(self)

    def keys(self):
        self._fill(); return dict.keys(self)

    def values(self):
        self._fill(); return dict.values(self)

    def get(self, k, d=None):
        self._fill(); return dict.get(self, k, d)


data = _LazyTable()


def iterPhonemes(**kwargs):
    """reference ipa.py:24-27: the symbols whose entry has the given values, e.g. iterPhonemes(_isVoiced=True)."""
    for k, v in data.items():
        if all(v.get(x) == y for x, y in kwargs.items()):
            yield k


def setFrame(frame, phoneme):
    """reference ipa.py:29-32: overwrite the fields the phoneme's entry defines (the class flags become plain attributes)."""
    for k, v in data[phoneme].items():
        setattr(frame, k, v)


def frame_arrays(ipaText, speed=1, basePitch=100, inflection=0.5, clauseType=None, voice=None):
    """One utterance as arrays: (frames[n, 47] f64, isnull[n] u8, duration_ms[n], fade_ms[n])."""
    L = _native.load()
    text = ipaText.encode("utf8")
    args = (text, float(speed), float(basePitch), float(inflection), _clause_code(clauseType), _voice_arg(voice))
    n = L.speechPlayer_ipa_frames(*args, None, None, None, None, 0)
    if n == -2:
        raise KeyError(clauseType)          # as the reference: intonationParamTable[clauseType]
    if n < 0:
        raise KeyError("unknown voice %r" % (voice,))
    frames = np.zeros((n, 47)); nul = np.zeros(n, np.uint8); dur = np.zeros(n); fade = np.zeros(n)
    if n:
        got = L.speechPlayer_ipa_frames(*args, frames.ctypes.data, nul.ctypes.data, dur.ctypes.data, fade.ctypes.data, n)
        assert got == n
    return frames, nul, dur, fade


def frame_vectors(ipaText, speed=1, basePitch=100, inflection=0.5, clauseType=None, voice=None):
    """Yield (47-double vector | None, duration_ms, fade_ms)."""
    frames, nul, dur, fade = frame_arrays(ipaText, speed, basePitch, inflection, clauseType, voice)
    for k in range(len(nul)):
        yield (None if nul[k] else frames[k]), float(dur[k]), float(fade[k])


def generateFramesAndTiming(ipaText, speed=1, basePitch=100, inflection=0.5, clauseType=None):
    """Yield (Frame | None, duration_ms, fade_ms); same signature as reference ipa.py:336."""
    for vec, dur, fade in frame_vectors(ipaText, speed, basePitch, inflection, clauseType):
        yield (None if vec is None else Frame.from_array(vec)), dur, fade


# one frame as the producer hands it to the device (include/speechPlayer_batch.h, speechPlayer_frameRecord_t): 32 bytes
RECORD_DTYPE = np.dtype([("voicePitch", "<f8"), ("endVoicePitch", "<f8"), ("shape", "<u4"), ("min", "<u4"), ("fade", "<u4"), ("index", "<i4")])
RECORD_SILENCE = 0xFFFFFFFF


def _text_pointers(texts, textOf=None):
    """char* per utterance as a uint64 array (-> array, number of utterances, what must stay alive): the strings are encoded once and
    utterances that repeat a sentence (textOf[u]: index into texts) repeat its pointer -- the producer recognises equal pointers
    without reading the text."""
    enc = [ctypes.create_string_buffer(t.encode("utf8")) for t in texts]
    addr = np.array([ctypes.addressof(b) for b in enc] or [0], dtype=np.uint64)
    ptrs = addr[:len(enc)] if textOf is None else addr[np.asarray(textOf, dtype=np.int64)]
    ptrs = np.ascontiguousarray(ptrs if len(ptrs) else np.zeros(1, np.uint64))
    return ptrs, (len(enc) if textOf is None else len(textOf)), enc


def _clauses(clauseType, n):
    if clauseType is None or isinstance(clauseType, str):
        return bytes([_clause_code(clauseType)]) * n + b"\0"
    return bytes(_clause_code(c) for c in clauseType) + b"\0"


def frames_for_batch(texts, sampleRate=22050, speed=1, basePitch=100, inflection=0.5, clauseType=None,
                     trailing_silence_ms=150.0, voice=None, textOf=None):
    """Pack many utterances for BatchPlayer.setUtterances (speechPlayer_ipa_pack).  basePitch and clauseType may be
    sequences (one per utterance).  Each utterance ends with NULL(trailing_silence_ms, 0) as in reference
    test_speakIpa.py:27 (None: no trailing silence).  -> dict(frame_start, frames, min, fade, isnull)."""
    L = _native.load()
    ptrs, n, keep = _text_pointers(texts, textOf)
    pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
    tail = -1.0 if trailing_silence_ms is None else float(trailing_silence_ms)
    start = np.zeros(n + 1, np.int64)
    head = (int(sampleRate), n, ptrs.ctypes.data, float(speed), pitch.ctypes.data, float(inflection), _clauses(clauseType, n), _voice_arg(voice), tail)
    total = L.speechPlayer_ipa_pack(*head, start.ctypes.data, None, None, None, None, 0)
    if total == -2:
        raise KeyError("unknown clause type in %r" % (clauseType,))
    if total < 0:
        raise KeyError("unknown voice %r" % (voice,))
    frames = np.zeros((total, 47)); m = np.zeros(total, np.uint32); f = np.zeros(total, np.uint32); nul = np.zeros(total, np.uint8)
    got = L.speechPlayer_ipa_pack(*head, start.ctypes.data, frames.ctypes.data, m.ctypes.data, f.ctypes.data, nul.ctypes.data, total)
    assert got == total
    del keep
    return dict(frame_start=start, frames=frames, min=m, fade=f, isnull=nul)


class _RecordsView(ctypes.Structure):
    _fields_ = [("nShapes", ctypes.c_longlong), ("shapes", ctypes.c_void_p), ("nLists", ctypes.c_longlong), ("listStart", ctypes.c_void_p),
                ("nRecords", ctypes.c_longlong), ("records", ctypes.c_void_p), ("nUtterances", ctypes.c_longlong), ("listOf", ctypes.c_void_p)]


def records_for_batch(texts, sampleRate=22050, speed=1, basePitch=100, inflection=0.5, clauseType=None,
                      trailing_silence_ms=150.0, voice=None, textOf=None):
    """The same batch in COMPACT form (speechPlayer_ipa_records; what BatchPlayer.setIpa hands the device): -> dict(shapes[nShapes, 47],
    list_start[nLists + 1], records[nRecords] of RECORD_DTYPE, list_of[nUtterances]).  voice: a name, or a sequence of voice indices."""
    L = _native.load()
    ptrs, n, keep = _text_pointers(texts, textOf)
    pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
    tail = -1.0 if trailing_silence_ms is None else float(trailing_silence_ms)
    by_name = voice is None or isinstance(voice, str)
    vo = None if by_name else np.ascontiguousarray(np.broadcast_to(np.asarray(voice, dtype=np.int32), (n,)))
    h = L.speechPlayer_ipa_records(int(sampleRate), n, ptrs.ctypes.data, float(speed), pitch.ctypes.data, float(inflection), _clauses(clauseType, n),
                                   None if vo is None else vo.ctypes.data, _voice_arg(voice) if by_name else None, tail)
    del keep
    if not h:
        raise KeyError("speechPlayer_ipa_records: %s" % _native.last_error())
    try:
        v = _RecordsView()
        assert L.speechPlayer_records_view(h, ctypes.byref(v)) == 0
        grab = lambda ptr, count, dt: np.frombuffer(ctypes.string_at(ptr, count * np.dtype(dt).itemsize), dtype=dt).copy() if count else np.zeros(0, dt)
        return dict(shapes=grab(v.shapes, v.nShapes * 47, np.float64).reshape(-1, 47), list_start=grab(v.listStart, v.nLists + 1, np.int64),
                    records=grab(v.records, v.nRecords, RECORD_DTYPE), list_of=grab(v.listOf, v.nUtterances, np.uint32))
    finally:
        L.speechPlayer_records_free(h)


def expand_records(pk):
    """records_for_batch's result as full frames, utterance by utterance (numpy restatement of klatt_expand_frames, for checks):
    -> dict(frame_start, frames, min, fade, isnull) as frames_for_batch returns it."""
    rec, ls, lo = pk["records"], pk["list_start"], pk["list_of"]
    n_per = (ls[1:] - ls[:-1])[lo]
    fs = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    idx = np.concatenate([np.arange(ls[l], ls[l + 1]) for l in lo]) if len(lo) else np.zeros(0, np.int64)
    r = rec[idx]
    silent = r["shape"] == RECORD_SILENCE
    frames = np.zeros((len(r), 47))
    frames[~silent] = pk["shapes"][r["shape"][~silent]]
    frames[~silent, 0] = r["voicePitch"][~silent]
    frames[~silent, 46] = r["endVoicePitch"][~silent]
    return dict(frame_start=fs, frames=frames, min=r["min"].copy(), fade=r["fade"].copy(), isnull=silent.astype(np.uint8))


def voiceIndex(name):
    """Index of a voice (preset or defined) by name; -1 if there is none."""
    return _native.load().speechPlayer_voiceIndex(name.encode("utf8"))


def defineVoice(name, entries):
    """A voice of the caller's own in the presets' form (reference __init__.py:86-125; speechPlayer_voiceDefine): entries maps a frame
    field (name or index 0..46) to an absolute value, or to a (absolute | None, multiplier | None) pair -- e.g. {"cf1": (None, 0.9),
    "voicePitch": (None, 1.2)}.  -> the voice's index (usable wherever a preset's is)."""
    names = [n for n, _ in Frame._fields_]
    par, ab, mu = [], [], []
    for k, v in entries.items():
        par.append(names.index(k) if isinstance(k, str) else int(k))
        a, m = v if isinstance(v, tuple) else (v, None)
        ab.append(np.nan if a is None else float(a)); mu.append(np.nan if m is None else float(m))
    par = np.asarray(par, np.int32); ab = np.asarray(ab, np.float64); mu = np.asarray(mu, np.float64)
    i = _native.load().speechPlayer_voiceDefine(name.encode("utf8"), len(par), par.ctypes.data, ab.ctypes.data, mu.ctypes.data)
    if i < 0:
        raise ValueError("defineVoice(%r): %s" % (name, _native.last_error()))
    return i


# ---- the optional text front-end (include/speechPlayer_batch.h: speechPlayer_text_*; eSpeak NG loaded at run time) ----
def textAvailable():
    """True when eSpeak NG could be loaded; otherwise False, and _native.last_error() says what is missing."""
    return bool(_native.load().speechPlayer_text_available())


def splitClauses(text):
    """The NVDA driver's split of a text into clauses (reference __init__.py:84, :189-205):
    -> [(clause text, clause type or None, pause after it in ms)]."""
    import ctypes
    L = _native.load()
    b = text.encode("utf8")
    n = L.speechPlayer_text_clauses(b, None, None, None, None, 0)
    if n < 0:
        raise ValueError("bad text")
    beg = (ctypes.c_longlong * max(n, 1))(); end = (ctypes.c_longlong * max(n, 1))()
    typ = (ctypes.c_char * max(n, 1))(); pause = (ctypes.c_double * max(n, 1))()
    L.speechPlayer_text_clauses(b, beg, end, typ, pause, n)
    return [(b[beg[i]:end[i]].decode("utf8"), (typ[i].decode("latin1") if typ[i] != b"\0" else None), pause[i]) for i in range(n)]


def fixups(ipaText):
    """The four replacements and the strip the driver applies to eSpeak's IPA (reference __init__.py:214-218)."""
    import ctypes
    L = _native.load()
    b = ipaText.encode("utf8")
    need = L.speechPlayer_text_fixups(b, None, 0)
    buf = ctypes.create_string_buffer(int(need))
    L.speechPlayer_text_fixups(b, buf, need)
    return buf.value.decode("utf8")


def textToIpa(text, espeakVoice="en"):
    """One clause of text -> IPA through eSpeak NG (mode word 0x36100 + 0x82, reference __init__.py:210) + fixups.
    RuntimeError when the library is not installed."""
    import ctypes
    L = _native.load()
    b = text.encode("utf8")
    v = espeakVoice.encode("utf8") if espeakVoice else None
    need = L.speechPlayer_text_toIpa(b, v, None, 0)
    if need < 0:
        raise RuntimeError("speechPlayer_text_toIpa failed: %s" % _native.last_error())
    buf = ctypes.create_string_buffer(int(need))
    L.speechPlayer_text_toIpa(b, v, buf, need)
    return buf.value.decode("utf8")

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


def voices():
    """Names of the voice presets (reference __init__.py:86-116), in the driver's sorted order."""
    L = _native.load()
    return [L.speechPlayer_voiceName(i).decode("utf8") for i in range(L.speechPlayer_voiceCount())]


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


def frames_for_batch(texts, sampleRate=22050, speed=1, basePitch=100, inflection=0.5, clauseType=None,
                     trailing_silence_ms=150.0, voice=None):
    """Pack many utterances for BatchPlayer.setUtterances (speechPlayer_ipa_pack).  basePitch and clauseType may be
    sequences (one per text).  Each utterance ends with NULL(trailing_silence_ms, 0) as in reference
    test_speakIpa.py:27 (None: no trailing silence).  -> dict(frame_start, frames, min, fade, isnull)."""
    L = _native.load()
    n = len(texts)
    enc = [t.encode("utf8") for t in texts]
    ptrs = (ctypes.c_char_p * max(n, 1))(*enc)
    pitch = np.ascontiguousarray(np.broadcast_to(np.asarray(basePitch, dtype=np.float64), (n,)))
    if clauseType is None or isinstance(clauseType, str):
        clauses = bytes([_clause_code(clauseType)]) * n
    else:
        clauses = bytes(_clause_code(c) for c in clauseType)
    tail = -1.0 if trailing_silence_ms is None else float(trailing_silence_ms)
    start = np.zeros(n + 1, np.int64)
    head = (int(sampleRate), n, ptrs, float(speed), pitch.ctypes.data, float(inflection), clauses + b"\0", _voice_arg(voice), tail)
    total = L.speechPlayer_ipa_pack(*head, start.ctypes.data, None, None, None, None, 0)
    if total == -2:
        raise KeyError("unknown clause type in %r" % (clauseType,))
    if total < 0:
        raise KeyError("unknown voice %r" % (voice,))
    frames = np.zeros((total, 47)); m = np.zeros(total, np.uint32); f = np.zeros(total, np.uint32); nul = np.zeros(total, np.uint8)
    got = L.speechPlayer_ipa_pack(*head, start.ctypes.data, frames.ctypes.data, m.ctypes.data, f.ctypes.data, nul.ctypes.data, total)
    assert got == total
    return dict(frame_start=start, frames=frames, min=m, fade=f, isnull=nul)


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

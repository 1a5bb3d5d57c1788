"""IPA text -> frame streams: the producer that feeds the synthesis hot path.

Host-side counterpart of the reference's `ipa.generateFramesAndTiming` (reference ipa.py:336-353),
written from its behaviour: the same phoneme segmentation (ipa.py:39-119), /h/ colouring
(ipa.py:121-133), durations (ipa.py:135-184) and intonation contours (ipa.py:186-334), so that for the
same text and settings it yields the same (frame | None, duration_ms, fade_ms) triples -- checked
value for value against streams captured from the reference (tests/test_ipa_producer.py).
The phoneme parameter table and the intonation constants are data files generated from the
reference's tables by tests/golden/make_golden.py (nvspeechplayer_amd/data/*.json).

`frames_for_batch` turns many texts into the packed arrays `BatchPlayer.setUtterances` takes.
"""
import itertools
import json
import os

import numpy as np

from .speechPlayer import FRAME_FIELDS, Frame

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_FIELD_INDEX = {name: i for i, name in enumerate(FRAME_FIELDS)}

PRIMARY, SECONDARY = 1, 2
STRESS_PRIMARY_MARK, STRESS_SECONDARY_MARK, LENGTH_MARK, TIE_BAR = "ˈ", "ˌ", "ː", "͡"

with open(os.path.join(_DATA, "phonemes.json"), encoding="utf8") as _f:
    PHONEMES = json.load(_f)
with open(os.path.join(_DATA, "intonation.json")) as _f:
    INTONATION = json.load(_f)


class Phone(object):
    """One segment of an utterance: parameter values it sets, class flags, prosodic marks."""
    __slots__ = ("fields", "vowel", "voiced", "stop", "affricate", "liquid", "semivowel", "nasal", "copy_adjacent",
                 "silence", "pre_stop_gap", "post_stop_aspiration", "stress", "tied_to", "tied_from", "lengthened",
                 "word_start", "syllable_start", "duration", "fade")

    def __init__(self, entry=None):
        flags = entry["flags"] if entry else {}
        self.fields = dict(entry["fields"]) if entry else {}
        self.vowel = bool(flags.get("_isVowel")); self.voiced = bool(flags.get("_isVoiced"))
        self.stop = bool(flags.get("_isStop")); self.affricate = bool(flags.get("_isAfricate"))
        self.liquid = bool(flags.get("_isLiquid")); self.semivowel = bool(flags.get("_isSemivowel"))
        self.nasal = bool(flags.get("_isNasal")); self.copy_adjacent = bool(flags.get("_copyAdjacent"))
        self.silence = self.pre_stop_gap = self.post_stop_aspiration = False
        self.stress = 0
        self.tied_to = self.tied_from = self.lengthened = False
        self.word_start = self.syllable_start = False
        self.duration = self.fade = 0.0


def _scan(text):
    """Yield (char, Phone | None, stress) for every position that is not a stress mark.  A tie bar joins two
    symbols into one table entry when the table has it, a length mark prefers the lengthened entry
    (reference ipa.py:39-81)."""
    pending_stress = 0
    pos = 0
    n = len(text)
    while pos < n:
        ch = text[pos]
        if ch == STRESS_PRIMARY_MARK:
            pending_stress = PRIMARY; pos += 1; continue
        if ch == STRESS_SECONDARY_MARK:
            pending_stress = SECONDARY; pos += 1; continue
        nxt = text[pos + 1:pos + 2]
        lengthened = nxt == LENGTH_MARK
        tied_to = nxt == TIE_BAR
        tied_from = pos > 0 and text[pos - 1] == TIE_BAR
        entry = None
        step = 1
        if tied_to:
            entry = PHONEMES.get(text[pos:pos + 3])
            step += 2 if entry else 1
        elif lengthened:
            entry = PHONEMES.get(text[pos:pos + 2])
            step += 1
        if not entry:
            entry = PHONEMES.get(ch)
        pos += step
        if not entry:
            yield ch, None, 0
            continue
        ph = Phone(entry)
        stress, pending_stress = pending_stress, 0
        if tied_from:
            ph.tied_from = True
        elif tied_to:
            ph.tied_to = True
        ph.lengthened = lengthened
        yield ch, ph, stress


def segment(text):
    """Phones of an utterance with word/syllable starts, stress, inserted post-stop aspiration and pre-stop
    gaps (reference ipa.py:83-119)."""
    phones = []
    new_word = True
    last = None
    syllable_head = None
    for ch, ph, stress in _scan(text):
        if ch == " ":
            new_word = True
            continue
        if ph is None:
            continue
        if last is not None and not last.vowel and ph.vowel:
            last.syllable_start = True
            syllable_head = last
        elif stress == PRIMARY and last is not None and last.vowel:
            ph.syllable_start = True
            syllable_head = ph
        if last is not None and last.stop and not last.voiced and ph.voiced and not ph.stop and not ph.affricate:
            puff = Phone(PHONEMES["h"])
            puff.post_stop_aspiration = True
            phones.append(puff)
            last = puff
        if new_word:
            new_word = False
            ph.word_start = True
            ph.syllable_start = True
            syllable_head = ph
        if stress:
            syllable_head.stress = stress
        elif ph.stop or ph.affricate:
            gap = Phone()
            gap.silence = True
            gap.pre_stop_gap = True
            phones.append(gap)
        phones.append(ph)
        last = ph
    return phones


def colour_h(phones):
    """/h/-like phones borrow the formants of the next phone (or the previous one) (reference ipa.py:121-133)."""
    for i, ph in enumerate(phones):
        if not ph.copy_adjacent:
            continue
        nxt = phones[i + 1] if i + 1 < len(phones) else None
        prev = phones[i - 1] if i > 0 else None
        src = nxt if (nxt is not None and not nxt.silence) else prev
        if src is not None:
            for k, v in src.fields.items():
                if k not in ph.fields:
                    ph.fields[k] = v


def assign_times(phones, base_speed):
    """Duration and fade of every phone in ms (reference ipa.py:135-184)."""
    last = None
    syllable_stress = 0
    speed = base_speed
    for i, ph in enumerate(phones):
        nxt = phones[i + 1] if i + 1 < len(phones) else None
        if ph.syllable_start:
            syllable_stress = ph.stress
            if syllable_stress:
                speed = base_speed / 1.4 if syllable_stress == PRIMARY else base_speed / 1.1
            else:
                speed = base_speed
        dur = 60.0 / speed
        fade = 10.0 / speed
        if ph.pre_stop_gap:
            dur = 41.0 / speed
        elif ph.post_stop_aspiration:
            dur = 20.0 / speed
        elif ph.stop:
            dur = min(6.0 / speed, 6.0)
            fade = 0.001
        elif ph.affricate:
            dur = 24.0 / speed
            fade = 0.001
        elif not ph.voiced:
            dur = 45.0 / speed
        elif ph.vowel:
            if last is not None and (last.liquid or last.semivowel):
                fade = 25.0 / speed
            if ph.tied_to:
                dur = 40.0 / speed
            elif ph.tied_from:
                dur = 20.0 / speed
                fade = 20.0 / speed
            elif (not syllable_stress and not ph.syllable_start and nxt is not None and not nxt.word_start
                  and (nxt.liquid or nxt.nasal)):
                dur = 30.0 / speed if nxt.liquid else 40.0 / speed
        else:
            dur = 30.0 / speed
            if ph.liquid or ph.semivowel:
                fade = 20.0 / speed
        if ph.lengthened:
            dur *= 1.05
        ph.duration = dur
        ph.fade = fade
        last = ph


def _pitch_path(phones, start, end, base_pitch, inflection, start_percent, end_percent):
    """Linear pitch glide over the voiced time of phones[start:end] (reference ipa.py:186-205)."""
    p0 = base_pitch * (2 ** (((start_percent - 50) / 50.0) * inflection))
    p1 = base_pitch * (2 ** (((end_percent - 50) / 50.0) * inflection))
    voiced_ms = 0
    for ph in phones[start:end]:
        if ph.voiced:
            voiced_ms += ph.duration
    done = 0
    delta = p1 - p0
    cur = p0
    for ph in phones[start:end]:
        ph.fields["voicePitch"] = cur
        if ph.voiced:
            done += ph.duration
            cur = p0 + (delta * (done / float(voiced_ms)))
        ph.fields["endVoicePitch"] = cur


def assign_pitches(phones, base_pitch, inflection, clause_type):
    """Pre-head, head (stepping down over stressed syllables), nucleus and tail contours
    (reference ipa.py:278-334)."""
    t = INTONATION[clause_type or "."]
    n = len(phones)
    prehead_end = n
    for i, ph in enumerate(phones):
        if ph.syllable_start and ph.stress == PRIMARY:
            prehead_end = i
            break
    if prehead_end > 0:
        _pitch_path(phones, 0, prehead_end, base_pitch, inflection, t["preHeadStart"], t["preHeadEnd"])
    nucleus_start = nucleus_end = tail_start = tail_end = n
    for i in range(nucleus_end - 1, prehead_end - 1, -1):
        ph = phones[i]
        if ph.syllable_start:
            if ph.stress == PRIMARY:
                nucleus_start = i
                break
            nucleus_end = tail_start = i
    has_tail = tail_end - tail_start > 0
    if has_tail:
        _pitch_path(phones, tail_start, tail_end, base_pitch, inflection, t["tailStart"], t["tailEnd"])
    if nucleus_end - nucleus_start > 0:
        if has_tail:
            _pitch_path(phones, nucleus_start, nucleus_end, base_pitch, inflection, t["nucleusStart"], t["nucleusEnd"])
        else:
            _pitch_path(phones, nucleus_start, nucleus_end, base_pitch, inflection, t["nucleus0Start"], t["nucleus0End"])
    if prehead_end < nucleus_start:
        head_hi, head_lo = t["headStart"], t["headEnd"]
        steps = t["headSteps"]
        step_gen = itertools.chain(steps, itertools.cycle(steps[t["headExtendFrom"]:]))
        stressed_from = None
        unstressed_from = None
        stress_end_pitch = None
        for i in range(prehead_end, nucleus_start + 1):
            ph = phones[i]
            primary = ph.stress == PRIMARY
            if not ph.syllable_start:
                continue
            if stressed_from is not None:
                start_p = head_lo + (((head_hi - head_lo) / 100.0) * next(step_gen))
                stress_end_pitch = start_p + t["headStressEndDelta"]
                _pitch_path(phones, stressed_from, i, base_pitch, inflection, start_p, stress_end_pitch)
                stressed_from = None
            if primary:
                if unstressed_from is not None:
                    _pitch_path(phones, unstressed_from, i, base_pitch, inflection,
                                stress_end_pitch + t["headUnstressedRunStartDelta"],
                                stress_end_pitch + t["headUnstressedRunEndDelta"])
                    unstressed_from = None
                stressed_from = i
            elif unstressed_from is None:
                unstressed_from = i


def generateFramesAndTiming(ipaText, speed=1, basePitch=100, inflection=0.5, clauseType=None):
    """Yield (Frame | None, duration_ms, fade_ms); same signature as reference ipa.py:336."""
    for vec, dur, fade in frame_vectors(ipaText, speed, basePitch, inflection, clauseType):
        yield (None if vec is None else Frame.from_array(vec)), dur, fade


def frame_vectors(ipaText, speed=1, basePitch=100, inflection=0.5, clauseType=None):
    """Like generateFramesAndTiming but with 47-double numpy vectors instead of ctypes Frames."""
    phones = segment(ipaText)
    if not phones:
        return
    colour_h(phones)
    assign_times(phones, speed)
    assign_pitches(phones, basePitch, inflection, clauseType)
    for ph in phones:
        if ph.silence:
            yield None, ph.duration, ph.fade
            continue
        vec = np.zeros(47)
        vec[_FIELD_INDEX["preFormantGain"]] = 1.0      # reference ipa.py:350-351
        vec[_FIELD_INDEX["outputGain"]] = 2.0
        for k, v in ph.fields.items():
            vec[_FIELD_INDEX[k]] = v
        yield vec, ph.duration, ph.fade


def frames_for_batch(texts, sampleRate=22050, speed=1, basePitch=100, inflection=0.5, clauseType=None,
                     trailing_silence_ms=150.0):
    """Pack many utterances for BatchPlayer.setUtterances.  basePitch may be a sequence (one per text).
    Each utterance ends with NULL(trailing_silence_ms, 0) as in reference test_speakIpa.py:27.
    -> dict(frame_start, frames, min, fade, isnull)."""
    conv = lambda ms: int(ms * (sampleRate / 1000.0))      # reference speechPlayer.py:53
    frames, mins, fades, nul, start = [], [], [], [], [0]
    for i, text in enumerate(texts):
        pitch = basePitch[i] if hasattr(basePitch, "__len__") else basePitch
        n = 0
        for vec, dur, fade in frame_vectors(text, speed, pitch, inflection, clauseType):
            frames.append(np.zeros(47) if vec is None else vec)
            nul.append(vec is None); mins.append(conv(dur)); fades.append(conv(fade)); n += 1
        if trailing_silence_ms is not None:
            frames.append(np.zeros(47)); nul.append(True); mins.append(conv(trailing_silence_ms)); fades.append(0); n += 1
        start.append(start[-1] + n)
    return dict(frame_start=np.array(start, np.int64), frames=np.array(frames).reshape(-1, 47),
                min=np.array(mins, np.uint32), fade=np.array(fades, np.uint32), isnull=np.array(nul, np.uint8))

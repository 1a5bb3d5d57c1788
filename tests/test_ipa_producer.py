"""The IPA -> frames producer (nvspeechplayer_amd/ipa.py) against streams captured from the reference's
own ipa.generateFramesAndTiming (tests/golden/ref_frames.npz): every frame value, NULL flag, duration and
fade must be identical, for the eight sampleIpa.txt lines x five clause types x two speeds, pitch /
inflection variants, and ten extra lines with tie bars, length and stress marks and unknown symbols
(SURVEY.md section 8(f) rank 2)."""
import numpy as np
import pytest

from tests import scenarios

CLAUSES = {0: ".", 1: ",", 2: "?", 3: "!", 4: None}


@pytest.fixture(scope="module")
def ref():
    return scenarios.Ref()


def test_every_captured_case_is_reproduced_exactly(ref):
    from nvspeechplayer_amd import ipa
    z = np.load(scenarios.GOLDEN + "/ref_frames.npz")
    lines = [b.decode("utf8") for b in z["ipa_lines"]]
    n_frames = 0
    for i, meta in enumerate(ref.ipa_meta):
        li, speed, clause, pitch, infl = int(meta[0]), float(meta[1]), CLAUSES[int(meta[2])], float(meta[3]), float(meta[4])
        got = list(ipa.frame_vectors(lines[li], speed=speed, basePitch=pitch, inflection=infl, clauseType=clause))
        a, b = ref.ipa_start[i], ref.ipa_start[i + 1]
        assert len(got) == b - a, (i, len(got), b - a)
        for k, (vec, dur, fade) in enumerate(got):
            assert (vec is None) == bool(ref.ipa_isnull[a + k]), (i, k)
            assert dur == ref.ipa_dur_ms[a + k] and fade == ref.ipa_fade_ms[a + k], (i, k, dur, fade)
            if vec is not None:
                assert np.array_equal(vec, ref.ipa_frames[a + k]), (i, k, np.flatnonzero(vec != ref.ipa_frames[a + k]))
            n_frames += 1
    assert n_frames == len(ref.ipa_frames) and len(ref.ipa_meta) == 126


def test_frame_objects_and_batch_packing(ref):
    from nvspeechplayer_amd import ipa, Frame
    out = list(ipa.generateFramesAndTiming("hælou", speed=1.0, basePitch=100, inflection=0.5, clauseType="."))
    assert all(f is None or isinstance(f, Frame) for f, _, _ in out)
    case = ref.ipa_case(ref.find_ipa(0))
    pk = ipa.frames_for_batch(["hælou", "", "hæv ju enj wʊl"], basePitch=[100.0, 100.0, 100.0], clauseType=".")
    assert list(pk["frame_start"][:2]) == [0, len(case)]
    assert pk["frame_start"][2] - pk["frame_start"][1] == 1          # empty text: only the trailing silence
    assert [int(x) for x in pk["min"][:len(case)]] == [c[1] for c in case]
    assert [int(x) for x in pk["fade"][:len(case)]] == [c[2] for c in case]
    assert pk["isnull"][len(case) - 1] == 1


def test_unknown_symbols_stress_and_ties():
    from nvspeechplayer_amd import ipa
    # unknown characters are skipped, stress marks move to the syllable head, a tie bar forms an affricate
    ph = ipa.segment("ˈt͡ʃɑ #ˌpɑː")
    names = [(p.stop, p.affricate, p.vowel, p.silence, p.stress, p.lengthened) for p in ph]
    assert any(p.affricate for p in ph) and any(p.lengthened for p in ph)
    assert ph[0].stress == 1 or ph[1].stress == 1
    assert len(list(ipa.frame_vectors(""))) == 0

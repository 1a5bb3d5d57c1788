"""The native IPA -> frames producer (nvspeechplayer_amd/csrc/frame_producer.cpp, called through the C-ABI of
include/speechPlayer_batch.h) against streams captured from the reference's own ipa.generateFramesAndTiming and
applyVoiceToFrame (tests/golden/ref_frames.npz, made by tests/golden/make_golden.py): every frame value, NULL flag,
duration and fade must be identical, for the eight sampleIpa.txt lines x five clause types x two speeds, pitch /
inflection variants, ten extra lines with tie bars, length and stress marks and unknown symbols, and the four voice
presets.  Host code: runs without a GPU."""
import ctypes
import os

import numpy as np
import pytest

from tests import scenarios

CLAUSES = {0: ".", 1: ",", 2: "?", 3: "!", 4: None}


@pytest.fixture(scope="module")
def ref():
    return scenarios.Ref()


@pytest.fixture(scope="module")
def golden():
    return np.load(scenarios.GOLDEN + "/ref_frames.npz")


def test_every_captured_case_is_reproduced_exactly(ref, golden):
    from nvspeechplayer_amd import ipa
    lines = [b.decode("utf8") for b in golden["ipa_lines"]]
    n_frames = 0
    for i, meta in enumerate(ref.ipa_meta):
        li, speed, clause, pitch, infl = int(meta[0]), float(meta[1]), CLAUSES[int(meta[2])], float(meta[3]), float(meta[4])
        frames, nul, dur, fade = ipa.frame_arrays(lines[li], speed=speed, basePitch=pitch, inflection=infl, clauseType=clause)
        a, b = ref.ipa_start[i], ref.ipa_start[i + 1]
        assert len(nul) == b - a, (i, len(nul), b - a)
        assert np.array_equal(nul, ref.ipa_isnull[a:b]), i
        assert np.array_equal(dur, ref.ipa_dur_ms[a:b]) and np.array_equal(fade, ref.ipa_fade_ms[a:b]), i
        real = nul == 0
        assert np.array_equal(frames[real], ref.ipa_frames[a:b][real]), (i, np.argwhere(frames[real] != ref.ipa_frames[a:b][real])[:5])
        assert not frames[~real].any()
        n_frames += b - a
    assert n_frames == len(ref.ipa_frames) and len(ref.ipa_meta) == 126


def test_voice_presets_match_the_captured_cases(golden):
    """The four presets of the NVDA driver applied to every frame of the eight sampleIpa lines: through the producer's
    `voice` argument and through speechPlayer_applyVoiceToFrame on an unvoiced frame."""
    from nvspeechplayer_amd import Frame, ipa
    names = [b.decode("utf8") for b in golden["voice_names"]]
    assert ipa.voices() == names and len(names) == 4
    lines = [b.decode("utf8") for b in golden["ipa_lines"]]
    meta, want = golden["voice_case_meta"], golden["voice_case_frames"]
    k = 0
    for vi, vname in enumerate(names):
        for li in range(8):
            frames, nul, _, _ = ipa.frame_arrays(lines[li], speed=1.0, basePitch=100.0, inflection=0.5, clauseType=".", voice=vname)
            plain, _, _, _ = ipa.frame_arrays(lines[li], speed=1.0, basePitch=100.0, inflection=0.5, clauseType=".")
            for j in np.flatnonzero(nul == 0):
                assert tuple(meta[k]) == (vi, li)
                assert np.array_equal(frames[j], want[k]), (vname, li, j, np.flatnonzero(frames[j] != want[k]))
                f = Frame.from_array(plain[j])
                ipa.applyVoiceToFrame(f, vname)
                assert np.array_equal(f.as_array(), want[k])
                k += 1
    assert k == len(want)
    assert any(not np.array_equal(a, b) for a, b in zip(want[:50], want[200:250]))       # the presets differ
    with pytest.raises(KeyError):
        ipa.frame_arrays("hælou", voice="Nobody")
    f = Frame.from_array(np.arange(47.0))
    ipa.applyVoiceToFrame(f, "Caleb")                   # the reference's key carries a trailing blank; both spellings work
    assert f.voiceAmplitude == 0.0 and f.aspirationAmplitude == 1.0


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
    # per-text clause types and pitches, no trailing silence, another sample rate; instanced streams are identical copies
    texts = ["hæv ju enj wʊl", "ðɪs ɪz veɹj fɑn", "hæv ju enj wʊl", "hæv ju enj wʊl"]
    pk = ipa.frames_for_batch(texts, sampleRate=16000, speed=0.8, basePitch=[90.0, 120.0, 90.0, 91.0], clauseType=["?", None, "?", "?"],
                              trailing_silence_ms=None)
    fs = pk["frame_start"]
    one = [ipa.frame_arrays(t, speed=0.8, basePitch=p, clauseType=c) for t, p, c in zip(texts, (90.0, 120.0, 90.0, 91.0), ("?", None, "?", "?"))]
    for u in range(4):
        fr, nul, dur, fade = one[u]
        assert fs[u + 1] - fs[u] == len(nul)
        assert np.array_equal(pk["frames"][fs[u]:fs[u + 1]], fr) and np.array_equal(pk["isnull"][fs[u]:fs[u + 1]], nul)
        assert [int(x) for x in pk["min"][fs[u]:fs[u + 1]]] == [int(d * (16000 / 1000.0)) for d in dur]      # reference speechPlayer.py:53
        assert [int(x) for x in pk["fade"][fs[u]:fs[u + 1]]] == [int(d * (16000 / 1000.0)) for d in fade]
    assert np.array_equal(pk["frames"][fs[0]:fs[1]], pk["frames"][fs[2]:fs[3]])
    assert not np.array_equal(pk["frames"][fs[0]:fs[1]], pk["frames"][fs[3]:fs[4]])       # 1 Hz apart


def test_unknown_symbols_stress_and_ties():
    from nvspeechplayer_amd import ipa
    # unknown characters are skipped, a tie bar forms an affricate (one frame, 24 ms, preceded by a gap unless it carries the
    # stress mark), a length mark lengthens by 5 %, an empty text yields nothing
    fr, nul, dur, fade = ipa.frame_arrays("t͡ʃɑ #pɑː")
    assert list(nul) == [1, 0, 0, 1, 0, 0, 0]                       # gap t͡ʃ ɑ | gap p (aspiration) ɑː
    assert dur[1] == 24.0 and fade[1] == 0.001 and dur[0] == 41.0
    assert dur[5] == 20.0 and dur[6] == 60.0 * 1.05
    fr2, nul2, dur2, _ = ipa.frame_arrays("ˈt͡ʃɑ")
    assert list(nul2) == [0, 0] and dur2[0] == 24.0 / (1 / 1.4)     # stressed: no gap, slower syllable
    assert len(ipa.frame_arrays("")[1]) == 0 and len(ipa.frame_arrays("#7 ")[1]) == 0
    # invalid UTF-8 is an unknown symbol, not a crash
    L = __import__("nvspeechplayer_amd")._native.load()
    assert L.speechPlayer_ipa_frames(b"h\xff\xfe\xe6lou", 1.0, 100.0, 0.5, 0, None, None, None, None, None, 0) >= 3


def test_producer_under_sanitizers(tmp_path):
    """The producer takes text from outside: 40 000 random symbol sequences (invalid UTF-8 included) and a packed batch with
    duplicates and empty texts run clean under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build of the same source)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_producer")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "native", "fuzz_producer.cpp"),
                           os.path.join(root, "nvspeechplayer_amd", "csrc", "frame_producer.cpp"), "-o", exe])
    out = subprocess.check_output([exe], stderr=subprocess.STDOUT).decode()
    assert out.startswith("ok ") and "runtime error" not in out and "AddressSanitizer" not in out, out


def test_phoneme_table_surface_of_the_reference():
    """`data`, `setFrame`, `iterPhonemes` (reference ipa.py:22-32), served from the producer's own table: every entry equals the
    captured reference table value for value and flag for flag; an unknown clause type raises KeyError as the reference does."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import ipa
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_frames.npz"))
    names = [b.decode("utf8") for b in z["phoneme_names"]]
    assert sorted(ipa.data) == sorted(names) and len(ipa.data) == 49
    fields = [n for n, _ in eng.Frame._fields_]
    for i, n in enumerate(names):
        fr = eng.Frame()
        ipa.setFrame(fr, n)
        assert np.array_equal(np.array([getattr(fr, k) for k in fields]), z["phoneme_frames"][i]), n
        assert sorted(k for k in ipa.data[n] if not k.startswith("_")) == sorted(f for f, m in zip(fields, z["phoneme_mask"][i]) if m), n
        for fl in ("_isVowel", "_isVoiced", "_isNasal", "_isStop", "_isLiquid", "_isSemivowel", "_isAfricate", "_copyAdjacent"):
            assert bool(ipa.data[n].get(fl)) == bool(z["phoneme" + fl][i]), (n, fl)
    assert sorted(ipa.iterPhonemes(_isVoiced=True)) == sorted(names[i] for i in z["voiced_order"])
    with pytest.raises(KeyError):
        list(ipa.generateFramesAndTiming("hælou", clauseType=";"))
    with pytest.raises(KeyError):
        ipa.frames_for_batch(["hælou", "wɜːld"], clauseType=[".", ";"])


def test_compact_form_expands_to_the_packed_frames():
    """The producer's compact form (speechPlayer_ipa_records: a (voice, composite) shape table, 32-byte records, lists that utterances
    share) expanded with numpy equals what speechPlayer_ipa_pack hands out as full frames, array for array -- for BASELINE configs[2] and
    configs[4] (workloads.cfg2_spec / cfg4_spec against workloads.make, which multiplies the frames by each variant's factors itself),
    for the captured lines with tie bars, length marks and unknown symbols, with and without the closing silence, a voice per text."""
    from nvspeechplayer_amd import ipa, workloads
    keys = ("frame_start", "frames", "min", "fade", "isnull")

    def both(texts, **kw):
        full = ipa.frames_for_batch(texts, **kw)
        pk = ipa.records_for_batch(texts, **kw)
        got = ipa.expand_records(pk)
        for k in keys:
            assert np.array_equal(full[k], got[k]), k
        return pk
    b = workloads.make("cfg2", 1536, first=700)
    sp = workloads.cfg2_spec(1536, first=700)
    pk = both(sp["texts"], textOf=sp["textOf"], basePitch=sp["basePitch"], clauseType=".", trailing_silence_ms=150.0)
    e = ipa.expand_records(pk)
    for k in keys:
        assert np.array_equal(b[k], e[k]), k
    assert len(pk["list_start"]) - 1 == 512 and pk["list_of"].max() == 511 and len(pk["records"]) == pk["list_start"][-1] < len(b["min"])
    assert pk["shapes"].shape[0] == 57 and pk["records"].dtype.itemsize == 32
    b4 = workloads.make("cfg4", 3072, first=1024)
    sp4 = workloads.cfg4_spec(3072, first=1024, per=1024)
    pk4 = ipa.records_for_batch(sp4["texts"], textOf=sp4["textOf"], basePitch=sp4["basePitch"], clauseType=".", voice=sp4["voice"])
    e4 = ipa.expand_records(pk4)
    for k in keys:
        assert np.array_equal(b4[k], e4[k]), k
    assert np.array_equal(b4["seeds"], sp4["noiseSeed"]) and len(np.unique(sp4["voice"])) == 3
    golden_lines = [b.decode("utf8") for b in np.load(scenarios.GOLDEN + "/ref_frames.npz")["ipa_lines"]]
    for tail in (None, 0.0, 150.0):
        for voice in (None, "Adam", "Caleb"):
            both(golden_lines + ["", "x#"], clauseType=[".", "?", "!", ",", None] * 4, basePitch=np.linspace(60, 240, 20), speed=0.6,
                 trailing_silence_ms=tail, voice=voice)


def test_defined_voices_behave_like_presets():
    """speechPlayer_voiceDefine: a caller's voice in the presets' form (absolute value first, then the multiplier; reference
    __init__.py:118-125) applied by the producer and by speechPlayer_applyVoiceToFrame; a definition can be replaced, a preset's name
    cannot be taken, an index out of range is refused."""
    from nvspeechplayer_amd import Frame, ipa
    n0 = len(ipa.voices(defined=True))
    v = ipa.defineVoice("test-voice", {"cf1": (None, 0.9), "cb1": 80.0, "voicePitch": (150.0, 1.1), 46: (None, 2.0)})
    assert v == ipa.voiceIndex("test-voice") >= 4 and ipa.voices() == ipa.voices(defined=True)[:4]
    plain, nul, _, _ = ipa.frame_arrays("hælou", basePitch=100.0, clauseType=".")
    mine, nul2, _, _ = ipa.frame_arrays("hælou", basePitch=100.0, clauseType=".", voice="test-voice")
    real = nul == 0
    assert np.array_equal(nul, nul2)
    assert np.array_equal(mine[real, 7], plain[real, 7] * 0.9) and (mine[real, 15] == 80.0).all()
    assert (mine[real, 0] == 150.0 * 1.1).all() and np.array_equal(mine[real, 46], plain[real, 46] * 2.0)
    others = [k for k in range(47) if k not in (0, 7, 15, 46)]
    assert np.array_equal(mine[:, others], plain[:, others])
    fr = Frame.from_array(plain[real][0])
    ipa.applyVoiceToFrame(fr, "test-voice")
    assert np.array_equal(np.array([getattr(fr, n) for n, _ in Frame._fields_]), mine[real][0])
    assert ipa.defineVoice("test-voice", {"cf1": (None, 0.5)}) == v                  # replaced in place
    again, _, _, _ = ipa.frame_arrays("hælou", basePitch=100.0, clauseType=".", voice="test-voice")
    assert np.array_equal(again[real, 7], plain[real, 7] * 0.5) and np.array_equal(again[real, 15], plain[real, 15])
    with pytest.raises(ValueError):
        ipa.defineVoice("Adam", {"cf1": 1.0})
    with pytest.raises(KeyError):
        ipa.records_for_batch(["hælou"], voice=[len(ipa.voices(defined=True))])
    assert len(ipa.voices(defined=True)) == max(n0, v + 1)

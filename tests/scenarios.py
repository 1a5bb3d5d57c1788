"""Shared parity scenarios: frame streams + the call sequence that plays them.

A scenario is a list of operations against the speechPlayer C-ABI, in SAMPLES:
    ("q", frame[47] | None, minSamples, fadeSamples, userIndex, purge)
    ("s", n)         one synthesize(n) call
    ("drain",)       synthesize(8192) until a short count
It can be played on the oracle (play_oracle) or on the HIP engine (tests do that
through the product C-ABI) and the PCM / index marks compared call by call.

Inputs come from tests/golden/ref_frames.npz, i.e. from the reference's own frame
producer (see tests/golden/make_golden.py); the call recipes follow the
reference's demo scripts, cited per scenario.
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SR = 22050

# parameter indices (frame.h:24-42)
VOICEPITCH, VIBOFFSET, VIBSPEED, TURB, OPENQ, VOICEAMP, ASPAMP = range(7)
CANP, FRICAMP, BYPASS, PREGAIN, OUTGAIN, ENDPITCH = 23, 24, 43, 44, 45, 46


def ms(x, sr=SR):
    """speechPlayer.py:53 ms -> samples (truncation)."""
    return int(x * (sr / 1000.0))


class Ref:
    def __init__(self, path=None):
        z = np.load(path or os.path.join(GOLDEN, "ref_frames.npz"))
        self.names = [b.decode("utf8") for b in z["phoneme_names"]]
        self.frames = z["phoneme_frames"]
        self.mask = z["phoneme_mask"].astype(bool)
        self.field_names = [b.decode() for b in z["field_names"]]
        self.is_vowel = z["phoneme_isVowel"].astype(bool)
        self.is_voiced = z["phoneme_isVoiced"].astype(bool)
        self.voiced_order = z["voiced_order"]
        self.ipa_meta = z["ipa_case_meta"]
        self.ipa_frames = z["ipa_frames"]
        self.ipa_isnull = z["ipa_isnull"]
        self.ipa_dur_ms = z["ipa_dur_ms"]
        self.ipa_fade_ms = z["ipa_fade_ms"]
        self.ipa_start = z["ipa_start"]

    def phoneme(self, name):
        return self.frames[self.names.index(name)].copy()

    def set_frame(self, frame, name):
        """ipa.setFrame (ipa.py:29-32): overwrite the fields the phoneme entry defines."""
        i = self.names.index(name)
        frame[self.mask[i]] = self.frames[i][self.mask[i]]
        return frame

    def vowels(self):
        return [n for n, v in zip(self.names, self.is_vowel) if v]

    def ipa_case(self, i, sr=SR):
        """-> list of (frame|None, M, F) incl. the trailing NULL(150 ms, 0) of test_speakIpa.py:27."""
        a, b = self.ipa_start[i], self.ipa_start[i + 1]
        out = []
        for k in range(a, b):
            fr = None if self.ipa_isnull[k] else self.ipa_frames[k].copy()
            out.append((fr, ms(self.ipa_dur_ms[k], sr), ms(self.ipa_fade_ms[k], sr)))
        out.append((None, ms(150, sr), 0))
        return out

    def find_ipa(self, line, speed=1.0, clause=0, pitch=100.0, infl=0.5):
        for i, m in enumerate(self.ipa_meta):
            if int(m[0]) == line and m[1] == speed and int(m[2]) == clause and m[3] == pitch and m[4] == infl:
                return i
        raise KeyError((line, speed, clause, pitch, infl))


def vowel_frame(ref, name, pitch, end_pitch=None, out_gain=1.0):
    """test_playVowelchart.py:27-30 + ipa.setFrame."""
    f = np.zeros(47)
    f[PREGAIN] = 1.0
    f[VOICEAMP] = 1.0
    f[OUTGAIN] = out_gain
    ref.set_frame(f, name)
    f[VOICEPITCH] = pitch
    f[ENDPITCH] = pitch if end_pitch is None else end_pitch
    return f


def q(frame, m, f, index=-1, purge=False):
    return ("q", None if frame is None else np.asarray(frame, dtype=np.float64), int(m), int(f), int(index), bool(purge))


class Scenario:
    def __init__(self, name, ops, sr=SR, seed=0, batchable=False):
        self.name, self.ops, self.sr, self.seed, self.batchable = name, ops, sr, seed, batchable

    def frames(self):
        """(frames[n,47], min, fade, index, isnull) of the queue ops (batchable scenarios)."""
        fr, m, f, ix, nu = [], [], [], [], []
        for op in self.ops:
            if op[0] == "q":
                assert not op[5]
                fr.append(np.zeros(47) if op[1] is None else op[1])
                m.append(op[2]); f.append(op[3]); ix.append(op[4]); nu.append(op[1] is None)
        return (np.array(fr).reshape(-1, 47), np.array(m, np.uint32), np.array(f, np.uint32),
                np.array(ix, np.int32), np.array(nu, np.uint8))


def build_scenarios(ref):
    sc = []
    # cfg0: SURVEY 8(c) recipe == test_playVowelchart.py frame set-up, /a/ at 120 Hz, 1 s
    fa = vowel_frame(ref, "a", 120.0)
    sc.append(Scenario("cfg0_a_1s", [q(fa, ms(1000), ms(50)), ("s", 22050), ("s", 22050)], batchable=False))
    sc.append(Scenario("cfg0_a_1s_batch", [q(fa, ms(1000), ms(50)), ("drain",)], batchable=True))

    # steady vowels (BASELINE cfg1 recipe at 0.25 s): every vowel at three pitches
    for vi, v in enumerate(ref.vowels()):
        for pi, pitch in enumerate((80.0, 163.5, 320.0)):
            f = vowel_frame(ref, v, pitch)
            sc.append(Scenario("vowel_%02d_p%d" % (vi, pi),
                               [q(f, ms(250), ms(50)), q(None, ms(50), ms(50)), ("drain",)], batchable=True))

    # sampleIpa.txt through the reference frame producer (test_speakIpa.py:25-27)
    for i in range(len(ref.ipa_meta)):
        ops = [q(fr, m, f) for (fr, m, f) in ref.ipa_case(i)]
        ops.append(("drain",))
        li, speed, clause, pitch, infl = ref.ipa_meta[i]
        sc.append(Scenario("ipa_l%d_s%02d_c%d_p%d_i%02d" % (li, speed * 10, clause, pitch, infl * 10), ops,
                           seed=1000 + i, batchable=True))

    # the same line pulled in uneven chunks (streaming contract, __init__.py:67)
    i0 = ref.find_ipa(2)
    ops = [q(fr, m, f, index=k) for k, (fr, m, f) in enumerate(ref.ipa_case(i0))]
    for n in (8192, 1, 77, 1000, 64, 63, 65, 4096):
        ops.append(("s", n))
    ops.append(("drain",))
    sc.append(Scenario("stream_chunks", ops, seed=7))

    # purge during steady state and during a fade, drain, then resume (frame.cpp:103-112;
    # test_midiSing.py:105-132 usage pattern)
    fs = vowel_frame(ref, "s", 110.0); fz = vowel_frame(ref, "z", 130.0, 90.0)
    fm = vowel_frame(ref, "m", 100.0, 140.0); fi = vowel_frame(ref, "i", 200.0, 100.0)
    ops = [q(fa, 4000, 500, index=1), q(fs, 3000, 800, index=2), q(fz, 3000, 800, index=3), ("s", 3000),
           q(fm, 2500, 600, index=4, purge=True), ("s", 300),       # purge in steady state of fa
           q(fi, 2000, 700, index=5, purge=True), ("s", 2500),      # purge inside the fade into fm
           q(None, 500, 300, index=6), ("drain",),
           q(fz, 1500, 200, index=7), q(None, 0, 441, index=-1, purge=True), ("s", 100), ("drain",),
           q(fs, 1200, 100, index=8), q(None, 300, 300), ("drain",)]
    sc.append(Scenario("purge_resume", ops, seed=11))

    # vowel-chart demo call pattern (test_playVowelchart.py:32-44)
    voiced = [ref.names[i] for i in ref.voiced_order]
    ops = []
    for a, b in ((voiced[0], voiced[5]), (voiced[9], voiced[20]), (voiced[30], voiced[3])):
        ops.append(q(None, 0, ms(20), purge=True))
        ops.append(q(vowel_frame(ref, a, 40.0, 300.0), ms(300), ms(50)))
        ops.append(q(vowel_frame(ref, b, 300.0, 40.0), ms(500), ms(400)))
        ops.append(q(None, ms(50), ms(50)))
        ops.append(("s", 9000))
    ops.append(("drain",))
    sc.append(Scenario("vowelchart_pairs", ops, seed=3))

    # vibrato + breathiness + open quotient (test_sayHannah.py:14-30, test_midiSing.py:60-61)
    def hannah(name, pitch, amp=1.0):
        f = vowel_frame(ref, name, pitch)
        f[VIBOFFSET] = 0.1; f[VIBSPEED] = 5.5; f[VOICEAMP] = amp
        f[TURB] = 0.3; f[OPENQ] = 0.4
        return f
    ops = [q(hannah("æ", 150.0, 0.0), ms(120), ms(100)), q(hannah("æ", 150.0), ms(120), ms(40)),
           q(hannah("n", 100.0), ms(120), ms(40)), q(hannah("ɑ", 90.0), ms(80), ms(40)),
           q(None, ms(40), ms(40)), ("drain",)]
    sc.append(Scenario("hannah_vibrato", ops, seed=5, batchable=True))

    # NaN means "hold the previous value" (utils.h:21)
    fn = vowel_frame(ref, "u", 140.0, 100.0)
    for k in (8, 16, 26, 38, BYPASS, OUTGAIN, VIBSPEED):
        fn[k] = np.nan
    ops = [q(vowel_frame(ref, "e", 120.0), 2000, 400), q(fn, 3000, 1500), q(vowel_frame(ref, "o", 90.0), 1500, 300),
           q(None, 400, 400), ("drain",)]
    sc.append(Scenario("nan_hold", ops, seed=9, batchable=True))

    # duration edge cases: fade longer than the frame, zero fade (clamped to 1), zero-length silences
    ops = [q(None, 0, 0), q(fa, 10, 2000), q(fs, 1, 1), q(fz, 2, 0), q(None, 0, 50), q(fm, 700, 699),
           q(fi, 700, 700), q(fa, 700, 701), q(None, 1, 1), ("drain",)]
    sc.append(Scenario("duration_edges", ops, seed=13, batchable=True))

    # 16 kHz, the NVDA driver's rate (__init__.py:137,144)
    i1 = ref.find_ipa(0)
    ops = [q(fr, m, f) for (fr, m, f) in ref.ipa_case(i1, sr=16000)] + [("drain",)]
    sc.append(Scenario("ipa_l0_16k", ops, sr=16000, seed=21, batchable=True))

    # other sample rates the C-ABI accepts (reference src/speechPlayer.cpp:25-32 takes any): everything that depends on the rate --
    # the resonator coefficients, the phase increments, ms -> samples, the tracks' evaluation -- at 44.1 kHz and at 8 kHz
    # (at 8 kHz the upper formants lie beyond the Nyquist frequency: the coefficients simply follow the formulas)
    for sr, line, seed in ((44100, 1, 23), (8000, 3, 25)):
        ops = [q(fr, m, f) for (fr, m, f) in ref.ipa_case(ref.find_ipa(line), sr=sr)] + [("drain",)]
        sc.append(Scenario("ipa_l%d_%dk" % (line, sr // 1000), ops, sr=sr, seed=seed, batchable=True))
        fv = vowel_frame(ref, "ɑ", 110.0, 170.0); fv[VIBOFFSET] = 0.1; fv[VIBSPEED] = 5.5
        ops = [q(fv, ms(180, sr), ms(40, sr)), q(vowel_frame(ref, "s", 120.0), ms(90, sr), ms(30, sr)),
               q(vowel_frame(ref, "m", 100.0, 90.0), ms(120, sr), ms(50, sr)), q(None, ms(40, sr), ms(40, sr)), ("drain",)]
        sc.append(Scenario("vowel_fric_nasal_%dk" % (sr // 1000), ops, sr=sr, seed=seed + 1, batchable=True))
    return sc


def play_oracle(scn):
    """-> (list of int16 arrays, one per synth/drain op; list of lastIndex after each such op)"""
    from tests import oracle
    p = oracle.OraclePlayer(scn.sr, noise=oracle.NOISE_COUNTER, seed=scn.seed)
    pcm, marks = [], []
    for op in scn.ops:
        if op[0] == "q":
            p.queue(op[1], op[2], op[3], op[4], op[5])
        elif op[0] == "s":
            pcm.append(p.synthesize(op[1])); marks.append(p.last_index())
        else:
            pcm.append(p.drain()); marks.append(p.last_index())
    p.close()
    return pcm, marks


STORED_PCM = ("cfg0_a_1s", "stream_chunks", "purge_resume", "vowelchart_pairs", "hannah_vibrato", "nan_hold",
              "duration_edges", "ipa_l0_16k", "ipa_l1_44k", "ipa_l3_8k", "vowel_fric_nasal_44k", "vowel_fric_nasal_8k")


def write_expected_pcm(ref_path, outdir):
    """Called by make_golden.py: expected PCM (oracle, counter noise) for every scenario as SHA-1 +
    length, and the full PCM for a small subset."""
    ref = Ref(ref_path)
    table, store = {}, {}
    for scn in build_scenarios(ref):
        pcm, marks = play_oracle(scn)
        flat = np.concatenate(pcm) if pcm else np.zeros(0, np.int16)
        table[scn.name] = {"sha1": hashlib.sha1(flat.tobytes()).hexdigest(), "samples": int(len(flat)),
                           "calls": [int(len(x)) for x in pcm], "marks": [int(m) for m in marks]}
        if scn.name in STORED_PCM or (scn.name.startswith("ipa_") and "_s10_c0_p100_i05" in scn.name):
            store[scn.name] = flat
    with open(os.path.join(outdir, "expected.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(outdir, "expected_pcm.npz"), **store)
    print("expected.json: %d scenarios; expected_pcm.npz: %d stored" % (len(table), len(store)))

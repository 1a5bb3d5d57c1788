"""Synthetic frame-stream batches for BASELINE.json's configurations.

The phoneme parameter vectors and the sampleIpa.txt frame streams are the ones the
reference's own frame producer emits (captured as data in tests/golden/ref_frames.npz
by tests/golden/make_golden.py); this module only instances them into batches, following
the recipes of SURVEY.md section 8(d).
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FRAMES = os.path.join(ROOT, "tests", "golden", "ref_frames.npz")
SR = 22050

VOICEPITCH, VOICEAMP, PREGAIN, OUTGAIN, ENDPITCH = 0, 5, 44, 45, 46


def ms(x, sr=SR):
    """reference speechPlayer.py:53"""
    return int(x * (sr / 1000.0))


class Batch(dict):
    """frames[nF,47] f64, min[nF] u32, fade[nF] u32, index[nF] i32, isnull[nF] u8,
    frame_start[nU+1] i64, seeds[nU] u32, name, sr"""

    @property
    def n_utt(self):
        return len(self["frame_start"]) - 1

    def sample_counts(self):
        m = self["min"].astype(np.int64)
        f = np.maximum(self["fade"].astype(np.int64), 1)
        per = np.maximum(m, f + 1) + 1
        c = np.concatenate([[0], np.cumsum(per)])
        fs = self["frame_start"]
        return c[fs[1:]] - c[fs[:-1]]

    def algorithmic_bytes(self):
        """2 B per output sample + 388 B per frame read."""
        return 2 * int(self.sample_counts().sum()) + 388 * int(len(self["min"]))

    def slice(self, first, count):
        fs = self["frame_start"]
        a, b = int(fs[first]), int(fs[first + count])
        out = Batch(frames=self["frames"][a:b], min=self["min"][a:b], fade=self["fade"][a:b], index=self["index"][a:b],
                    isnull=self["isnull"][a:b], frame_start=(fs[first:first + count + 1] - a).astype(np.int64),
                    seeds=self["seeds"][first:first + count], name=self["name"], sr=self["sr"])
        return out


def _load():
    return np.load(REF_FRAMES)


def cfg1_steady_vowels(n_utt=4096, seconds=1.0, first=0, sr=SR):
    """BASELINE configs[1]: steady vowels, full vowel-chart sweep, `seconds` each.
    Utterance u: vowel = sorted(_isVowel phonemes)[u mod 20]; frame = zero + preFormantGain =
    voiceAmplitude = outputGain = 1 (reference test_playVowelchart.py:27-30) + the phoneme's fields;
    pitch 80..320 Hz over (u div 20); one frame (M = seconds, F = 50 ms) + NULL(50 ms, 50 ms)."""
    z = _load()
    names = [b.decode("utf8") for b in z["phoneme_names"]]
    vowels = [i for i in range(len(names)) if z["phoneme_isVowel"][i]]
    protos = np.zeros((len(vowels), 47))
    for j, i in enumerate(vowels):
        f = np.zeros(47)
        f[PREGAIN] = 1.0; f[VOICEAMP] = 1.0; f[OUTGAIN] = 1.0
        mask = z["phoneme_mask"][i].astype(bool)
        f[mask] = z["phoneme_frames"][i][mask]
        protos[j] = f
    u = first + np.arange(n_utt)
    frames = np.zeros((n_utt * 2, 47))
    frames[0::2] = protos[u % len(vowels)]
    pitch = 80.0 * 2.0 ** (((u // len(vowels)) % 205) / 205.0 * 2.0)
    frames[0::2, VOICEPITCH] = pitch
    frames[0::2, ENDPITCH] = pitch
    M, F = ms(1000 * seconds, sr), ms(50, sr)
    return Batch(frames=frames, min=np.tile(np.array([M, F], np.uint32), n_utt),
                 fade=np.tile(np.array([F, F], np.uint32), n_utt), index=np.full(n_utt * 2, -1, np.int32),
                 isnull=np.tile(np.array([0, 1], np.uint8), n_utt),
                 frame_start=np.arange(n_utt + 1, dtype=np.int64) * 2,
                 seeds=(np.arange(n_utt) + first).astype(np.uint32),
                 name="cfg1: %d steady vowels x %.3g s (vowel-chart sweep)" % (n_utt, seconds), sr=sr)


def cfg2_ipa_utterances(n_utt=65536, first=0, max_seconds=None, sr=SR):
    """BASELINE configs[2] (and configs[3] with max_seconds=0.5): utterance u = sampleIpa.txt line
    (u mod 8) through the frame producer (nvspeechplayer_amd/ipa.py, identical to the reference's on the
    captured cases) with speed 1, inflection 0.5, clause '.', base pitch 100*2^(((u div 8) mod 64 - 32)/64) Hz,
    + NULL(150 ms, 0) (reference test_speakIpa.py:25-27); noise seed = u.  SURVEY.md section 8(d)."""
    from . import ipa
    z = _load()
    lines = [b.decode("utf8") for b in z["ipa_lines"]][:8]
    cache = {}

    def stream(line, variant):
        key = (line, variant)
        if key not in cache:
            pk = ipa.frames_for_batch([lines[line]], sampleRate=sr, speed=1.0, basePitch=100.0 * 2.0 ** ((variant - 32) / 64.0),
                                      inflection=0.5, clauseType=".", trailing_silence_ms=150.0)
            fr, nu, M, F = pk["frames"], pk["isnull"], pk["min"], pk["fade"]
            if max_seconds is not None:
                per = np.maximum(M.astype(np.int64), np.maximum(F.astype(np.int64), 1) + 1) + 1
                keep = max(1, int(np.searchsorted(np.cumsum(per), max_seconds * sr, side="right")))
                fr, nu, M, F = fr[:keep], nu[:keep], M[:keep], F[:keep]
            cache[key] = (fr, nu, M, F)
        return cache[key]

    frames, mins, fades, nul, fs = [], [], [], [], [0]
    for k in range(n_utt):
        u = first + k
        fr, nu, M, F = stream(u % 8, (u // 8) % 64)
        frames.append(fr); mins.append(M); fades.append(F); nul.append(nu)
        fs.append(fs[-1] + len(M))
    nF = fs[-1]
    tag = "cfg2" if max_seconds is None else "cfg3-slice(<=%.2gs)" % max_seconds
    return Batch(frames=np.concatenate(frames), min=np.concatenate(mins), fade=np.concatenate(fades),
                 index=np.full(nF, -1, np.int32), isnull=np.concatenate(nul), frame_start=np.array(fs, np.int64),
                 seeds=(np.arange(n_utt) + first).astype(np.uint32),
                 name="%s: %d sampleIpa utterances, 64 pitch variants" % (tag, n_utt), sr=sr)


def cfg4_voice_variants(n_variants=256, utt_per_variant=16384, first_variant=0, sr=SR):
    """BASELINE configs[4]: voice-parameter variants x utterances with per-frame pitch/formant glides.
    Variant v draws multipliers from a generator seeded 1234+v, in the style of the NVDA driver's voice
    presets (reference nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:86-125): cf1..cf3 x U[0.75, 1.05],
    cb1 x U[1, 1.3], voicePitch and endVoicePitch x U[0.75, 1.5], fricationAmplitude x U[0.7, 1],
    pa6 x U[1, 1.3]; its utterances are the cfg2 generator's."""
    base = cfg2_ipa_utterances(utt_per_variant, first=0, sr=sr)
    F = {n: i for i, n in enumerate(["voicePitch", "vibratoPitchOffset", "vibratoSpeed", "voiceTurbulenceAmplitude",
                                     "glottalOpenQuotient", "voiceAmplitude", "aspirationAmplitude"])}
    frames, seeds = [], []
    for k in range(n_variants):
        v = first_variant + k
        rng = np.random.default_rng(1234 + v)
        g = base["frames"].copy()
        g[:, 7:10] *= rng.uniform(0.75, 1.05, size=3)          # cf1..cf3
        g[:, 15] *= rng.uniform(1.0, 1.3)                      # cb1
        pm = rng.uniform(0.75, 1.5)
        g[:, 0] *= pm; g[:, 46] *= pm                          # voicePitch, endVoicePitch
        g[:, 24] *= rng.uniform(0.7, 1.0)                      # fricationAmplitude
        g[:, 42] *= rng.uniform(1.0, 1.3)                      # pa6
        frames.append(g)
        seeds.append(base["seeds"].astype(np.uint64) + np.uint64(v) * np.uint64(utt_per_variant))
    nF = len(base["min"])
    fs = np.concatenate([base["frame_start"][:-1] + k * nF for k in range(n_variants)] + [[n_variants * nF]])
    return Batch(frames=np.concatenate(frames), min=np.tile(base["min"], n_variants), fade=np.tile(base["fade"], n_variants),
                 index=np.full(nF * n_variants, -1, np.int32), isnull=np.tile(base["isnull"], n_variants),
                 frame_start=fs.astype(np.int64), seeds=(np.concatenate(seeds) & np.uint64(0xFFFFFFFF)).astype(np.uint32),
                 name="cfg4: %d voice variants x %d utterances" % (n_variants, utt_per_variant), sr=sr)


def make(workload, n_utt=None, first=0):
    if workload == "cfg1":
        return cfg1_steady_vowels(n_utt or 4096, first=first)
    if workload == "cfg2":
        return cfg2_ipa_utterances(n_utt or 65536, first=first)
    if workload == "cfg3":
        return cfg2_ipa_utterances(n_utt or 131072, first=first, max_seconds=0.5)
    if workload == "cfg4":     # n_utt utterances per variant block of 1024; the full config is 32 variants x 16384 per GPU
        per = 1024
        return cfg4_voice_variants(max(1, (n_utt or 32768) // per), per, first_variant=first // per)
    raise ValueError(workload)

"""Synthetic frame-stream batches for BASELINE.json's configurations.

Inputs are data shipped with the package (nvspeechplayer_amd/data/workload_inputs.npz: the sampleIpa.txt lines
and the phoneme parameter vectors, dumped by tests/golden/make_golden.py); the speech streams come out of the
native frame producer (csrc/frame_producer.cpp through ipa.frames_for_batch -- one C call per batch, no Python
loop per utterance).  Recipes: SURVEY.md section 8(d).
"""
import os

import numpy as np

INPUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "workload_inputs.npz")
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
    return np.load(INPUTS)


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
    u = first + np.arange(n_utt, dtype=np.int64)
    variant = (u // 8) % 64
    pitch_of = np.array([100.0 * 2.0 ** ((v - 32) / 64.0) for v in range(64)])      # libm pow per variant (numpy's differs in the last place)
    pk = ipa.frames_for_batch([lines[i] for i in (u % 8)], sampleRate=sr, speed=1.0, basePitch=pitch_of[variant],
                              inflection=0.5, clauseType=".", trailing_silence_ms=150.0)
    fs, frames, M, F, nul = pk["frame_start"], pk["frames"], pk["min"], pk["fade"], pk["isnull"]
    if max_seconds is not None:
        # keep the leading frames of every utterance whose spans end within max_seconds (at least one frame)
        per = np.maximum(M.astype(np.int64), np.maximum(F.astype(np.int64), 1) + 1) + 1
        c = np.cumsum(per)
        before = np.concatenate([[0], c])[fs[:-1]]                    # samples before each utterance
        owner = np.repeat(np.arange(n_utt), np.diff(fs))
        within = (c - before[owner]) <= max_seconds * sr
        within[fs[:-1]] = True
        frames, M, F, nul = frames[within], M[within], F[within], nul[within]
        fs = np.concatenate([[0], np.cumsum(np.bincount(owner[within], minlength=n_utt))]).astype(np.int64)
    nF = int(fs[-1])
    tag = "cfg2" if max_seconds is None else "cfg3-slice(<=%.2gs)" % max_seconds
    return Batch(frames=frames, min=M, fade=F, index=np.full(nF, -1, np.int32), isnull=nul, frame_start=fs,
                 seeds=(np.arange(n_utt) + first).astype(np.uint32),
                 name="%s: %d sampleIpa utterances, 64 pitch variants" % (tag, n_utt), sr=sr)


def cfg4_voice_variants(n_variants=256, utt_per_variant=16384, first_variant=0, sr=SR, first_utt=0, n_utt=None):
    """BASELINE configs[4]: voice-parameter variants x utterances with per-frame pitch/formant glides.
    Variant v draws multipliers from a generator seeded 1234+v, in the style of the NVDA driver's voice
    presets (reference nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:86-125): cf1..cf3 x U[0.75, 1.05],
    cb1 x U[1, 1.3], voicePitch and endVoicePitch x U[0.75, 1.5], fricationAmplitude x U[0.7, 1],
    pa6 x U[1, 1.3]; its utterances are the cfg2 generator's.
    The config is the flat list (variant, utterance); `first_utt` / `n_utt` cut a contiguous piece out of the
    n_variants x utt_per_variant list that starts at variant `first_variant` (for the shards of a node)."""
    base = cfg2_ipa_utterances(utt_per_variant, first=0, sr=sr)
    total = n_variants * utt_per_variant
    lo = first_utt
    hi = total if n_utt is None else min(total, first_utt + n_utt)
    bfs = base["frame_start"]
    # the pieces (variant, first utterance, end) of the flat list, then ONE allocation of the shard's arrays, filled piece by
    # piece (a list of per-variant copies and a concatenate held the frames twice: 2 x 4.7 GB per rank at configs[4]'s per-GPU size)
    pieces = []
    for k in range(lo // utt_per_variant, (max(hi, lo + 1) - 1) // utt_per_variant + 1):
        a, b = max(lo, k * utt_per_variant) - k * utt_per_variant, min(hi, (k + 1) * utt_per_variant) - k * utt_per_variant
        if b > a:
            pieces.append((first_variant + k, a, b))
    nU = sum(b - a for _, a, b in pieces)
    nF = sum(int(bfs[b] - bfs[a]) for _, a, b in pieces)
    frames = np.empty((nF, 47)); mins = np.empty(nF, np.uint32); fades = np.empty(nF, np.uint32); nul = np.empty(nF, np.uint8)
    fs = np.zeros(nU + 1, np.int64); seeds = np.empty(nU, np.uint32)
    atF = atU = 0
    for v, a, b in pieces:
        n = int(bfs[b] - bfs[a])
        rng = np.random.default_rng(1234 + v)
        g = frames[atF:atF + n]
        g[:] = base["frames"][bfs[a]:bfs[b]]
        g[:, 7:10] *= rng.uniform(0.75, 1.05, size=3)          # cf1..cf3
        g[:, 15] *= rng.uniform(1.0, 1.3)                      # cb1
        pm = rng.uniform(0.75, 1.5)
        g[:, 0] *= pm; g[:, 46] *= pm                          # voicePitch, endVoicePitch
        g[:, 24] *= rng.uniform(0.7, 1.0)                      # fricationAmplitude
        g[:, 42] *= rng.uniform(1.0, 1.3)                      # pa6
        mins[atF:atF + n] = base["min"][bfs[a]:bfs[b]]; fades[atF:atF + n] = base["fade"][bfs[a]:bfs[b]]; nul[atF:atF + n] = base["isnull"][bfs[a]:bfs[b]]
        fs[atU + 1:atU + 1 + (b - a)] = atF + (bfs[a + 1:b + 1] - bfs[a])
        seeds[atU:atU + (b - a)] = ((base["seeds"][a:b].astype(np.uint64) + np.uint64(v) * np.uint64(utt_per_variant)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        atF += n; atU += b - a
    return Batch(frames=frames, min=mins, fade=fades, index=np.full(nF, -1, np.int32), isnull=nul, frame_start=fs, seeds=seeds,
                 name="cfg4: %d voice variants x %d utterances" % (n_variants, utt_per_variant), sr=sr)


# ---- the same recipes in COMPACT form: what a caller hands BatchPlayer.setIpa (texts, pitches, voices) or setUtterancesShared ------
CFG2_PERIOD = 512      # 8 sampleIpa lines x 64 base pitches: the distinct frame streams of configs[2] / [3] (and of one voice of configs[4])


def cfg2_spec(n_utt=65536, first=0, sr=SR):
    """BASELINE configs[2] as arguments of BatchPlayer.setIpa: the eight sampleIpa lines once, per utterance which line it speaks,
    its base pitch and its noise seed.  bp.setIpa(**spec) gives the batch cfg2_ipa_utterances(n_utt, first) describes."""
    z = _load()
    lines = [b.decode("utf8") for b in z["ipa_lines"]][:8]
    u = first + np.arange(n_utt, dtype=np.int64)
    pitch_of = np.array([100.0 * 2.0 ** ((v - 32) / 64.0) for v in range(64)])
    return dict(texts=lines, textOf=(u % 8).astype(np.int64), basePitch=pitch_of[(u // 8) % 64], speed=1.0, inflection=0.5, clauseType=".",
                trailing_silence_ms=150.0, noiseSeed=u.astype(np.uint32))


def cfg4_voice(v):
    """Voice variant v of configs[4] as a defined voice (nvspeechplayer_amd.ipa.defineVoice): the multipliers cfg4_voice_variants draws."""
    from . import ipa
    rng = np.random.default_rng(1234 + v)
    cf = rng.uniform(0.75, 1.05, size=3)
    cb1 = rng.uniform(1.0, 1.3)
    pm = rng.uniform(0.75, 1.5)
    fr = rng.uniform(0.7, 1.0)
    pa6 = rng.uniform(1.0, 1.3)
    return ipa.defineVoice("cfg4-variant-%d" % v, {0: (None, pm), 7: (None, cf[0]), 8: (None, cf[1]), 9: (None, cf[2]), 15: (None, cb1),
                                                   24: (None, fr), 42: (None, pa6), 46: (None, pm)})


def cfg4_spec(n_utt, first=0, per=16384, sr=SR):
    """BASELINE configs[4] as arguments of BatchPlayer.setIpa: utterance i of the flat (variant, utterance) list speaks with the defined
    voice of variant i // per and is utterance i % per of the cfg2 generator; noise seeds as cfg4_voice_variants sets them."""
    i = first + np.arange(n_utt, dtype=np.int64)
    v, w = i // per, i % per
    spec = cfg2_spec(1, 0, sr)
    pitch_of = np.array([100.0 * 2.0 ** ((k - 32) / 64.0) for k in range(64)])
    index_of = np.zeros(int(v.max()) + 1 if n_utt else 1, np.int32)
    for k in np.unique(v):
        index_of[int(k)] = cfg4_voice(int(k))
    spec.update(textOf=(w % 8).astype(np.int64), basePitch=pitch_of[(w // 8) % 64], voice=index_of[v],
                noiseSeed=((w + v * per) & 0xFFFFFFFF).astype(np.uint32))
    return spec


def shared(workload, n_utt=None, first=0):
    """A configuration as frame lists that utterances SHARE (BatchPlayer.setUtterancesShared): configs[2] / [3] are 512 distinct streams
    instanced.  -> (lists: Batch of the 512 streams, list_of[n_utt], seeds[n_utt])."""
    assert workload in ("cfg2", "cfg3")
    n = n_utt or PER_GPU[workload]
    lists = make(workload, CFG2_PERIOD, 0)
    u = first + np.arange(n, dtype=np.int64)
    return lists, (u % CFG2_PERIOD).astype(np.uint32), u.astype(np.uint32)


# utterances per GPU of each BASELINE configuration (configs[3] is 10^6 over 8 GPUs, configs[4] 256 x 16384 over 8)
PER_GPU = {"cfg1": 4096, "cfg2": 65536, "cfg3": 125000, "cfg4": 32 * 16384}
CFG4_PER_VARIANT = 16384


def make(workload, n_utt=None, first=0):
    """Utterances first .. first + n_utt - 1 of a configuration's (unbounded) utterance list."""
    if workload == "cfg1":
        return cfg1_steady_vowels(n_utt or 4096, first=first)
    if workload == "cfg2":
        return cfg2_ipa_utterances(n_utt or 65536, first=first)
    if workload == "cfg3":
        return cfg2_ipa_utterances(n_utt or PER_GPU["cfg3"], first=first, max_seconds=0.5)
    if workload == "cfg4":
        # the flat (variant, utterance) list with 16384 utterances per variant; counts that are not a multiple of
        # 16384 use blocks of 1024 per variant instead (small tests keep several variants)
        n = n_utt or PER_GPU["cfg4"]
        per = CFG4_PER_VARIANT if (n % CFG4_PER_VARIANT == 0 and first % CFG4_PER_VARIANT == 0) or n > 65536 else 1024
        nv = (first + n + per - 1) // per
        b = cfg4_voice_variants(nv, per, first_utt=first, n_utt=n)
        b["name"] = "cfg4: utterances %d..%d of voice variants x %d utterances" % (first, first + n - 1, per)
        return b
    raise ValueError(workload)


def sample_counts(workload, n_utt, first=0):
    """Samples per utterance of utterances first .. first + n_utt - 1 without building their frames: every recipe's timing
    repeats with a short period (8 lines x 64 pitch variants; the voice variants of cfg4 change no duration)."""
    period = 1 if workload == "cfg1" else 512
    one = make("cfg2" if workload == "cfg4" else workload, period, 0).sample_counts()
    return one[(first + np.arange(n_utt, dtype=np.int64)) % period]


def rotated(b, seed=1):
    """Every utterance keeps its frames (and so its length) but starts at a random one of them: waves still hold
    64 utterances of equal length after the sort, yet no two lanes fade at the same time -- the closest cheap
    stand-in for a batch of 64 different sentences of similar length."""
    rng = np.random.default_rng(seed)
    fs = b["frame_start"]
    perm = np.arange(len(b["min"]))
    for u in range(b.n_utt):
        a, e = int(fs[u]), int(fs[u + 1]) - 1          # the last frame (the closing NULL frame) stays last
        if e - a > 1:
            k = int(rng.integers(0, e - a))
            perm[a:e] = np.roll(np.arange(a, e), -k)
    out = {k: b[k][perm] for k in ("frames", "min", "fade", "index", "isnull")}
    return Batch(frame_start=fs, seeds=b["seeds"], name=b["name"] + " rotated", sr=b["sr"], **out)


def jittered(b, seed=1):
    """Every frame's duration and fade scaled by its own random factor in [0.7, 1.3]: no two utterances of the batch share a
    timing or a length -- what a batch of unrelated sentences looks like to the kernels."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(0.7, 1.3, size=len(b["min"]))
    out = {key: b[key] for key in ("frames", "index", "isnull")}
    out["min"] = np.maximum(1, (b["min"] * k)).astype(np.uint32)
    out["fade"] = (b["fade"] * k).astype(np.uint32)
    return Batch(frame_start=b["frame_start"], seeds=b["seeds"], name=b["name"] + " jittered", sr=b["sr"], **out)


def distinct(b, seed=1):
    """Every frame's formant frequencies scaled by its own random factor (1 +- 0.5 %): no two fades of the batch are alike, so
    nothing shares a track (klatt_tracks.h).  With `jittered` on top: a batch in which nothing is shared AND nothing is aligned --
    65 536 different sentences in different voices, as the kernels see them (bench.py: all_different)."""
    rng = np.random.default_rng(seed)
    fr = b["frames"].copy()
    k = rng.uniform(0.995, 1.005, size=(len(fr), 1))
    fr[:, 7:15] *= k
    fr[:, 25:31] *= k
    out = {key: b[key] for key in ("min", "fade", "index", "isnull")}
    return Batch(frame_start=b["frame_start"], seeds=b["seeds"], name=b["name"] + " distinct", sr=b["sr"], frames=fr, **out)


def all_different(b, seed=1):
    return jittered(distinct(b, seed), seed)

"""The batch in compact form on the GPU (include/speechPlayer_batch.h: speechPlayer_batch_setRecords / _setUtterancesShared / _setIpa /
_setIpaVoices): frames the DEVICE builds from 32-byte records (klatt_expand_frames) against the frames captured from the reference's
own producer (tests/golden/ref_frames.npz: ipa.generateFramesAndTiming and applyVoiceToFrame) bit for bit, PCM of shared frame lists
against the same utterances as a plain batch and against the oracle, and the device-side verification of frames the planner
recognised by their hash (klatt_verify_shared).  Needs a GPU."""
import numpy as np
import pytest

from tests import oracle, scenarios

pytestmark = pytest.mark.gpu

CLAUSES = {0: ".", 1: ",", 2: "?", 3: "!", 4: None}


@pytest.fixture(scope="module")
def ref():
    return scenarios.Ref()


@pytest.fixture(scope="module")
def golden():
    return np.load(scenarios.GOLDEN + "/ref_frames.npz")


def ms_to_samples(ms, sr):
    """reference speechPlayer.py:53"""
    return np.array([int(x * (sr / 1000.0)) for x in ms], np.uint32)


def test_frames_built_on_the_device_equal_the_reference_producers(ref, golden):
    """Every captured case of the reference's generateFramesAndTiming (126: eight sampleIpa lines x clause types x speeds, pitch and
    inflection variants, the extra lines) through speechPlayer_batch_setIpa: the frames downloaded from HBM -- expanded there from
    32-byte records -- equal the captured frames value for value, silences and durations (in samples) included; and they equal what the
    host's ipa.frames_for_batch packs."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import ipa
    lines = [b.decode("utf8") for b in golden["ipa_lines"]]
    groups = {}
    for i, meta in enumerate(ref.ipa_meta):
        groups.setdefault((float(meta[1]), float(meta[4])), []).append(i)      # one call per (speed, inflection)
    n_frames = 0
    for sr in (22050, 16000):
        bp = eng.BatchPlayer(sr)
        for (speed, infl), cases in groups.items():
            texts = [lines[int(ref.ipa_meta[i][0])] for i in cases]
            clauses = [CLAUSES[int(ref.ipa_meta[i][2])] for i in cases]
            pitch = [float(ref.ipa_meta[i][3]) for i in cases]
            bp.setIpa(texts, speed=speed, basePitch=pitch, inflection=infl, clauseType=clauses, trailing_silence_ms=None)
            pk = ipa.frames_for_batch(texts, sampleRate=sr, speed=speed, basePitch=pitch, inflection=infl, clauseType=clauses, trailing_silence_ms=None)
            for u, i in enumerate(cases):
                a, b = ref.ipa_start[i], ref.ipa_start[i + 1]
                fr, m, f, ix, nul = bp.frames(u)
                assert len(nul) == b - a, (i, len(nul), b - a)
                assert np.array_equal(nul, ref.ipa_isnull[a:b]), i
                real = nul == 0
                assert np.array_equal(fr[real], ref.ipa_frames[a:b][real]), (i, np.argwhere(fr[real] != ref.ipa_frames[a:b][real])[:5])
                assert not fr[~real].any()
                assert np.array_equal(m, ms_to_samples(ref.ipa_dur_ms[a:b], sr)), i
                assert np.array_equal(f, np.maximum(ms_to_samples(ref.ipa_fade_ms[a:b], sr), 1)), i      # (fade >= 1: reference src/speechPlayer.cpp:36)
                assert (ix == -1).all()
                k0, k1 = pk["frame_start"][u], pk["frame_start"][u + 1]
                assert np.array_equal(fr, pk["frames"][k0:k1]) and np.array_equal(m, pk["min"][k0:k1]) and np.array_equal(nul, pk["isnull"][k0:k1])
                n_frames += (b - a) if sr == 22050 else 0
        bp.close()
    assert n_frames == len(ref.ipa_frames) and len(ref.ipa_meta) == 126


def test_voice_presets_applied_before_the_device_expansion(golden):
    """The 644 captured preset frames (four presets x every frame of the eight sampleIpa lines, reference applyVoiceToFrame): ONE
    setIpa call with a voice per utterance; the frames in HBM equal the captured ones."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import ipa
    names = [b.decode("utf8") for b in golden["voice_names"]]
    lines = [b.decode("utf8") for b in golden["ipa_lines"]]
    meta, want = golden["voice_case_meta"], golden["voice_case_frames"]
    bp = eng.BatchPlayer(22050)
    voice = [ipa.voiceIndex(n) for n in names for _ in range(8)]
    assert voice == [v for v in range(4) for _ in range(8)]
    bp.setIpa(lines[:8], textOf=list(range(8)) * 4, speed=1.0, basePitch=100.0, inflection=0.5, clauseType=".", voice=voice, trailing_silence_ms=None)
    k = 0
    for vi in range(4):
        for li in range(8):
            fr, _, _, _, nul = bp.frames(vi * 8 + li)
            for j in np.flatnonzero(nul == 0):
                assert tuple(meta[k]) == (vi, li)
                assert np.array_equal(fr[j], want[k]), (names[vi], li, j)
                k += 1
    assert k == len(want) == 644
    bp.close()


def digests_of(bp):
    bp.synthesize()
    return bp.digest(per_utterance=True)[1]


@pytest.mark.parametrize("mode", [0, 1])
def test_shared_lists_and_records_give_the_pcm_of_the_plain_batch(mode):
    """configs[2] (4096 utterances of it), configs[3]'s truncated lists and configs[4] with its voice variants: as a plain batch
    (every utterance its own frames), as 512 shared lists, as records through setIpa -- same per-utterance digests; and a sample of
    the utterances against the oracle."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads
    n, first = 4096, 3000
    plain = workloads.make("cfg2", n, first=first)
    a = eng.BatchPlayer(22050, mode=mode)
    a.setUtterances(plain["frame_start"], plain["frames"], plain["min"], plain["fade"], plain["index"], plain["isnull"], plain["seeds"])
    want = digests_of(a)
    info_plain = a.kernelInfo()
    lists, list_of, seeds = workloads.shared("cfg2", n, first=first)
    b = eng.BatchPlayer(22050, mode=mode)
    b.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
    assert b.totalSamples == a.totalSamples and b.totalFrames == a.totalFrames == len(plain["min"])
    assert np.array_equal(digests_of(b), want)
    assert b.kernelInfo() == info_plain and info_plain["tracked_utterances"] > 0      # the same groups: one sentence is quiet, the others take tracks
    c = eng.BatchPlayer(22050, mode=mode)
    c.setIpa(**workloads.cfg2_spec(n, first=first))
    assert c.totalSamples == a.totalSamples
    assert np.array_equal(digests_of(c), want)
    assert c.kernelInfo() == info_plain
    if mode == 0:
        for u in (0, 1, 517, n - 1):
            exp, _, _ = oracle.batch_synthesize(22050, plain.slice(u, 1), threads=1)
            assert np.array_equal(c.read(u), exp) and np.array_equal(b.read(u), exp)
            fr, m, f, ix, nul = c.frames(u)
            k0, k1 = plain["frame_start"][u], plain["frame_start"][u + 1]
            assert np.array_equal(fr, plain["frames"][k0:k1]) and np.array_equal(nul, plain["isnull"][k0:k1])
    # configs[3]: the lists cut to 0.5 s
    plain3 = workloads.make("cfg3", 2048, first=100)
    a.setUtterances(plain3["frame_start"], plain3["frames"], plain3["min"], plain3["fade"], plain3["index"], plain3["isnull"], plain3["seeds"])
    lists3, list_of3, seeds3 = workloads.shared("cfg3", 2048, first=100)
    b.setUtterancesShared(lists3["frame_start"], lists3["frames"], lists3["min"], lists3["fade"], list_of3, lists3["index"], lists3["isnull"], seeds3)
    assert np.array_equal(digests_of(b), digests_of(a))
    # configs[4]: defined voices, one per variant
    plain4 = workloads.make("cfg4", 4096, first=1024)
    a.setUtterances(plain4["frame_start"], plain4["frames"], plain4["min"], plain4["fade"], plain4["index"], plain4["isnull"], plain4["seeds"])
    c.setIpa(**workloads.cfg4_spec(4096, first=1024, per=1024))
    assert c.totalSamples == a.totalSamples
    assert np.array_equal(digests_of(c), digests_of(a))
    for x in (a, b, c):
        x.close()


def test_ragged_lists_with_marks_nulls_and_unspoken_lists():
    """Shared lists of the scenario corpus (NaN holds, zero-length frames, vibrato, nasals; index marks): some lists spoken many times
    with different seeds, some once, some by nobody, one empty -- every utterance equals a fresh oracle player, marks included; in every
    kernel layout."""
    import nvspeechplayer_amd as eng
    ref = scenarios.Ref()
    sel = [s for s in scenarios.build_scenarios(ref) if s.batchable and s.sr == 22050]
    frames, mins, fades, idx, nul, start = [], [], [], [], [], [0]
    for j, s in enumerate(sel):
        fr, m, f, ix, nu = s.frames()
        if j == 3:
            fr, m, f, ix, nu = fr[:0], m[:0], f[:0], ix[:0], nu[:0]      # an empty list
        frames.append(fr); mins.append(m); fades.append(f); idx.append(ix); nul.append(nu)
        start.append(start[-1] + len(m))
    lists = dict(frames=np.concatenate(frames), min=np.concatenate(mins), fade=np.concatenate(fades), index=np.concatenate(idx),
                 isnull=np.concatenate(nul), frame_start=np.array(start, np.int64))
    k = np.arange(len(lists["index"]))
    lists["index"] = np.where((lists["index"] == -1) & (k % 4 == 1), (k % 991).astype(np.int32), lists["index"]).astype(np.int32)
    rng = np.random.default_rng(5)
    nl = len(sel)
    list_of = np.concatenate([rng.integers(0, nl // 2, 300), np.arange(nl // 2, nl - 2), [3, 3]]).astype(np.uint32)      # the last two lists: nobody
    rng.shuffle(list_of)
    seeds = rng.integers(0, 2 ** 32, len(list_of), dtype=np.uint32)
    # the same utterances as a plain batch for the oracle
    fs = lists["frame_start"]
    rows = np.concatenate([np.arange(fs[l], fs[l + 1]) for l in list_of])
    plain = dict(frames=lists["frames"][rows], min=lists["min"][rows], fade=lists["fade"][rows], index=lists["index"][rows], isnull=lists["isnull"][rows],
                 frame_start=np.concatenate([[0], np.cumsum((fs[1:] - fs[:-1])[list_of])]).astype(np.int64), seeds=seeds)
    exp, exp_start, total = oracle.batch_synthesize(22050, plain, threads=4)
    exp_marks = oracle.batch_last_index(22050, plain, threads=4)
    for layout in (-1, 1, 0):
        bp = eng.BatchPlayer(22050, layout=layout)
        bp.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
        assert bp.totalSamples == total
        bp.synthesize()
        got, got_start = bp.readAll()
        assert np.array_equal(got_start, exp_start)
        d = got.astype(np.int32) - exp.astype(np.int32)
        assert np.abs(d).max() <= 1 and np.count_nonzero(d) <= 5 * len(exp) // 1000000 + 1, (layout, int(np.count_nonzero(d)))
        plainb = eng.BatchPlayer(22050, layout=layout)
        plainb.setUtterances(plain["frame_start"], plain["frames"], plain["min"], plain["fade"], plain["index"], plain["isnull"], plain["seeds"])
        plainb.synthesize()
        assert [bp.getLastIndex(u) for u in range(len(list_of))] == [plainb.getLastIndex(u) for u in range(len(list_of))]
        assert [bp.getLastIndex(u) for u in range(len(list_of))] == list(exp_marks)
        assert np.array_equal(plainb.readAll()[0], got)
        bp.close(); plainb.close()


def test_bad_compact_arguments_are_refused():
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import ipa, _native
    bp = eng.BatchPlayer(22050)
    pk = ipa.records_for_batch(["hælou", "wɜːld"], clauseType=".")
    bp.setRecords(pk["shapes"], pk["list_start"], pk["records"], pk["list_of"])
    n = bp.totalSamples
    bad = pk["records"].copy()
    bad["shape"][1] = len(pk["shapes"])
    with pytest.raises(RuntimeError, match="names shape"):
        bp.setRecords(pk["shapes"], pk["list_start"], bad, pk["list_of"])
    with pytest.raises(RuntimeError, match="not a list"):
        bp.setRecords(pk["shapes"], pk["list_start"], pk["records"], [0, 2])
    with pytest.raises(RuntimeError, match="no listOf"):
        bp._check(bp._dll.speechPlayer_batch_setRecords(bp._h, len(pk["shapes"]), pk["shapes"].ctypes.data, 2, pk["list_start"].ctypes.data,
                                                        pk["records"].ctypes.data, 3, None, None))
    assert bp.totalSamples == n      # a refused call leaves the batch that was there
    bp.synthesize()
    assert len(bp.read(1)) == bp.utteranceSamples(1) > 0
    # an empty batch, and a batch of empty lists
    bp.setRecords(pk["shapes"], [0], pk["records"][:0], None)
    assert bp.totalSamples == 0
    bp.synthesize()
    bp.setRecords(pk["shapes"], [0, 0, 0], pk["records"][:0], [1, 0, 1])
    assert bp.totalSamples == 0 and bp.nUtterances == 3
    bp.synthesize()
    assert len(bp.read(2)) == 0
    bp.close()


def test_hashed_shapes_are_verified_where_the_frames_are():
    """The planner recognises a frame by a 128-bit hash of its 45 shape values; every frame so recognised is compared with the frame
    that first carried the hash, on the device (klatt_verify_shared).  With the planner looking at 6 bits of the hash different frames
    collide for certain: the batch must come out WITHOUT tracks, with a message, successfully -- and with the PCM it always has.  With
    all 128 bits the same batch is tracked and no message is left.  Records need no hash: not affected."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads, host_array, _native
    b = workloads.make("cfg2", 1024, first=64)
    bp = eng.BatchPlayer(22050)
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    want = digests_of(bp)
    tracked = bp.kernelInfo()["tracked_utterances"]
    assert tracked >= 512 and _native.last_error_code() == 0
    L = _native.load()
    try:
        assert L.speechPlayer_setGlobalOption(b"plan_hash_bits", 6) == 0
        for pinned in (False, True):
            fr = b["frames"]
            if pinned:
                fr = host_array(b["frames"].shape, np.float64)
                fr[...] = b["frames"]
            bp.setUtterances(b["frame_start"], fr, b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
            assert "one 128-bit shape hash and different values" in _native.last_error() and _native.last_error_code() == 0
            assert bp.kernelInfo()["tracked_utterances"] == 0
            assert np.array_equal(digests_of(bp), want)
        c = eng.BatchPlayer(22050)
        c.setIpa(**workloads.cfg2_spec(1024, first=64))
        assert c.kernelInfo()["tracked_utterances"] == tracked
        assert np.array_equal(digests_of(c), want)
        c.close()
    finally:
        L.speechPlayer_setGlobalOption(b"plan_hash_bits", 128)
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    assert bp.kernelInfo()["tracked_utterances"] == tracked
    assert np.array_equal(digests_of(bp), want)
    bp.close()


def test_read_all_async_needs_a_launch():
    """speechPlayer_batch_readAllAsync hands out what a launch produced: on a batch that was set and never synthesised it refuses
     where readAll returns nothing."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads, host_array
    b = workloads.make("cfg1", 64)
    bp = eng.BatchPlayer(22050)
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    out = host_array(bp.totalSamples, np.int16)
    with pytest.raises(RuntimeError, match="not been synthesised"):
        bp.readAllAsync(out)
    bp.synthesize(wait=False)
    got, starts = bp.readAllAsync(out)
    bp.readWait()
    ref_pcm, ref_starts = bp.readAll()
    assert np.array_equal(got, ref_pcm) and np.array_equal(starts, ref_starts)
    bp.close()


def test_a_handle_pulled_alone_fills_its_wavefront_and_changes_nothing(ref):
    """A lone handle is advanced in all 64 lanes of its wavefront (64 identical control entries) on the kernel's LONE instantiation,
    whose fade stretches -- whole chunks, and the runs either side of an event inside a chunk -- are computed side by side across those
    lanes and handed out through LDS (speechPlayer_setGlobalOption("live_replicate")): the same PCM,
    call lengths and index marks as in one lane, pull by pull -- ragged pulls through speech with fades, silences, marks and a purge
    in the middle of a fade; and the same when the option changes between pulls of one handle."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import _native
    L = _native.load()
    case = ref.ipa_case(ref.find_ipa(1)) + ref.ipa_case(ref.find_ipa(5))
    pulls = [8192, 1, 17, 4096, 333, 8192, 64, 5000, 16, 15, 8192, 8192, 2048]

    def run(policy):
        p = eng.SpeechPlayer(22050, noiseSeed=9)
        out, marks = [], []
        k = 0
        for j, (fr, m, f) in enumerate(case):
            p.queueFrameSamples(None if fr is None else eng.Frame.from_array(fr), m, f, userIndex=(j if j % 3 == 0 else -1))
        for i in range(60):
            assert L.speechPlayer_setGlobalOption(b"live_replicate", {"on": 1, "off": 0, "alternate": i % 2}[policy]) == 0
            n = pulls[i % len(pulls)]
            if i == 7:      # a purge inside a fade: the frame that follows cuts over from the interpolated values
                fr, m, f = case[3]
                p.queueFrameSamples(eng.Frame.from_array(fr), m, f, userIndex=777, purgeQueue=True)
                for fr2, m2, f2 in case[4:]:
                    p.queueFrameSamples(None if fr2 is None else eng.Frame.from_array(fr2), m2, f2)
            buf = p.synthesize(n)
            got = np.zeros(0, np.int16) if buf is None else np.frombuffer(buf, dtype=np.int16)[:buf.length].copy()
            out.append(got); marks.append(p.getLastIndex())
            if len(got) < n:
                break
        p.close()
        return out, marks
    try:
        one_lane = run("off")
        for policy in ("on", "alternate"):
            got = run(policy)
            assert [len(x) for x in got[0]] == [len(x) for x in one_lane[0]] and got[1] == one_lane[1], policy
            assert all(np.array_equal(a, b) for a, b in zip(got[0], one_lane[0])), policy
        assert sum(len(x) for x in one_lane[0]) > 60000 and 777 in one_lane[1]
    finally:
        L.speechPlayer_setGlobalOption(b"live_replicate", 1)


def test_live_trim_releases_and_rebuilds_the_arena(ref):
    """speechPlayer_setGlobalOption("live_trim", 1): when the last live handle of a device is terminated its arena (state blocks, rings,
    pull buffers) is released; the next handle starts a new one and synthesises what a handle always does."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import _native
    L = _native.load()
    case = ref.ipa_case(ref.find_ipa(0))

    def speak():
        p = eng.SpeechPlayer(22050, noiseSeed=4)
        for fr, m, f in case:
            p.queueFrameSamples(None if fr is None else eng.Frame.from_array(fr), m, f)
        parts = []
        while True:
            buf = p.synthesize(4096)
            if buf is None:
                break
            parts.append(np.frombuffer(buf, dtype=np.int16)[:buf.length].copy())
            if buf.length < 4096:
                break
        return p, np.concatenate(parts)
    try:
        a, want = speak()
        b, same = speak()
        assert np.array_equal(want, same)
        assert L.speechPlayer_setGlobalOption(b"live_trim", 1) == 0      # handles live: nothing is released
        c, again = speak()
        assert np.array_equal(want, again)
        for p in (a, b, c):
            p.close()                                                    # the last one trims the arena
        for _ in range(2):
            d, fresh = speak()                                           # a new arena
            assert np.array_equal(want, fresh)
            d.close()
    finally:
        L.speechPlayer_setGlobalOption(b"live_trim", 0)


@pytest.mark.parametrize("options", [dict(tracks=0, direct=2), dict(tracks=0, direct=0), dict(tracks=1, direct=1, track_budget_mb=1)])
def test_shared_lists_on_every_noisy_path(options):
    """Shared frame lists on the direct stages (their per-frame seeds and headers are shared like the frames: one record run per LIST),
    on the stages with the frame state machine, and with a track budget that runs out: the PCM of the plain batch, utterance by
    utterance, and the oracle's on a sample; marks included."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads
    n, first = 1536, 200
    plain = workloads.jittered(workloads.make("cfg2", n, first=first), seed=3)
    k = np.arange(len(plain["index"]))
    plain["index"] = np.where(k % 7 == 3, (k % 9973).astype(np.int32), -1).astype(np.int32)
    # 96 jittered lists, each spoken by 16 utterances with seeds of their own
    lists = plain.slice(0, 96)
    list_of = (np.arange(n) % 96).astype(np.uint32)
    seeds = (np.arange(n) * 7 + 1).astype(np.uint32)
    fs = lists["frame_start"]
    rows = np.concatenate([np.arange(fs[l], fs[l + 1]) for l in list_of])
    same = workloads.Batch(frames=lists["frames"][rows], min=lists["min"][rows], fade=lists["fade"][rows], index=lists["index"][rows], isnull=lists["isnull"][rows],
                           frame_start=np.concatenate([[0], np.cumsum((fs[1:] - fs[:-1])[list_of])]).astype(np.int64), seeds=seeds, name="same", sr=22050)
    a = eng.BatchPlayer(22050); b = eng.BatchPlayer(22050)
    for name, value in options.items():
        a.setOption(name, value); b.setOption(name, value)
    a.setUtterances(same["frame_start"], same["frames"], same["min"], same["fade"], same["index"], same["isnull"], same["seeds"])
    b.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
    ia, ib = a.kernelInfo(), b.kernelInfo()
    # (the budget of 1 MB holds no jittered batch's tracks: all or nothing, so that case runs on the direct stages too -- whose seeds are per LIST)
    assert ia["direct_utterances"] == ib["direct_utterances"] and (ia["direct_utterances"] > 0) == (options["direct"] != 0), (ia, ib)
    assert ia["tracked_utterances"] == ib["tracked_utterances"] == 0
    if options["direct"]:
        assert ib["direct_mbytes"] * 8 < ia["direct_mbytes"]
    a.synthesize(); b.synthesize()
    da, db = a.digest(per_utterance=True)[1], b.digest(per_utterance=True)[1]
    assert np.array_equal(da, db)
    assert [a.getLastIndex(u) for u in range(0, n, 37)] == [b.getLastIndex(u) for u in range(0, n, 37)]
    for u in (0, 95, 96, n - 1):
        exp, _, _ = oracle.batch_synthesize(22050, same.slice(u, 1), threads=1)
        got = b.read(u)
        d = got.astype(np.int32) - exp.astype(np.int32)
        assert len(got) == len(exp) and np.abs(d).max() <= 1 and np.count_nonzero(d) <= 1, u
        assert b.getLastIndex(u) == int(oracle.batch_last_index(22050, same.slice(u, 1))[0])
    a.close(); b.close()


def test_a_few_handles_pulled_together_fill_their_wavefront(ref):
    """A few live handles in a pull: every one in a wavefront of its own (the default: option "live_alone"); or, sharing one wavefront
    ("live_alone" 1), its empty lanes advance replicas of them (fewer than 32 handles; option "live_replicate").  Five handles with
    different sentences, seeds and queue lengths, pulled together in ragged pulls until the last one has drained: the same samples,
    counts and marks as in one lane each of one wavefront."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import _native
    L = _native.load()
    pulls = [4096, 100, 8192, 1, 3000, 8192]

    def run(replicate, alone=1):
        assert L.speechPlayer_setGlobalOption(b"live_replicate", replicate) == 0 and L.speechPlayer_setGlobalOption(b"live_alone", alone) == 0
        players = []
        for j in range(5):
            p = eng.SpeechPlayer(22050, noiseSeed=100 + j)
            for k, (fr, m, f) in enumerate(ref.ipa_case(ref.find_ipa(j))):
                p.queueFrameSamples(None if fr is None else eng.Frame.from_array(fr), m, f, userIndex=(k if k % 4 == 1 else -1))
            players.append(p)
        got = [[] for _ in players]
        marks = []
        for i in range(40):
            n = pulls[i % len(pulls)]
            out = np.zeros((5, n), np.int16)
            produced = eng.SpeechPlayer.synthesizeMany(players, n, out=out)
            for j in range(5):
                got[j].append(out[j, :produced[j]].copy())
            marks.append([p.getLastIndex() for p in players])
            if not produced.any():
                break
        for p in players:
            p.close()
        return [np.concatenate(g) for g in got], marks
    try:
        off, marks_off = run(0)
        for on, marks_on in (run(1), run(1, alone=1536)):
            assert marks_on == marks_off and [len(x) for x in on] == [len(x) for x in off] and min(len(x) for x in off) > 8000
            assert all(np.array_equal(a, b) for a, b in zip(on, off))
    finally:
        L.speechPlayer_setGlobalOption(b"live_replicate", 1)
        L.speechPlayer_setGlobalOption(b"live_alone", 1536)


@pytest.mark.parametrize("deal", [0, 1])
def test_node_batch_from_ipa_text(deal):
    """speechPlayer_node_setIpa: the compact form over the devices of a node (three shards on this box's one GPU), under the contiguous
    and the sorted deal: every utterance's digest, a sample of PCM and lengths equal the single-device batch set from the same text;
    configs[4]'s recipe with a voice per utterance."""
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads
    from nvspeechplayer_amd.sharding import shard_bounds, shard_deal
    n = 3000
    for spec in (workloads.cfg2_spec(n, first=77), workloads.cfg4_spec(n, first=512, per=1024)):
        one = eng.BatchPlayer(22050)
        one.setIpa(**spec)
        one.synthesize()
        want = one.digest(per_utterance=True)[1]
        lens = np.array([one.utteranceSamples(u) for u in range(n)])
        node = eng.NodePlayer(22050, [0, 0, 0])
        node.setOption("deal", deal)
        node.setIpa(**spec)
        assert node.totalSamples == one.totalSamples
        parts = shard_deal(lens, 3, "sorted") if deal else [np.arange(a, b) for a, b in zip(shard_bounds(lens, 3)[:-1], shard_bounds(lens, 3)[1:])]
        for d in range(3):
            assert np.array_equal(node.shardUtterances(d), parts[d])
        node.synthesize()
        assert np.array_equal(node.digests(), want)
        for u in (0, 1, n // 2, n - 1):
            assert np.array_equal(node.read(u), one.read(u))
        node.close(); one.close()

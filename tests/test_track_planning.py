"""Host logic of the tracks (speechPlayer_planTracks: what speechPlayer_batch_setUtterances plans; no GPU needed):
which resonators a fade moves, which fades share a track, where the tracks lie, and the all-or-nothing budget rule -- against a
plain Python walk of the reference's frame rules (src/frame.cpp:55-72: silence keeps the last spoken shape, the first frame
after silence starts from its own shape, any other frame fades from the last spoken frame's values)."""
import ctypes

import numpy as np

from nvspeechplayer_amd import _native

RES_F = [13, 14, 12, 11, 10, 9, 8, 7, 25, 26, 27, 28, 29, 30]      # N0, NP, c6..c1, p1..p6 (frame.h:24-42 order)
RES_B = [21, 22, 20, 19, 18, 17, 16, 15, 31, 32, 33, 34, 35, 36]
# entry kinds 14..23: pairs of parameters (-1: none); 44 = preFormantGain, which silence gates off (frame.cpp:61,66)
PAIRS = [(23, -1), (41, 42), (43, 45), (24, 44), (37, 38), (39, 40), (1, 2), (3, 4), (5, 6), (44, -1)]
SHAPE = [p for r in range(14) for p in (RES_F[r], RES_B[r])] + [23, 41, 42, 43, 45, 24, 44, 37, 38, 39, 40, 1, 2, 3, 4, 5, 6]
FIRST = 25                                                          # entries of a fade's first sample: 24 kinds, N0 takes two


def plan(frame_start, frames, fade, isnull, eligible=None, budget_mb=16384):
    L = _native.load()
    nu, nf = len(frame_start) - 1, len(fade)
    off = np.zeros(nf, np.uint64); mask = np.zeros(nf, np.uint32); tracked = np.zeros(nu, np.uint8)
    entries = ctypes.c_ulonglong(0)
    frames = np.ascontiguousarray(frames, np.float64); fade = np.ascontiguousarray(fade, np.uint32)
    isnull = np.ascontiguousarray(isnull, np.uint8); frame_start = np.ascontiguousarray(frame_start, np.int64)
    el = None if eligible is None else np.ascontiguousarray(eligible, np.uint8)
    n = L.speechPlayer_planTracks(nu, frame_start.ctypes.data, frames.ctypes.data, fade.ctypes.data, isnull.ctypes.data,
                                  None if el is None else el.ctypes.data, budget_mb, off.ctypes.data, mask.ctypes.data,
                                  tracked.ctypes.data, ctypes.byref(entries))
    return n, int(entries.value), off, mask, tracked


def expected(frame_start, frames, fade, isnull):
    """(mask, key) per frame; key = (values at the fade's start, at its end, fade length).  The end points follow
    reference src/frame.cpp:55-72: silence = the previous request's values with the gain gated off; the first frame after
    silence (or the first of all) starts from its own values with the gain gated off; else from the previous request's values."""
    masks, keys = [], []
    for u in range(len(frame_start) - 1):
        prev, prev_null = {p: 0.0 for p in range(47)}, True
        for k in range(frame_start[u], frame_start[u + 1]):
            F = max(int(fade[k]), 1)
            if isnull[k]:
                old, new = dict(prev), dict(prev)
                new[44] = 0.0
                prev_null = True
            else:
                new = {p: float(frames[k][p]) for p in range(47)}
                old = dict(new) if prev_null else dict(prev)
                if prev_null:
                    old[44] = 0.0
                prev_null = False
            prev = new
            m = 0
            for r in range(14):
                if old[RES_F[r]] != new[RES_F[r]] or old[RES_B[r]] != new[RES_B[r]]:
                    m |= 1 << r
            for e, (a, b) in enumerate(PAIRS):
                if old[a] != new[a] or (b >= 0 and old[b] != new[b]):
                    m |= 1 << (14 + e)
            masks.append(m)
            keys.append((np.array([old[p] for p in SHAPE]).tobytes(), np.array([new[p] for p in SHAPE]).tobytes(), F))
    return masks, keys


def random_frames(rng, n_utt):
    """few distinct shapes, so that fades repeat across utterances; NULL frames anywhere; -0.0 against 0.0"""
    shapes = rng.uniform(100, 5000, size=(6, 47))
    shapes[1] = shapes[0].copy(); shapes[1][9] += 1.0     # differs from shape 0 in one formant only
    shapes[2][13] = 0.0; shapes[3] = shapes[2].copy(); shapes[3][13] = -0.0   # equal by value, different bits
    shapes[4] = shapes[0].copy(); shapes[4][44] += 0.25   # differs from shape 0 in the gain only
    shapes[5] = shapes[0].copy(); shapes[5][5] *= 0.5     # ... in a source parameter only
    frames, fade, nul, start = [], [], [], [0]
    for _ in range(n_utt):
        n = int(rng.integers(1, 7))
        for _ in range(n):
            frames.append(shapes[rng.integers(0, 6)].copy()); fade.append(int(rng.choice([0, 1, 2, 50, 300])))
            nul.append(rng.random() < 0.25)
        start.append(start[-1] + n)
    return np.array(start, np.int64), np.array(frames), np.array(fade, np.uint32), np.array(nul, np.uint8)


def test_masks_sharing_and_layout():
    rng = np.random.default_rng(5)
    fs, frames, fade, nul = random_frames(rng, 300)
    n_tracks, entries, off, mask, tracked = plan(fs, frames, fade, nul)
    assert tracked.all()
    masks, keys = expected(fs, frames, fade, nul)
    assert [int(m) for m in mask] == masks
    # equal fades share a track, different fades do not; tracks tile [0, entries) without overlap, sized 15 + (F - 1) * slots
    by_key, spans = {}, {}
    for k, key in enumerate(keys):
        assert by_key.setdefault(key, int(off[k])) == int(off[k])
        slots = bin(masks[k]).count("1") + (masks[k] & 1)
        spans[int(off[k])] = FIRST + (key[2] - 1) * slots
    assert len(by_key) == n_tracks == len(set(by_key.values()))
    pos = 0
    for o in sorted(spans):
        assert o == pos
        pos += spans[o]
    assert pos == entries
    # the anti-resonator takes two entries per sample
    k = next(k for k, m in enumerate(masks) if m & 1 and fade[k] > 1)
    assert spans[int(off[k])] == FIRST + (max(int(fade[k]), 1) - 1) * (bin(masks[k]).count("1") + 1)


def test_eligibility_and_budget():
    rng = np.random.default_rng(6)
    fs, frames, fade, nul = random_frames(rng, 200)
    el = (rng.random(200) < 0.7).astype(np.uint8)
    n_tracks, entries, off, mask, tracked = plan(fs, frames, fade, nul, eligible=el)
    assert np.array_equal(tracked, el)
    for u in range(200):
        if not el[u]:
            assert not off[fs[u]:fs[u + 1]].any() and not mask[fs[u]:fs[u + 1]].any()
    # a budget that cannot hold the batch: nothing is tracked (a split batch measured slower than either kernel alone)
    n0, e0, off0, mask0, tracked0 = plan(fs, frames, fade, nul, budget_mb=0)
    assert n0 == 0 and e0 == 0 and not tracked0.any() and not off0.any()
    # a budget that holds everything but a few utterances keeps the rest tracked: make one utterance enormous
    fade2 = fade.copy(); frames2 = frames.copy()
    k = int(fs[7])
    frames2[k + 0] = rng.uniform(100, 5000, 47)
    if fs[8] - fs[7] > 1:
        frames2[k + 1] = rng.uniform(100, 5000, 47); nul2 = nul.copy(); nul2[k] = 0; nul2[k + 1] = 0; fade2[k + 1] = 50_000_000
        n2, e2, off2, mask2, tracked2 = plan(fs, frames2, fade2, nul2, budget_mb=64)
        assert not tracked2[7] and tracked2.sum() == 199


def test_planning_in_parts_gives_the_plan_of_one_pass(monkeypatch):
    """Large batches are planned by several host threads and merged (plan_tracks): same tracks, same places, as one pass --
    with the budget met, with it met by the merged tracks only, with it missed (the one pass decides) and with untrackable
    utterances among them (the tracks of this batch take 2.5 MB)."""
    rng = np.random.default_rng(11)
    start, frames, fade, nul = random_frames(rng, 70000)          # > 200 000 frames: the parts are used
    assert len(fade) > 200000
    el = (rng.random(len(start) - 1) < 0.9).astype(np.uint8)
    for budget, eligible in ((16384, None), (16384, el), (20, None), (3, el), (2, None), (0, None)):
        monkeypatch.setenv("SPEECHPLAYER_PLAN_THREADS", "1")
        one = plan(start, frames, fade, nul, eligible, budget)
        for threads in ("2", "5", "8"):
            monkeypatch.setenv("SPEECHPLAYER_PLAN_THREADS", threads)
            many = plan(start, frames, fade, nul, eligible, budget)
            assert many[:2] == one[:2]
            for a, b in zip(many[2:], one[2:]):
                assert np.array_equal(a, b)
    assert one[0] == 0 and not one[4].any()                       # no budget: nothing tracked
    # a fade too long for a track (2^27 entries) keeps its utterance out, in a part as in one pass; more than a tenth of them, everything
    two = np.diff(start) >= 2
    for share in (0.05, 0.9):
        fade2 = fade.copy()
        hit = two & (rng.random(len(start) - 1) < share)
        fade2[start[:-1][hit] + 1] = 1 << 23                      # on the second frame (the first only moves the gain: from silence)
        monkeypatch.setenv("SPEECHPLAYER_PLAN_THREADS", "1")
        one = plan(start, frames, fade2, nul)
        monkeypatch.setenv("SPEECHPLAYER_PLAN_THREADS", "8")
        many = plan(start, frames, fade2, nul)
        assert many[:2] == one[:2] and all(np.array_equal(a, b) for a, b in zip(many[2:], one[2:]))
        if share < 0.1:
            assert one[4][~hit].all() and not one[4][hit].all()  # (a hit utterance stays in when its long fade moves few kinds)
        else:
            assert not one[4].any()


def test_probe_of_a_short_batch_of_long_utterances(monkeypatch):
    """the look at the first utterances (plan_tracks) read past the batch when the threaded path was entered with
    fewer than 256 utterances (130 utterances x 1600 frames, two plan threads).  The arrays are placed so that the bytes behind
    frameStart are a PROT_NONE page: an overrun is a segfault, not a silent read."""
    import mmap
    rng = np.random.default_rng(3)
    n_utt, per = 130, 1600
    shapes = rng.uniform(100, 5000, size=(5, 47))
    frames = shapes[rng.integers(0, 5, n_utt * per)]
    fade = rng.choice([1, 2, 50, 300], n_utt * per).astype(np.uint32)
    nul = (rng.random(n_utt * per) < 0.1).astype(np.uint8)
    page = mmap.PAGESIZE
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    nbytes = (n_utt + 1) * 8
    span = (nbytes + page - 1) // page * page
    base = libc.mmap(None, span + page, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base not in (None, ctypes.c_void_p(-1).value)
    assert libc.mprotect(base + span, page, 0) == 0      # PROT_NONE
    fs = np.frombuffer((ctypes.c_char * nbytes).from_address(base + span - nbytes), dtype=np.int64)
    fs[:] = np.arange(n_utt + 1, dtype=np.int64) * per
    L = _native.load()
    off = np.zeros(n_utt * per, np.uint64); mask = np.zeros(n_utt * per, np.uint32); tracked = np.zeros(n_utt, np.uint8)
    entries = ctypes.c_ulonglong(0)
    out = {}
    for threads in ("1", "2"):
        monkeypatch.setenv("SPEECHPLAYER_PLAN_THREADS", threads)
        n = L.speechPlayer_planTracks(n_utt, fs.ctypes.data, np.ascontiguousarray(frames).ctypes.data, fade.ctypes.data, nul.ctypes.data, None, 16384,
                                      off.ctypes.data, mask.ctypes.data, tracked.ctypes.data, ctypes.byref(entries))
        out[threads] = (n, int(entries.value), off.copy(), mask.copy(), tracked.copy())
    assert out["1"][:2] == out["2"][:2] and all(np.array_equal(a, b) for a, b in zip(out["1"][2:], out["2"][2:]))
    assert out["1"][4].all()


FACTS = np.dtype([("h0", "<u8"), ("h1", "<u8"), ("flags", "<u4"), ("pad", "<u4")])


def frame_facts(frames, sr=22050, on_device=0):
    L = _native.load()
    frames = np.ascontiguousarray(frames, np.float64).reshape(-1, 47)
    out = np.zeros(len(frames), FACTS)
    n = L.speechPlayer_frameFacts(frames.ctypes.data, len(frames), sr, on_device, out.ctypes.data)
    assert n == len(frames), _native.last_error()
    return out


def facts_frames(rng, n):
    """Frames that exercise every flag: plain speech-like ones, noise gains, NaN / inf anywhere, negative bandwidths, coupled nasal
    pairs, out-of-range formants, negative zeros."""
    f = np.zeros((n, 47))
    f[:, 0] = rng.uniform(60, 300, n); f[:, 46] = f[:, 0] * rng.uniform(0.8, 1.2, n); f[:, 5] = rng.uniform(0, 1, n)
    f[:, 7:13] = np.sort(rng.uniform(200, 5000, (n, 6)), axis=1); f[:, 13] = rng.uniform(200, 500, n); f[:, 14] = rng.uniform(200, 500, n)
    f[:, 15:23] = rng.uniform(40, 900, (n, 8)); f[:, 25:31] = np.sort(rng.uniform(200, 5000, (n, 6)), axis=1); f[:, 31:37] = rng.uniform(40, 900, (n, 6))
    f[:, 37:43] = rng.uniform(0, 1, (n, 6)); f[:, 43] = rng.uniform(0, 1, n); f[:, 44] = 1.0; f[:, 45] = 2.0
    pick = lambda p: rng.random(n) < p
    for col in (3, 6, 24):
        f[pick(0.2), col] = rng.uniform(0.01, 1)
    f[pick(0.1), 23] = 0.5
    f[pick(0.03), rng.integers(0, 47)] = np.nan
    f[pick(0.02), rng.integers(0, 47)] = np.inf
    f[pick(0.03), 31 + rng.integers(0, 6)] = -5.0
    f[pick(0.03), 7 + rng.integers(0, 6)] = 3e8
    f[pick(0.03), 21] = 0.5
    f[pick(0.05), 4] = -0.0
    return f


def expected_flags(f, sr=22050):
    max_bw, max_f = 690.0 * sr / np.pi, 9900.0 * sr / (2 * np.pi)
    with np.errstate(invalid="ignore"):
        noise = (f[:, 3] != 0) | (f[:, 6] != 0) | (f[:, 24] != 0)
        noise |= (~(np.abs(f[:, 25:31]) <= 1e6) | ~(f[:, 31:37] >= 0) | ~(f[:, 31:37] <= 1e6)).any(axis=1)
        nonfinite = ~np.isfinite(f).all(axis=1)
        nasal = (f[:, 23] != 0) | ~(f[:, 21] >= 1) | ~(f[:, 22] >= 0) | ~(f[:, 21] <= 1e6) | ~(f[:, 22] <= 1e6) | ~(np.abs(f[:, 13]) <= 1e6) | \
                ~(np.abs(f[:, 14]) <= 1e6) | ~(np.abs(f[:, 5]) <= 1e30) | ~(np.abs(f[:, 44]) <= 1e30) | ~(np.abs(f[:, 0]) <= 1e30) | ~(np.abs(f[:, 46]) <= 1e30)
        freq = np.concatenate([f[:, 7:15], f[:, 25:31]], axis=1); bw = np.concatenate([f[:, 15:23], f[:, 31:37]], axis=1)
        unbounded = (~(np.abs(freq) <= max_f)).any(axis=1) | (~(np.abs(bw) <= max_bw)).any(axis=1)
    return noise * 1 + nonfinite * 2 + nasal * 4 + unbounded * 8


def test_frame_facts_flags_and_hash():
    """klatt_plan.h on the host (speechPlayer_frameFacts, onDevice = 0): the flags against a numpy restatement of the rules
    speechPlayer_batch_setUtterances classifies by; the 128-bit hash stands for the 45 shape values -- equal for frames that differ
    in their pitches only, different for a change of any one other parameter (a sign of zero included), and for the same values in
    other places."""
    rng = np.random.default_rng(9)
    f = facts_frames(rng, 6000)
    got = frame_facts(f)
    assert np.array_equal(got["flags"], expected_flags(f).astype(np.uint32))
    assert set(np.unique(got["flags"])) >= {0, 1, 2, 4, 8}                     # every flag occurs, and none at all
    g = f.copy(); g[:, 0] *= 1.5; g[:, 46] += 3.0                               # the pitches are not part of a shape
    other = frame_facts(g)
    assert np.array_equal(other["h0"], got["h0"]) and np.array_equal(other["h1"], got["h1"])
    base = f[:200].copy()
    h = frame_facts(base)
    for p in range(1, 46):
        g = base.copy(); g[:, p] = np.where(np.isfinite(g[:, p]), g[:, p] + 1.0, 7.0)
        c = frame_facts(g)
        assert not (c["h0"] == h["h0"]).any() and not (c["h1"] == h["h1"]).any(), p
    g = base.copy(); g[:, [7, 8]] = g[:, [8, 7]]                               # two formants exchanged: other places, other hash
    c = frame_facts(g)
    assert not ((c["h0"] == h["h0"]) & (c["h1"] == h["h1"])).any()
    z = np.zeros((2, 47)); z[1, 4] = -0.0
    c = frame_facts(z)
    assert c["h0"][0] != c["h0"][1]                                              # bit patterns, as the planner's memcmp compared them
    pairs = np.stack([got["h0"], got["h1"]], axis=1)
    shape_rows = np.ascontiguousarray(f[:, 1:46]).view(np.uint64)
    assert len(np.unique(pairs, axis=0)) == len(np.unique(shape_rows, axis=0))  # as many distinct hashes as distinct shapes

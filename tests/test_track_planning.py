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

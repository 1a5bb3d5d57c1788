"""Host logic of the direct stages (speechPlayer_planDirect: the fade end points speechPlayer_batch_setUtterances hands klatt_seeds;
no GPU needed) against a plain Python walk of the reference's dequeue rules, written the way the reference states them
(src/frame.cpp:55-72): the frame manager holds the previous REQUEST; a NULL request becomes a copy of the previous request's frame
with preFormantGain = 0 (:59-63); a real request after a NULL one (or after nothing) makes the OLD side a copy of the new frame with
preFormantGain = 0 (:64-67); otherwise the old side is the previous request's frame as it stands."""
import numpy as np

from nvspeechplayer_amd import _native

NONE = 0xFFFFFFFF


def plan(frame_start, isnull):
    L = _native.load()
    nf = int(frame_start[-1])
    frm = np.zeros(nf, np.uint32); to = np.zeros(nf, np.uint32); flags = np.zeros(nf, np.uint32)
    fs = np.ascontiguousarray(frame_start, np.int64); nul = np.ascontiguousarray(isnull, np.uint8)
    n = L.speechPlayer_planDirect(len(fs) - 1, fs.ctypes.data, nul.ctypes.data, frm.ctypes.data, to.ctypes.data, flags.ctypes.data)
    assert n == nf, _native.last_error()
    return frm, to, flags


def reference_walk(frame_start, isnull):
    """Per frame (source frame of the fade's start, gated?, source frame of its end, gated?).  A request's frame is described by the
    frame its values come from (None: the zeroed frame of a fresh handle) and whether its preFormantGain has been set to 0."""
    out = []
    for u in range(len(frame_start) - 1):
        old_src, old_gated, old_is_null = None, False, True      # a fresh frame manager: no request yet (oldFrameRequest->NULLFrame = true)
        for k in range(frame_start[u], frame_start[u + 1]):
            if isnull[k]:
                new_src, new_gated = old_src, True               # :59-63 the old frame with preFormantGain = 0
                from_src, from_gated = old_src, old_gated
            else:
                new_src, new_gated = k, False
                if old_is_null:
                    from_src, from_gated = k, True               # :64-67 the old side becomes the new frame, gain 0
                else:
                    from_src, from_gated = old_src, old_gated
            out.append((from_src, from_gated, new_src, new_gated))
            old_src, old_gated, old_is_null = new_src, new_gated, bool(isnull[k])
    return out


def check(frame_start, isnull):
    frm, to, flags = plan(frame_start, isnull)
    exp = reference_walk(frame_start, isnull)
    for k, (fs, fg, ts, tg) in enumerate(exp):
        assert frm[k] == (NONE if fs is None else fs), k
        assert to[k] == (NONE if ts is None else ts), k
        # a gate on the zeroed frame changes nothing: the engine sets the bit there too (the values are zero either way)
        if fs is not None:
            assert bool(flags[k] & 1) == fg, (k, flags[k], fg)
        assert bool(flags[k] & 2) == tg, (k, flags[k], tg)


def test_fade_ends_of_plain_speech():
    #            u0: a b c _   u1: _ a _ _ b   u2: a
    isnull = [0, 0, 0, 1,      1, 0, 1, 1, 0,  0]
    check([0, 4, 9, 10], isnull)
    frm, to, flags = plan([0, 4, 9, 10], isnull)
    assert list(frm[:4]) == [0, 0, 1, 2] and list(to[:4]) == [0, 1, 2, 2] and list(flags[:4]) == [1, 0, 0, 2]
    assert frm[4] == NONE and to[4] == NONE                      # silence first: nothing to hold
    assert list(frm[5:9]) == [5, 5, 5, 8] and list(to[5:9]) == [5, 5, 5, 8] and list(flags[5:9]) == [1, 2, 3, 1]


def test_fade_ends_of_random_frame_lists():
    rng = np.random.default_rng(7)
    for _ in range(50):
        n_utt = int(rng.integers(1, 40))
        counts = rng.integers(0, 30, n_utt)
        frame_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        isnull = (rng.random(int(frame_start[-1])) < rng.uniform(0, 0.7)).astype(np.uint8)
        check(frame_start, isnull)


def test_bad_arguments_are_refused():
    L = _native.load()
    fs = np.array([0, 3, 2], np.int64)
    assert L.speechPlayer_planDirect(2, fs.ctypes.data, None, None, None, None) == -1
    assert "monotone" in _native.last_error()

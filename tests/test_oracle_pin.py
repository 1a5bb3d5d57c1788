"""Pin the CPU oracle to the reference.

The reference repository holds no automated tests or golden vectors for this path
(SURVEY.md section 4).  What pins the oracle are the known answers SURVEY.md
section 8(c) recorded from the compiled reference engine (glibc libm, glibc
rand() after srand(1)): the cfg0 PCM (SHA-1, first samples, extrema, call
lengths) and, for the eight sampleIpa.txt lines synthesised in one process in
order, the sample counts and SHA-1 prefixes.  The frame streams fed here come
from the reference's own frame producer (tests/golden/ref_frames.npz).
"""
import ctypes
import hashlib

import numpy as np
import pytest

from tests import oracle, scenarios

# SURVEY.md section 8(c), "Known-answer values captured"
CFG0_FIRST = [0, 0, 0, 0, 0, -2, -4, -8, -12, -17, -21, -25, -29, -32, -35, -37, -39, -40, -41, -41]
CFG0_SHA1 = "3372ce96a8e60706afbfd092c3b79e7e7c43355e"
CFG0_MINMAX = (-8857, 3350)
IPA_COUNTS = [8273, 29374, 29218, 20745, 12907, 41238, 13459, 29788]
IPA_SHA1_10 = ["657b7798a6", "f0c09fd146", "6dedd9b09d", "7ea34a8661", "183e33f142", "2616aefc56", "54e57a2c66",
               "56c09daa01"]


@pytest.fixture(scope="module")
def ref():
    return scenarios.Ref()


def test_cfg0_known_answer(ref):
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    fa = scenarios.vowel_frame(ref, "a", 120.0)
    p = oracle.OraclePlayer(22050, noise=oracle.NOISE_LIBC)
    p.queue(fa, scenarios.ms(1000), scenarios.ms(50))
    first = p.synthesize(22050)
    second = p.synthesize(22050)
    assert len(first) == 22050 and len(second) == 1
    assert first[:20].tolist() == CFG0_FIRST
    assert (int(first.min()), int(first.max())) == CFG0_MINMAX
    assert hashlib.sha1(first.tobytes()).hexdigest() == CFG0_SHA1


def test_sampleipa_known_answers_glibc_rand(ref):
    """All eight lines in ONE process, in order, srand(1) once, no reseeding."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    for line in range(8):
        case = ref.find_ipa(line, speed=1.0, clause=0, pitch=100.0, infl=0.5)
        p = oracle.OraclePlayer(22050, noise=oracle.NOISE_LIBC)
        for fr, m, f in ref.ipa_case(case):
            p.queue(fr, m, f)
        pcm = p.drain()
        assert len(pcm) == IPA_COUNTS[line]
        assert hashlib.sha1(pcm.tobytes()).hexdigest()[:10] == IPA_SHA1_10[line], "line %d" % line


def test_cfg0_is_noise_independent(ref):
    """Pure vowels have zero noise gains, so the counter noise stream gives the same PCM."""
    fa = scenarios.vowel_frame(ref, "a", 120.0)
    p = oracle.OraclePlayer(22050, noise=oracle.NOISE_COUNTER, seed=12345)
    p.queue(fa, scenarios.ms(1000), scenarios.ms(50))
    assert hashlib.sha1(p.synthesize(22050).tobytes()).hexdigest() == CFG0_SHA1

"""The optional text front-end (include/speechPlayer_batch.h, speechPlayer_text_*): what can be checked without eSpeak NG --
the clause splitting against the reference driver's regular expression and rules (nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:84,
:189-205, restated here), the replacements of :214-218, and the error a caller gets when the library is not installed.  The phonemes
themselves are eSpeak's: PARITY UNPINNED (the library is in neither the reference tree nor this image)."""
import re

import pytest

from nvspeechplayer_amd import _native, ipa

RE_TEXT_PAUSE = re.compile(r"(?<=[.?!,:;])\s", re.DOTALL | re.UNICODE)      # reference __init__.py:84


def driver_clauses(text):
    out = []
    for chunk in RE_TEXT_PAUSE.split(text):                                  # :189-205
        if not chunk:
            continue
        chunk = chunk.strip()
        if not chunk:
            continue
        c = chunk[-1]
        if c in (".", "!"):
            pause = 150.0
        elif c == "?":
            pause = 150.0
        elif c == ",":
            pause = 120.0
        else:
            pause, c = 100.0, None
        out.append((chunk, c, pause))
    return out


TEXTS = [
    "Hello, world. How are you?  Fine: thanks;\tbye !  ",
    "No punctuation at all",
    "",
    "   ",
    "One.Two. Three.\nFour!\r\nFive?Six ,seven , eight",
    "Trailing comma,",
    "a. b. c. d.",
    "Ellipsis... and more...   end.",
    "Non-breaking space after full stop. Next, thin space. Last",
    "Ideographic　space;　after semicolon",
    "Café, naïve? Été!",
    "x , y",
    ". . .",
    "comma,,double.  space",
]


def test_clause_split_follows_the_driver():
    for t in TEXTS:
        assert ipa.splitClauses(t) == driver_clauses(t), repr(t)


def test_fixups_are_the_drivers_replacements():
    def driver(chunk):                                                       # :214-218
        chunk = chunk.replace('ə͡l', 'ʊ͡l')
        chunk = chunk.replace('a͡ɪ', 'ɑ͡ɪ')
        chunk = chunk.replace('e͡ɪ', 'e͡i')
        chunk = chunk.replace('ə͡ʊ', 'o͡u')
        return chunk.strip()
    for s in ["", "  ", "həlˈoʊ", " tˈe͡ɪbə͡l ", "ə͡ʊ ə͡l a͡ɪ e͡ɪ",
              "a͡ɪa͡ɪ\n", "ə͡", "mˈa͡ɪ nˈe͡ɪm ɪz\t"]:
        assert ipa.fixups(s) == driver(s), repr(s)


def test_missing_espeak_is_a_clear_error():
    if ipa.textAvailable():
        pytest.skip("eSpeak NG is installed here")
    assert _native.last_error_code() == 4                                    # SPEECHPLAYER_ERR_TEXT_FRONTEND
    assert "libespeak-ng" in _native.last_error() and "speechPlayer_batch_setIpa" in _native.last_error()
    with pytest.raises(RuntimeError) as e:
        ipa.textToIpa("hello world.")
    assert "eSpeak" in str(e.value)
    assert _native.last_error_code() == 4
    # the IPA path is untouched by it
    assert len(list(ipa.generateFramesAndTiming("həlˈoʊ", clauseType="."))) > 0

"""CPU-side checks: the C-ABI library builds, loads and exports what the headers declare; the
arithmetic helpers the kernel relies on are exact; workload recipes and sharding are consistent."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(speechPlayer_[A-Za-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nvspeechplayer_amd import _native
    _native.build()
    lib = _native.load()
    names = declared_functions("speechPlayer.h") + declared_functions("speechPlayer_batch.h")
    assert len(names) >= 34
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) <= set(_native.EXPORTS)


def test_frame_struct_is_47_doubles():
    import ctypes
    from nvspeechplayer_amd import Frame
    assert ctypes.sizeof(Frame) == 376
    f = Frame.from_array(np.arange(47.0))
    assert f.voicePitch == 0.0 and f.endVoicePitch == 46.0 and f.cfN0 == 13.0 and f.preFormantGain == 44.0
    assert np.array_equal(f.as_array(), np.arange(47.0))


def test_no_gpu_means_loud_failure_not_fallback():
    """In this container there is no GPU: the product must refuse, not synthesise on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nvspeechplayer_amd as eng
    with pytest.raises(RuntimeError, match="no HIP device|failed"):
        eng.SpeechPlayer(22050)
    with pytest.raises(RuntimeError):
        eng.BatchPlayer(22050)


def test_failures_carry_an_error_code():
    """speechPlayer_synthesize returns 0 for "queue drained" and for a failed call alike (the reference has no error
    convention); speechPlayer_lastErrorCode tells them apart.  Without a GPU every call that needs one fails with
    SPEECHPLAYER_ERR_NO_DEVICE; calls on invalid handles fail with SPEECHPLAYER_ERR_ARGUMENT; host-only calls succeed."""
    import ctypes
    import torch
    from nvspeechplayer_amd import _native
    L = _native.load()
    buf = (ctypes.c_short * 16)()
    assert L.speechPlayer_synthesize(ctypes.c_void_p(12345), 16, buf) == 0          # invalid handle: returns 0 like a drained queue ...
    assert _native.last_error_code() == 1 and "invalid handle" in _native.last_error()   # ... but says so
    assert L.speechPlayer_batch_setOption(None, b"mode", 0) == -1
    if not torch.cuda.is_available():
        assert not L.speechPlayer_initialize(22050)
        assert _native.last_error_code() == 2 and "no HIP device" in _native.last_error()
        assert not L.speechPlayer_batch_create(22050, -1) and _native.last_error_code() == 2
    assert L.speechPlayer_voiceCount() == 4
    assert L.speechPlayer_ipa_frames(b"h", 1.0, 100.0, 0.5, 0, None, None, None, None, None, 0) == 1     # host-only producer: no GPU needed


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nvspeechplayer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "klatt_oracle" not in text.replace("restated by oracle/klatt_oracle.c", ""), f
                assert "from tests" not in text and "import tests" not in text, f


def test_kernel_arithmetic_helpers_on_host(tmp_path):
    """Markstein division == '/', fast_exp / fast_cos within 1 ulp of libm (tests/native/check_math.cpp)."""
    exe = str(tmp_path / "check_math")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "check_math.cpp")])
    out = json.loads(subprocess.check_output([exe]).decode())
    assert out["div_checked"] > 2.1e9 and out["div_bad"] == 0          # includes all 2^31 numerators of the noise division
    assert out["exp_max_ulp"] <= 1 and out["cos_max_ulp"] <= 1
    assert out["exp_wide_max_ulp"] <= 1 and out["cos_wide_max_ulp"] <= 2
    assert out["exp0"] == 1.0 and out["cos0"] == 1.0
    assert out["unreduced_checked"] > 7e6 and out["unreduced_bad"] == 0     # the short path returns the same bits


def test_workload_recipes():
    from nvspeechplayer_amd import workloads
    b = workloads.make("cfg1", 64)
    assert b.n_utt == 64 and len(b["min"]) == 128
    assert np.all(b.sample_counts() == 22051 + 1104)          # (max(22050,1103)+1) + (max(1102,1103)+1)
    assert b.algorithmic_bytes() == 2 * 64 * 23155 + 388 * 128
    assert b["frames"][0][44] == 1.0 and b["frames"][0][45] == 1.0 and b["frames"][0][5] == 1.0
    assert 79.9 < b["frames"][0][0] < 80.1 and b["frames"][0][0] == b["frames"][0][46]
    c = workloads.make("cfg2", 16)
    assert list(c.sample_counts()[:8]) == [8273, 29374, 29218, 20745, 12907, 41238, 13459, 29788]
    assert abs(c["frames"][1][0] / c["frames"][1 + len(c["min"]) // 2][0] - 2.0 ** (-1 / 64.0)) < 1e-12   # pitch variants 0 and 1
    assert np.array_equal(c.sample_counts()[:8], c.sample_counts()[8:16])      # pitch variants keep the timing
    d = workloads.make("cfg3", 8)
    assert np.all(d.sample_counts() <= 11025 + 2000)
    e = workloads.cfg4_voice_variants(3, 16)
    assert e.n_utt == 48 and np.array_equal(e.sample_counts()[:16], c.sample_counts()) and np.array_equal(e.sample_counts()[16:32], c.sample_counts())
    assert not np.array_equal(e["frames"][0], e["frames"][len(c["min"])])      # another variant, other formants
    e2 = workloads.cfg4_voice_variants(1, 16, first_variant=2)
    assert np.array_equal(e2["frames"], e["frames"][2 * len(c["min"]):])         # variants are reproducible
    s = c.slice(3, 5)
    assert s.n_utt == 5 and np.array_equal(s.sample_counts(), c.sample_counts()[3:8])
    # lengths without frames (what the ranks of a node use to deal shards), and pieces of the flat cfg4 list
    for wl in ("cfg1", "cfg2", "cfg3", "cfg4"):
        assert np.array_equal(workloads.make(wl, 300, first=1000).sample_counts(), workloads.sample_counts(wl, 300, 1000)), wl
    whole = workloads.cfg4_voice_variants(3, 1024)
    piece = workloads.make("cfg4", 1500, first=700)
    fs = whole["frame_start"]
    assert np.array_equal(piece["frames"], whole["frames"][fs[700]:fs[2200]]) and np.array_equal(piece["seeds"], whole["seeds"][700:2200])
    assert workloads.PER_GPU == {"cfg1": 4096, "cfg2": 65536, "cfg3": 125000, "cfg4": 32 * 16384}      # BASELINE configs[3] / 8, configs[4] / 8


def test_shard_bounds():
    from nvspeechplayer_amd.sharding import shard_bounds
    rng = np.random.default_rng(0)
    counts = rng.integers(1000, 40000, size=1000)
    for world in (1, 2, 4, 8):
        b = shard_bounds(counts, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        per = [counts[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(per) - min(per) <= 2 * counts.max()
    assert list(shard_bounds([], 4)) == [0, 0, 0, 0, 0]
    assert list(shard_bounds([5], 2)) in ([0, 0, 1], [0, 1, 1])


def test_headers_are_plain_c_and_cpp(tmp_path):
    """include/*.h must be consumable by a C host (C99) and a C++ host, and the frame must keep the reference's
    376-byte layout (reference src/frame.h:20-45); the C client of the GPU suite compiles against them too."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "speechPlayer.h"\n#include "speechPlayer_batch.h"\n'
                   'typedef char frame_is_376_bytes[sizeof(speechPlayer_frame_t) == 376 ? 1 : -1];\n'
                   'typedef char sample_is_2_bytes[sizeof(sample) == 2 ? 1 : -1];\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc,
                           os.path.join(ROOT, "tests", "native", "c_client.c")])


def test_forged_shape_hashes_are_refused():
    """The track planner recognises a frame by a 128-bit hash of its 45 shape values (csrc/klatt_plan.h) and no longer compares the
    values on a hit; what it took on trust is verified afterwards -- on the device in setUtterances (klatt_verify_shared), on the host
    in this view (speechPlayer_planTracksFacts).  Two frames that differ in ONE value and carry the SAME facts (forged here: the real
    hash function would need ~2^64 tries) must be refused, with the frame named; honest facts for the same frames plan fine."""
    import numpy as np
    from nvspeechplayer_amd import _native, workloads
    L = _native.load()
    b = workloads.make("cfg2", 16)
    fs, fr = b["frame_start"], b["frames"].copy()
    nf = len(fr)
    facts = np.zeros(nf, dtype=[("h0", "<u8"), ("h1", "<u8"), ("flags", "<u4"), ("pad", "<u4")])
    assert L.speechPlayer_frameFacts(fr.ctypes.data, nf, 22050, 0, facts.ctypes.data) == nf
    fade = np.maximum(b["fade"], 1).astype(np.uint32)
    nul = b["isnull"]
    args = lambda f: (len(fs) - 1, fs.ctypes.data, fr.ctypes.data, fade.ctypes.data, nul.ctypes.data, None, 4096, f, None, None, None, None)
    at = np.zeros(1, np.int64)
    n_tracks = L.speechPlayer_planTracksFacts(*args(facts.ctypes.data), at.ctypes.data)
    assert n_tracks > 0 and at[0] == -1
    assert n_tracks == L.speechPlayer_planTracksFacts(*args(None), at.ctypes.data) == L.speechPlayer_planTracks(*args(None)[:6], 4096, None, None, None, None)
    # frame k of the second utterance gets another cf2 and the facts of what it was
    k = int(fs[1]) + 2
    assert not nul[k]
    fr[k, 8] += 1.0
    honest = facts.copy()
    assert L.speechPlayer_frameFacts(fr.ctypes.data, nf, 22050, 0, honest.ctypes.data) == nf
    assert (honest["h0"][k], honest["h1"][k]) != (facts["h0"][k], facts["h1"][k])       # the hash sees one changed value
    assert np.array_equal(np.delete(honest, k), np.delete(facts, k))
    assert L.speechPlayer_planTracksFacts(*args(honest.ctypes.data), at.ctypes.data) > 0 and at[0] == -1
    assert L.speechPlayer_planTracksFacts(*args(facts.ctypes.data), at.ctypes.data) == -2       # the forged ones
    # (the forged frame is the first of its hash here, so the frame NAMED is a later one that really holds those values: the pair is what counts)
    assert at[0] >= 0 and "one 128-bit shape hash and different values" in _native.last_error()
    import re
    pair = [int(x) for x in re.search(r"frames (\d+) and (\d+)", _native.last_error()).groups()]
    assert k in pair and at[0] == pair[0] and not np.array_equal(fr[pair[0], 1:46], fr[pair[1], 1:46])
    # a value whose bit pattern equals its position's key no longer hides its partner (hi ^ lo of a * c is 0 when a is)
    one = np.zeros((2, 47)); out = np.zeros(2, dtype=facts.dtype)
    def splitmix(n):
        z = ((n + 1) * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return z ^ (z >> 31)
    one[:, 1] = np.array([splitmix(0)], np.uint64).view(np.float64)[0]      # parameter 1 ^ key(0) == 0 in the first chain
    one[1, 2] = 123.0                                                        # its partner in that product
    assert L.speechPlayer_frameFacts(one.ctypes.data, 2, 22050, 0, out.ctypes.data) == 2
    assert out["h0"][0] != out["h0"][1] and out["h1"][0] != out["h1"][1]


def test_global_options_are_checked():
    """speechPlayer_setGlobalOption (no GPU needed): known options take their values, "live_mode" takes 0 or 1 only, an unknown name is
    refused -- and the Python wrapper raises where the C call returns non-zero."""
    import nvspeechplayer_amd as eng
    try:
        eng.setGlobalOption("live_mode", 1)
        eng.setGlobalOption("live_alone", 7)
        with pytest.raises(ValueError):
            eng.setGlobalOption("live_mode", 2)
        with pytest.raises(ValueError):
            eng.setGlobalOption("no_such_option", 1)
    finally:
        eng.setGlobalOption("live_mode", 0)
        eng.setGlobalOption("live_alone", 1536)

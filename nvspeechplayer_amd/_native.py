"""Build and load the engine's C-ABI library (lib/libspeechPlayer.so).

The library is hand-written HIP for gfx950 compiled with hipcc; it is the only
synthesis path of this package.  If it is missing or cannot be loaded the package
raises -- there is no CPU fallback.
"""
import ctypes
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libspeechPlayer.so")
# translation units: (source, compiler, what it depends on besides itself)
#   klatt_engine.hip    kernels + the C-ABI's GPU half       hipcc for gfx950
#   frame_producer.cpp  IPA text -> frame streams (host)     g++
SOURCES = ["klatt_engine.hip", "frame_producer.cpp"]
INCLUDE_DIR = os.path.join(os.path.dirname(PKG_DIR), "include")
OBJ_DIR = os.path.join(PKG_DIR, "build_tmp")
# -amdgpu-sched-strategy=max-memory-clause: the machine scheduler keeps memory operations together instead of chasing occupancy
# (the kernels' occupancy is fixed by __launch_bounds__ and LDS anyway).  Same PCM; cfg2 14.65 -> 14.2 ms, cfg4 -3 %, the quiet
# kernels within 1 % (tools/ab_probe.py, profiles/r2_ab_noisy_variants.txt; max-ilp: +1.5 %, machine LICM off: +5 %).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
               "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
CXX_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wextra"]
LINK_FLAGS = ["-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,-rpath,/opt/rocm/lib"]

EXPORTS = [
    # include/speechPlayer.h
    "speechPlayer_initialize", "speechPlayer_queueFrame", "speechPlayer_synthesize",
    "speechPlayer_getLastIndex", "speechPlayer_terminate",
    # include/speechPlayer_batch.h
    "speechPlayer_batch_create", "speechPlayer_batch_destroy", "speechPlayer_batch_setOption",
    "speechPlayer_batch_setUtterances", "speechPlayer_batch_utteranceSamples",
    "speechPlayer_batch_totalSamples", "speechPlayer_batch_totalFrames", "speechPlayer_batch_sampleRate",
    "speechPlayer_batch_synthesize", "speechPlayer_batch_wait", "speechPlayer_batch_read",
    "speechPlayer_batch_readAll", "speechPlayer_batch_readAllAsync", "speechPlayer_batch_readWait", "speechPlayer_hostAlloc", "speechPlayer_hostFree",
    "speechPlayer_batch_readFloat", "speechPlayer_batch_digest", "speechPlayer_batch_getLastIndex", "speechPlayer_batch_devicePcm",
    "speechPlayer_batch_deviceOffset", "speechPlayer_batch_time", "speechPlayer_batch_kernelInfo",
    "speechPlayer_lastError", "speechPlayer_lastErrorCode", "speechPlayer_setNoiseSeed", "speechPlayer_synthesizeMany", "speechPlayer_setGlobalOption",
    "speechPlayer_synthesizeManyDevice", "speechPlayer_lastLiveKernelMs", "speechPlayer_lastLiveLaunches",
    "speechPlayer_ipa_frames", "speechPlayer_ipa_pack", "speechPlayer_batch_setIpa",
    "speechPlayer_text_available", "speechPlayer_text_clauses", "speechPlayer_text_fixups", "speechPlayer_text_toIpa", "speechPlayer_batch_setText",
    "speechPlayer_voiceCount", "speechPlayer_voiceName", "speechPlayer_applyVoiceToFrame",
    "speechPlayer_ipa_phonemeCount", "speechPlayer_ipa_phoneme",
    "speechPlayer_node_create", "speechPlayer_node_destroy", "speechPlayer_node_devices", "speechPlayer_node_setOption",
    "speechPlayer_node_setUtterances", "speechPlayer_node_synthesize", "speechPlayer_node_wait", "speechPlayer_node_totalSamples",
    "speechPlayer_node_read", "speechPlayer_node_getLastIndex", "speechPlayer_node_shardInfo", "speechPlayer_node_shardUtterances", "speechPlayer_node_part", "speechPlayer_node_setRecords", "speechPlayer_node_setIpa",
    "speechPlayer_node_time", "speechPlayer_planTracks", "speechPlayer_planDirect", "speechPlayer_frameFacts", "speechPlayer_planTracksFacts",
    "speechPlayer_batch_setUtterancesShared", "speechPlayer_batch_setRecords", "speechPlayer_batch_frames", "speechPlayer_batch_setIpaVoices",
    "speechPlayer_ipa_records", "speechPlayer_records_view", "speechPlayer_records_free", "speechPlayer_voiceIndex", "speechPlayer_voiceDefine", "speechPlayer_voicePresetCount",
]


def _deps(source):
    """Files a translation unit is built from: itself, every header / generated table beside it, the public headers."""
    own = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    pub = [os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR) if f.endswith(".h")]
    return [os.path.join(CSRC, source)] + own + pub


def _obj(source):
    return os.path.join(OBJ_DIR, os.path.splitext(source)[0] + ".o")


def _obj_stale(source):
    o = _obj(source)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(p) > t for p in _deps(source))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for src in SOURCES for p in _deps(src))


def build(force=False, verbose=False, extra_hipcc_flags=(), lib_path=None):
    """Compile the engine in-tree: hipcc for the HIP translation unit (cross-compiles for gfx950 without a GPU),
    g++ for the host-only one, objects under build_tmp/ (rebuilt when a source, header or generated table changed),
    linked into lib/libspeechPlayer.so.  `extra_hipcc_flags` / `lib_path`: A/B builds of the same ABI (tools/)."""
    out = lib_path or LIB_PATH
    variant = bool(extra_hipcc_flags) or lib_path is not None
    if not force and not variant and not _stale():
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time (the rank processes of bench.py, pytest-xdist workers and tools may all import at once); whoever
    # comes second finds the library current
    import fcntl
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not variant and not _stale():
            return out
        return _build_locked(hipcc, out, variant, force, verbose, extra_hipcc_flags)


def _build_locked(hipcc, out, variant, force, verbose, extra_hipcc_flags):
    objs = []
    for src in SOURCES:
        is_hip = src.endswith(".hip")
        # a variant build recompiles the HIP translation unit into an object of its own; host objects are shared
        own = variant and is_hip
        o = os.path.join(OBJ_DIR, os.path.basename(out) + "." + os.path.splitext(src)[0] + ".o") if own else _obj(src)
        if force or own or _obj_stale(src):
            flags = list(HIPCC_FLAGS)
            extra = list(extra_hipcc_flags)
            for f in [f for f in extra if f.startswith("--sched=")]:      # A/B builds: another machine-scheduler strategy ("none": the default)
                extra.remove(f)
                i = flags.index("-mllvm")
                del flags[i:i + 2]
                if f != "--sched=none":
                    flags += ["-mllvm", "-amdgpu-sched-strategy=" + f[len("--sched="):]]
            cmd = ([hipcc] + flags + extra if is_hip else ["g++"] + CXX_FLAGS) + ["-c", os.path.join(CSRC, src), "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    tmp = out + ".tmp.%d" % os.getpid()      # linked beside its place, then moved into it: nobody loads a half-written library
    cmd = [hipcc] + LINK_FLAGS + ["-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    if variant:      # a variant's object is scratch (3.5 MB each: 45 of them travelled with every snapshot at the end of round 4)
        for o in objs:
            if os.path.basename(o).startswith(os.path.basename(out) + "."):
                try:
                    os.remove(o)
                except OSError:
                    pass
    return out


_lib = None


def load():
    """ctypes handle of the engine library with prototypes set; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SPEECHPLAYER_LIB", LIB_PATH)     # A/B builds of the same ABI (tools/)
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the engine has no CPU fallback)" % path)
    L = ctypes.CDLL(path)
    vp, u32, i32, i64 = ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_longlong
    L.speechPlayer_initialize.restype = vp
    L.speechPlayer_initialize.argtypes = [i32]
    L.speechPlayer_queueFrame.restype = None
    L.speechPlayer_queueFrame.argtypes = [vp, vp, u32, u32, i32, ctypes.c_bool]
    L.speechPlayer_synthesize.restype = i32
    L.speechPlayer_synthesize.argtypes = [vp, u32, vp]
    L.speechPlayer_getLastIndex.restype = i32
    L.speechPlayer_getLastIndex.argtypes = [vp]
    L.speechPlayer_terminate.restype = None
    L.speechPlayer_terminate.argtypes = [vp]
    L.speechPlayer_setNoiseSeed.restype = i32
    L.speechPlayer_setNoiseSeed.argtypes = [vp, u32]
    L.speechPlayer_synthesizeMany.restype = i32
    L.speechPlayer_synthesizeMany.argtypes = [vp, i32, u32, vp, vp]
    L.speechPlayer_synthesizeManyDevice.restype = i32
    L.speechPlayer_synthesizeManyDevice.argtypes = [vp, i32, u32, vp, vp, vp]
    L.speechPlayer_lastLiveKernelMs.restype = ctypes.c_float
    L.speechPlayer_lastLiveKernelMs.argtypes = [i32]
    L.speechPlayer_lastLiveLaunches.restype = i32
    L.speechPlayer_lastLiveLaunches.argtypes = [i32]
    L.speechPlayer_setGlobalOption.restype = i32
    L.speechPlayer_setGlobalOption.argtypes = [ctypes.c_char_p, i32]
    L.speechPlayer_lastError.restype = ctypes.c_char_p
    L.speechPlayer_lastError.argtypes = []
    L.speechPlayer_lastErrorCode.restype = i32
    L.speechPlayer_lastErrorCode.argtypes = []
    L.speechPlayer_batch_create.restype = vp
    L.speechPlayer_batch_create.argtypes = [i32, i32]
    L.speechPlayer_batch_destroy.restype = None
    L.speechPlayer_batch_destroy.argtypes = [vp]
    L.speechPlayer_batch_setOption.restype = i32
    L.speechPlayer_batch_setOption.argtypes = [vp, ctypes.c_char_p, i32]
    L.speechPlayer_batch_setUtterances.restype = i32
    L.speechPlayer_batch_setUtterances.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.speechPlayer_batch_utteranceSamples.restype = i64
    L.speechPlayer_batch_utteranceSamples.argtypes = [vp, i64]
    L.speechPlayer_batch_totalSamples.restype = i64
    L.speechPlayer_batch_totalSamples.argtypes = [vp]
    L.speechPlayer_batch_totalFrames.restype = i64
    L.speechPlayer_batch_totalFrames.argtypes = [vp]
    L.speechPlayer_batch_synthesize.restype = i32
    L.speechPlayer_batch_synthesize.argtypes = [vp]
    L.speechPlayer_batch_wait.restype = i32
    L.speechPlayer_batch_wait.argtypes = [vp]
    L.speechPlayer_batch_read.restype = i64
    L.speechPlayer_batch_read.argtypes = [vp, i64, vp, i64]
    L.speechPlayer_batch_readFloat.restype = i64
    L.speechPlayer_batch_readFloat.argtypes = [vp, i64, vp, i64]
    L.speechPlayer_batch_readAll.restype = i64
    L.speechPlayer_batch_readAll.argtypes = [vp, vp, i64, vp]
    L.speechPlayer_batch_readAllAsync.restype = i64
    L.speechPlayer_batch_readAllAsync.argtypes = [vp, vp, i64, vp]
    L.speechPlayer_batch_readWait.restype = i32
    L.speechPlayer_batch_readWait.argtypes = [vp]
    L.speechPlayer_hostAlloc.restype = vp
    L.speechPlayer_hostAlloc.argtypes = [i64]
    L.speechPlayer_hostFree.restype = None
    L.speechPlayer_hostFree.argtypes = [vp]
    L.speechPlayer_batch_digest.restype = i32
    L.speechPlayer_batch_digest.argtypes = [vp, vp, vp]
    L.speechPlayer_batch_getLastIndex.restype = i32
    L.speechPlayer_batch_getLastIndex.argtypes = [vp, i64]
    L.speechPlayer_batch_devicePcm.restype = vp
    L.speechPlayer_batch_devicePcm.argtypes = [vp]
    L.speechPlayer_batch_deviceOffset.restype = i64
    L.speechPlayer_batch_deviceOffset.argtypes = [vp, i64]
    L.speechPlayer_batch_time.restype = i32
    L.speechPlayer_batch_time.argtypes = [vp, i32, vp]
    L.speechPlayer_batch_kernelInfo.restype = i32
    L.speechPlayer_batch_kernelInfo.argtypes = [vp, vp, i32]
    L.speechPlayer_batch_sampleRate.restype = i32
    L.speechPlayer_batch_sampleRate.argtypes = [vp]
    f64 = ctypes.c_double
    L.speechPlayer_ipa_frames.restype = i64
    L.speechPlayer_ipa_frames.argtypes = [ctypes.c_char_p, f64, f64, f64, i32, ctypes.c_char_p, vp, vp, vp, vp, i64]
    L.speechPlayer_ipa_pack.restype = i64
    L.speechPlayer_ipa_pack.argtypes = [i32, i64, vp, f64, vp, f64, ctypes.c_char_p, ctypes.c_char_p, f64, vp, vp, vp, vp, vp, i64]
    L.speechPlayer_text_available.restype = i32
    L.speechPlayer_text_available.argtypes = []
    L.speechPlayer_text_clauses.restype = i64
    L.speechPlayer_text_clauses.argtypes = [ctypes.c_char_p, vp, vp, vp, vp, i64]
    L.speechPlayer_text_fixups.restype = i64
    L.speechPlayer_text_fixups.argtypes = [ctypes.c_char_p, vp, i64]
    L.speechPlayer_text_toIpa.restype = i64
    L.speechPlayer_text_toIpa.argtypes = [ctypes.c_char_p, ctypes.c_char_p, vp, i64]
    L.speechPlayer_batch_setText.restype = i32
    L.speechPlayer_batch_setText.argtypes = [vp, i64, vp, ctypes.c_char_p, f64, vp, f64, ctypes.c_char_p, vp]
    L.speechPlayer_batch_setIpa.restype = i32
    L.speechPlayer_batch_setIpa.argtypes = [vp, i64, vp, f64, vp, f64, ctypes.c_char_p, ctypes.c_char_p, f64, vp]
    L.speechPlayer_node_create.restype = vp
    L.speechPlayer_node_create.argtypes = [i32, i32, vp]
    L.speechPlayer_node_destroy.restype = None
    L.speechPlayer_node_destroy.argtypes = [vp]
    L.speechPlayer_node_devices.restype = i32
    L.speechPlayer_node_devices.argtypes = [vp]
    L.speechPlayer_node_setOption.restype = i32
    L.speechPlayer_node_setOption.argtypes = [vp, ctypes.c_char_p, i32]
    L.speechPlayer_node_setUtterances.restype = i32
    L.speechPlayer_node_setUtterances.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.speechPlayer_node_synthesize.restype = i32
    L.speechPlayer_node_synthesize.argtypes = [vp]
    L.speechPlayer_node_wait.restype = i32
    L.speechPlayer_node_wait.argtypes = [vp]
    L.speechPlayer_node_totalSamples.restype = i64
    L.speechPlayer_node_totalSamples.argtypes = [vp]
    L.speechPlayer_node_read.restype = i64
    L.speechPlayer_node_read.argtypes = [vp, i64, vp, i64]
    L.speechPlayer_node_getLastIndex.restype = i32
    L.speechPlayer_node_getLastIndex.argtypes = [vp, i64]
    L.speechPlayer_node_shardInfo.restype = i32
    L.speechPlayer_node_shardInfo.argtypes = [vp, i32, vp, vp, vp, vp]
    L.speechPlayer_node_shardUtterances.restype = i64
    L.speechPlayer_node_shardUtterances.argtypes = [vp, i32, vp, i64]
    L.speechPlayer_node_setRecords.restype = i32
    L.speechPlayer_node_setRecords.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp]
    L.speechPlayer_node_setIpa.restype = i32
    L.speechPlayer_node_setIpa.argtypes = [vp, i32, i64, vp, f64, vp, f64, ctypes.c_char_p, vp, ctypes.c_char_p, f64, vp]
    L.speechPlayer_node_part.restype = vp
    L.speechPlayer_node_part.argtypes = [vp, i32]
    L.speechPlayer_node_time.restype = i32
    L.speechPlayer_node_time.argtypes = [vp, i32, vp]
    L.speechPlayer_voiceCount.restype = i32
    L.speechPlayer_voiceCount.argtypes = []
    L.speechPlayer_voiceName.restype = ctypes.c_char_p
    L.speechPlayer_voiceName.argtypes = [i32]
    L.speechPlayer_applyVoiceToFrame.restype = i32
    L.speechPlayer_applyVoiceToFrame.argtypes = [vp, ctypes.c_char_p]
    L.speechPlayer_ipa_phonemeCount.restype = i32
    L.speechPlayer_ipa_phonemeCount.argtypes = []
    L.speechPlayer_ipa_phoneme.restype = i32
    L.speechPlayer_ipa_phoneme.argtypes = [i32, vp, i32, vp, vp, vp]
    L.speechPlayer_planTracks.restype = i64
    L.speechPlayer_planTracks.argtypes = [i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]
    L.speechPlayer_frameFacts.restype = i64
    L.speechPlayer_frameFacts.argtypes = [vp, i64, i32, i32, vp]
    L.speechPlayer_planDirect.restype = i64
    L.speechPlayer_planDirect.argtypes = [i64, vp, vp, vp, vp, vp]
    L.speechPlayer_planTracksFacts.restype = i64
    L.speechPlayer_planTracksFacts.argtypes = [i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.speechPlayer_batch_setUtterancesShared.restype = i32
    L.speechPlayer_batch_setUtterancesShared.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.speechPlayer_batch_setRecords.restype = i32
    L.speechPlayer_batch_setRecords.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp]
    L.speechPlayer_batch_frames.restype = i64
    L.speechPlayer_batch_frames.argtypes = [vp, i64, vp, vp, vp, vp, vp, i64]
    L.speechPlayer_batch_setIpaVoices.restype = i32
    L.speechPlayer_batch_setIpaVoices.argtypes = [vp, i64, vp, f64, vp, f64, ctypes.c_char_p, vp, f64, vp]
    L.speechPlayer_ipa_records.restype = vp
    L.speechPlayer_ipa_records.argtypes = [i32, i64, vp, f64, vp, f64, ctypes.c_char_p, vp, ctypes.c_char_p, f64]
    L.speechPlayer_records_view.restype = i32
    L.speechPlayer_records_view.argtypes = [vp, vp]
    L.speechPlayer_records_free.restype = None
    L.speechPlayer_records_free.argtypes = [vp]
    L.speechPlayer_voicePresetCount.restype = i32
    L.speechPlayer_voicePresetCount.argtypes = []
    L.speechPlayer_voiceIndex.restype = i32
    L.speechPlayer_voiceIndex.argtypes = [ctypes.c_char_p]
    L.speechPlayer_voiceDefine.restype = i32
    L.speechPlayer_voiceDefine.argtypes = [ctypes.c_char_p, i32, vp, vp, vp]
    _lib = L
    return L


def last_error():
    return (load().speechPlayer_lastError() or b"").decode("utf8", "replace")


def last_error_code():
    """0 after a call that succeeded, a SPEECHPLAYER_ERR_* code after one that failed (include/speechPlayer_batch.h)."""
    return int(load().speechPlayer_lastErrorCode())

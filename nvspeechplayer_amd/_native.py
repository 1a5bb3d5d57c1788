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
SOURCES = ["klatt_engine.hip"]
HEADERS = ["klatt_device.h", "klatt_systolic.h", "klatt_math.h", os.path.join("..", "..", "include", "speechPlayer.h"),
           os.path.join("..", "..", "include", "speechPlayer_batch.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wall", "-Wno-unused-function", "-Wl,-rpath,/opt/rocm/lib"]

EXPORTS = [
    # include/speechPlayer.h
    "speechPlayer_initialize", "speechPlayer_queueFrame", "speechPlayer_synthesize",
    "speechPlayer_getLastIndex", "speechPlayer_terminate",
    # include/speechPlayer_batch.h
    "speechPlayer_batch_create", "speechPlayer_batch_destroy", "speechPlayer_batch_setOption",
    "speechPlayer_batch_setUtterances", "speechPlayer_batch_utteranceSamples",
    "speechPlayer_batch_totalSamples", "speechPlayer_batch_totalFrames",
    "speechPlayer_batch_synthesize", "speechPlayer_batch_wait", "speechPlayer_batch_read",
    "speechPlayer_batch_readAll", "speechPlayer_batch_readFloat", "speechPlayer_batch_getLastIndex", "speechPlayer_batch_devicePcm",
    "speechPlayer_batch_deviceOffset", "speechPlayer_batch_time", "speechPlayer_batch_kernelInfo",
    "speechPlayer_lastError", "speechPlayer_setNoiseSeed", "speechPlayer_synthesizeMany",
]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    """Compile the HIP engine in-tree (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % LIB_PATH)
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


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
    L.speechPlayer_lastError.restype = ctypes.c_char_p
    L.speechPlayer_lastError.argtypes = []
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
    _lib = L
    return L


def last_error():
    return (load().speechPlayer_lastError() or b"").decode("utf8", "replace")

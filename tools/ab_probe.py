"""A/B timing of alternative builds of the engine library (same ABI, other -D flags) on the GPU box.

    python tools/ab_probe.py build name=-DFLAG=1,-DOTHER=2 ...      (here: cross-compile variants into lib/variants/)
    python tools/ab_probe.py run [name ...]                          (on the GPU box: every variant in a process of its own)

Per variant: kernel ms of cfg2 (65 536 utterances), of cfg2 with rotated frame lists (lanes of a wave never fade together,
tools/mixed_probe.py) and of cfg2 at 16 384 utterances, plus the device-side digest of each PCM pool -- variants must agree.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARDIR = os.path.join(ROOT, "nvspeechplayer_amd", "lib", "variants")


def build(specs):
    from concurrent.futures import ThreadPoolExecutor
    from nvspeechplayer_amd import _native
    os.makedirs(VARDIR, exist_ok=True)

    def one(spec):
        name, _, flags = spec.partition("=")
        out = os.path.join(VARDIR, "libspeechPlayer_%s.so" % name)
        _native.build(extra_hipcc_flags=[f for f in flags.split(",") if f], lib_path=out)
        return out
    with ThreadPoolExecutor(4) as ex:
        for p in ex.map(one, specs):
            print("built", p)


def measure(workloads_wanted):
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer, workloads
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mixed_probe import rotate, jitter
    out = {}
    base = workloads.make("cfg2", 65536)
    for key in workloads_wanted:
        b = {"cfg2": lambda: base, "rot": lambda: rotate(base), "cfg2_16k": lambda: base.slice(0, 16384), "jit": lambda: jitter(base), "cfg1": lambda: workloads.make("cfg1", 4096), "cfg1_1k": lambda: workloads.make("cfg1", 1024), "cfg1_256": lambda: workloads.make("cfg1", 256), "cfg1_16": lambda: workloads.make("cfg1", 16), "vowels64k": lambda: workloads.make("cfg1", 65536),
             "cfg2_2k": lambda: base.slice(0, 2048), "cfg2_4k": lambda: base.slice(0, 4096), "cfg2_8k": lambda: base.slice(0, 8192), "cfg2_14k": lambda: base.slice(0, 14000),
             "cfg4": lambda: workloads.make("cfg4", 32768), "cfg3": lambda: workloads.make("cfg3", 125000)}[key]()
        bp = BatchPlayer(b["sr"])
        bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
        bp.time(2)
        ms = float(np.mean(bp.time(6)))
        out[key] = {"ms": round(ms, 3), "digest": "%016x" % bp.digest(), "scratch": bp.kernelInfo()["scratch_bytes"], "vgprs": bp.kernelInfo()["vgprs"]}
        bp.close()
    return out


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "one":
        print(json.dumps(measure(sys.argv[2:])), flush=True)
    else:
        names = [a for a in sys.argv[2:] if not a.startswith("+")] or sorted(f[len("libspeechPlayer_"):-3] for f in os.listdir(VARDIR) if f.endswith(".so"))
        wl = [a[1:] for a in sys.argv[2:] if a.startswith("+")] or ["cfg2", "rot", "cfg2_16k"]
        ref = {}
        for n in names:
            env = dict(os.environ, SPEECHPLAYER_LIB=os.path.join(VARDIR, "libspeechPlayer_%s.so" % n))
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "one"] + wl, env=env, capture_output=True, timeout=300)
                line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
                d = json.loads(line[-1]) if line else {"error": res.stderr.decode()[-300:]}
            except subprocess.TimeoutExpired:
                d = {"error": "timeout"}
            ok = ""
            for k, v in d.items():
                if isinstance(v, dict):
                    ok += "" if ref.setdefault(k, v["digest"]) == v["digest"] else " DIGEST-MISMATCH(%s)" % k
            print("%-14s %s%s" % (n, "  ".join("%s %.2f ms" % (k, v["ms"]) if isinstance(v, dict) else "%s %s" % (k, v) for k, v in d.items()),
                                 ok + ("  [scratch %s]" % next(iter(d.values())).get("scratch") if d and isinstance(next(iter(d.values())), dict) else "")), flush=True)

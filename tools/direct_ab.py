"""Timing of library variants on the direct stages' batches (tools/ab_probe.py build NAME=-DFLAGS first).
   python tools/direct_ab.py NAME ... [+workload ...] [n=65536]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VARDIR = os.path.join(ROOT, "nvspeechplayer_amd", "lib", "variants")

def one(wl, n):
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer, workloads
    from mixed_probe import jitter, distinct
    base = workloads.make("cfg2", n)
    out = {}
    for w in wl:
        b = {"cfg2": lambda: base, "jittered": lambda: jitter(base), "distinct": lambda: distinct(base), "all_different": lambda: jitter(distinct(base))}[w]()
        for mode in (0, 1):
            bp = BatchPlayer(b["sr"], mode=mode)
            bp.setOption("tracks", 0)
            bp.setOption("direct", 2)
            bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
            bp.time(1)
            out["%s/%d" % (w, mode)] = [round(float(np.median(bp.time(5))), 2), "%016x" % bp.digest()]
            bp.close()
    return out

if __name__ == "__main__":
    if sys.argv[1] == "one":
        n = [int(a[2:]) for a in sys.argv[2:] if a.startswith("n=")]
        print(json.dumps(one([a for a in sys.argv[2:] if not a.startswith("n=")], n[0] if n else 65536)), flush=True)
        sys.exit(0)
    names = [a for a in sys.argv[1:] if not a.startswith("+") and not a.startswith("n=")]
    wl = [a[1:] for a in sys.argv[1:] if a.startswith("+")] or ["all_different"]
    nn = [a for a in sys.argv[1:] if a.startswith("n=")]
    for nme in names:
        env = dict(os.environ)
        if nme != "base":
            env["SPEECHPLAYER_LIB"] = os.path.join(VARDIR, "libspeechPlayer_%s.so" % nme)
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "one"] + wl + nn, env=env, capture_output=True, timeout=400)
        line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": res.stderr.decode()[-400:]}
        print("%-10s %s" % (nme, "  ".join("%s %s ms (%s)" % (k, v[0], v[1][:6]) if isinstance(v, list) else str(v) for k, v in d.items())), flush=True)

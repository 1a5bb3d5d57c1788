"""The float experiment of DESIGN.md section 4 ("A float signal path"): a build with -DKLATT_SIGNAL_F32 (flat filter stages in float: klatt_device.h
sig_t) against the shipped double build -- kernel time, and how far its PCM is from the double PCM, overall and on the WORST
utterance (the tolerance north_star allows a float path is RMS < 1e-5 of full scale; an utterance over it is audibly the same
and formally out).

    python tools/ab_probe.py build base= f32=-DKLATT_SIGNAL_F32        (here)
    python tools/f32_probe.py [variant ...]                           (GPU box; default: f32)

Batches: BASELINE configs[2] (timing at 65 536 utterances, PCM of the first 8 192), the same with jittered durations, configs[4],
configs[1], the parity corpus of the test suite (tests/scenarios.py) and 6 000 random noisy utterances (tests/test_gpu_parity.py).  Each build runs in a process of its own (SPEECHPLAYER_LIB).
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
VARDIR = os.path.join(ROOT, "nvspeechplayer_amd", "lib", "variants")


def batches():
    from nvspeechplayer_amd import workloads
    from mixed_probe import jitter
    import numpy as np
    from tests import scenarios
    from tests.test_gpu_parity import make_batch, random_batch
    base = workloads.make("cfg2", 65536)
    corpus = make_batch([s for s in scenarios.build_scenarios(scenarios.Ref()) if s.batchable and s.sr == 22050])
    rnd = random_batch(np.random.default_rng(5), 6000, quiet_fraction=0.0)       # random formants, bandwidths down to 30 Hz, gains up to 2.5
    return [("cfg2", base, 8192), ("jittered", jitter(base), 8192), ("cfg4", workloads.make("cfg4", 32768), 8192), ("cfg1", workloads.make("cfg1", 4096), 4096),
            ("corpus", corpus, 1 << 30), ("random", rnd, 1 << 30)]


def child(outdir):
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer
    res = {}
    for name, b, keep in batches():
        for mode in (0, 1):
            bp = BatchPlayer(b["sr"] if "sr" in b else 22050, mode=mode)
            bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
            bp.synthesize()
            bp.time(2)
            ms = float(np.mean(bp.time(6)))
            info = bp.kernelInfo()
            pcm, starts = bp.readAll()
            end = int(starts[min(keep, len(starts) - 1)])
            np.save(os.path.join(outdir, "%s_%d.npy" % (name, mode)), pcm[:end])
            np.save(os.path.join(outdir, "%s_starts.npy" % name), starts[:min(keep, len(starts) - 1) + 1])
            res["%s/%d" % (name, mode)] = {"ms": ms, "vgprs": info["vgprs"], "scratch": info["scratch_bytes"], "tracked": info["tracked_utterances"]}
            bp.close()
    print(json.dumps(res), flush=True)


def compare(a, b, starts):
    import numpy as np
    d = (a.astype(np.float64) - b.astype(np.float64)) / 32768.0
    sq = np.concatenate([[0.0], np.cumsum(d * d)])
    n = np.diff(starts)
    ok = n > 0
    per = np.sqrt((sq[starts[1:]] - sq[starts[:-1]])[ok] / n[ok])
    return {"rms": float(np.sqrt(np.mean(d * d))), "worst_utt_rms": float(per.max()), "utt_over_1e-5": int((per > 1e-5).sum()), "utterances": int(ok.sum()),
            "differing": float(np.mean(a != b)), "max_lsb": int(np.abs(a.astype(np.int32) - b).max())}


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    import numpy as np
    names = sys.argv[1:] or ["f32"]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n in ["base"] + names:
            os.makedirs(os.path.join(tmp, n))
            env = dict(os.environ, SPEECHPLAYER_LIB=os.path.join(VARDIR, "libspeechPlayer_%s.so" % n))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(tmp, n)], env=env, capture_output=True, timeout=900)
            line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            if not line:
                print(n, "failed:", r.stderr.decode()[-500:])
                sys.exit(1)
            out[n] = json.loads(line[-1])
        for key in out["base"]:
            name, mode = key.split("/")
            starts = np.load(os.path.join(tmp, "base", "%s_starts.npy" % name))
            ref = np.load(os.path.join(tmp, "base", "%s_0.npy" % name))          # the double build, MODE_EXACT
            print("%-9s mode %s  double %7.3f ms (%d VGPRs)" % (name, mode, out["base"][key]["ms"], out["base"][key]["vgprs"]), flush=True)
            for n in names:
                c = compare(np.load(os.path.join(tmp, n, "%s_%s.npy" % (name, mode))), ref, starts)
                print("    %-10s %7.3f ms (%d VGPRs, scratch %d)  vs double MODE_EXACT: RMS %.2e  worst utterance %.2e  (%d of %d over 1e-5)  %.2f %% of samples differ, by at most %d LSB" % (
                    n, out[n][key]["ms"], out[n][key]["vgprs"], out[n][key]["scratch"], c["rms"], c["worst_utt_rms"], c["utt_over_1e-5"], c["utterances"], 100 * c["differing"], c["max_lsb"]), flush=True)

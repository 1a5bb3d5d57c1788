"""Where does speechPlayer_batch_setUtterances spend its time?  (SPEECHPLAYER_SET_TRACE / SPEECHPLAYER_PLAN_TRACE laps on stderr.)
    python tools/set_trace.py [workload] [utterances]"""
import os, sys, time
os.environ["SPEECHPLAYER_SET_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nvspeechplayer_amd import BatchPlayer, workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
t0 = time.perf_counter(); b = workloads.make(wl, n) if wl != "all_different" else workloads.all_different(workloads.make("cfg2", n)); t1 = time.perf_counter()
print("build %s: %.3f s" % (b["name"], t1 - t0), file=sys.stderr)
bp = BatchPlayer(b["sr"])
if "pinned" in sys.argv:      # the frames in page-locked memory (speechPlayer_hostAlloc)
    import numpy as np
    from nvspeechplayer_amd import host_array
    fr = host_array(b["frames"].shape, np.float64); fr[...] = b["frames"]; b["frames"] = fr
for rep in range(3):
    t0 = time.perf_counter()
    bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    print("setUtterances #%d: %.3f s" % (rep, time.perf_counter() - t0), file=sys.stderr)
bp.close()

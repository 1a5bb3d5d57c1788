"""klatt_expand_frames / klatt_verify_shared / klatt_frame_facts at configs[2]'s size with one list per utterance (1.58 M frames): run under
rocprofv3 --kernel-trace --stats for the kernels' durations (bytes per frame: expand 32 in + 392 out; verify 376 in; facts 376 in + 24 out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import ipa, workloads, host_array
sp = workloads.cfg2_spec(65536)
pk = ipa.records_for_batch(sp["texts"], textOf=sp["textOf"], basePitch=sp["basePitch"], clauseType=".", trailing_silence_ms=150.0)
ls, lo = pk["list_start"], pk["list_of"]
counts = (ls[1:] - ls[:-1])[lo]
fs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
rows = np.repeat(ls[lo] - fs[:-1], counts) + np.arange(int(fs[-1]), dtype=np.int64)
rec = pk["records"][rows]
bp = eng.BatchPlayer(22050)
for rep in range(4):
    t = time.perf_counter(); bp.setRecords(pk["shapes"], fs, rec, None, sp["noiseSeed"]); dt = time.perf_counter() - t
print("setRecords, one list per utterance: %d records, %.1f ms per call" % (len(rec), dt * 1e3))
plain = workloads.make("cfg2", 65536)
fr = host_array(plain["frames"].shape, np.float64); fr[...] = plain["frames"]
for rep in range(4):
    t = time.perf_counter(); bp.setUtterances(plain["frame_start"], fr, plain["min"], plain["fade"], plain["index"], plain["isnull"], plain["seeds"]); dt = time.perf_counter() - t
print("setUtterances, page-locked frames: %.1f ms per call" % (dt * 1e3))
bp.synthesize(); print("digest %016x" % bp.digest())

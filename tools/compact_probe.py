"""Probe of the compact batch forms on the GPU: group sizes, set times and kernel times of plain / shared / records."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvspeechplayer_amd as eng
from nvspeechplayer_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
plain = workloads.make("cfg2", n, first=first)
a = eng.BatchPlayer(22050)
for rep in range(2):
    t = time.perf_counter(); a.setUtterances(plain["frame_start"], plain["frames"], plain["min"], plain["fade"], plain["index"], plain["isnull"], plain["seeds"]); ta = time.perf_counter() - t
a.time(2); print("plain   set %.4f s kernel %.3f ms" % (ta, float(np.mean(a.time(10)))), a.kernelInfo())
lists, list_of, seeds = workloads.shared("cfg2", n, first=first)
b = eng.BatchPlayer(22050)
for rep in range(2):
    t = time.perf_counter(); b.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds); tb = time.perf_counter() - t
b.time(2); print("shared  set %.4f s kernel %.3f ms" % (tb, float(np.mean(b.time(10)))), b.kernelInfo())
c = eng.BatchPlayer(22050)
for rep in range(2):
    t = time.perf_counter(); spec = workloads.cfg2_spec(n, first=first); t1 = time.perf_counter() - t
    t = time.perf_counter(); c.setIpa(**spec); tc = time.perf_counter() - t
c.time(2); print("records spec %.4f s set %.4f s kernel %.3f ms" % (t1, tc, float(np.mean(c.time(10)))), c.kernelInfo())
a.synthesize(); b.synthesize(); c.synthesize()
da, db, dc = a.digest(True)[1], b.digest(True)[1], c.digest(True)[1]
print("digests equal:", np.array_equal(da, db), np.array_equal(da, dc))

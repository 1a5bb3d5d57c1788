import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nvspeechplayer_amd as eng
from tests import scenarios
ref = scenarios.Ref()
fa = scenarios.vowel_frame(ref, "a", 120.0)
fs = scenarios.vowel_frame(ref, "s", 120.0)
for name, fr in (("vowel", fa), ("fricative", fs)):
    p = eng.SpeechPlayer(22050)
    p.queueFrameSamples(eng.Frame.from_array(fr), 2000000, 100)
    p.synthesize(1000)
    t0 = time.perf_counter(); b = p.synthesize(400000); dt = time.perf_counter() - t0
    print("stream %s: %d samples in %.1f ms = %.3f us/sample" % (name, b.length, dt * 1e3, dt / b.length * 1e6))
    p.close()
    bp = eng.BatchPlayer(22050, layout=0)
    bp.setUtterances([0, 1], fr[None, :], [400000], [100])
    ms = bp.time(3)
    print("batch lane kernel %s: %.1f ms = %.3f us/sample" % (name, ms[-1], ms[-1] * 1e3 / 400101))
    bp.close()

"""Census of the loops of one kernel as compiled (device assembly of klatt_engine.hip with the build's flags): per loop, its
instructions by kind -- the figures DESIGN.md quotes for the steady loops of the flat stages.  No GPU needed.

    python tools/isa_census.py [mangled-kernel-name-substring] [min f64 operations per loop]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvspeechplayer_amd import _native

KERNEL = sys.argv[1] if len(sys.argv) > 1 else "klatt_systolicILi0ELb1ELi16ELi2ELb1ELb0ELb1E"      # MODE_EXACT, noisy, 16-sample hand-overs, 2 per CU, flat
MIN_F64 = int(sys.argv[2]) if len(sys.argv) > 2 else 200

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "engine.s")
    flags = [f for f in _native.HIPCC_FLAGS if f != "-fPIC"]
    subprocess.check_call(["hipcc"] + flags + ["--cuda-device-only", "-S", "-x", "hip", os.path.join(ROOT, "nvspeechplayer_amd", "csrc", "klatt_engine.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    src = open(out).read().split("\n")
start = [i for i, l in enumerate(src) if l.startswith("_ZN5klatt") and KERNEL in l and l.rstrip().split(":")[0].endswith("E") and ":" in l][0]
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
lines = src[start:end]
meta = [l.strip() for l in src if KERNEL in l and ".name:" in l]
i = next(k for k, l in enumerate(src) if KERNEL in l and ".name:" in l)
info = {m.group(1): m.group(2) for m in (re.match(r"\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", l) for l in src[i - 12:i + 14]) if m}
print("%s: %d lines of ISA, %s" % (src[start].split(":")[0], len(lines), ", ".join("%s %s" % kv for kv in sorted(info.items()))))
labels = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for k, l in enumerate(lines):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(1), k) < k:
        loops.append((labels[m.group(1)], k))
print("loops with at least %d f64 operations (first line .. last line: instructions | f64, global loads, LDS, barriers, v_mul_lo_u32 | the most frequent):" % MIN_F64)
for a, b in loops:
    body = [l.strip() for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
    c = {}
    for l in body:
        c[l.split()[0]] = c.get(l.split()[0], 0) + 1
    f64 = sum(v for k, v in c.items() if "f64" in k)
    if f64 >= MIN_F64:
        print("%6d ..%6d: %5d | f64 %4d  global_load %3d  ds %3d  s_barrier %d  v_mul_lo_u32 %2d | %s" % (
            a, b, len(body), f64, sum(v for k, v in c.items() if k.startswith("global_load")), sum(v for k, v in c.items() if k.startswith("ds_")),
            c.get("s_barrier", 0), c.get("v_mul_lo_u32", 0), " ".join("%s:%d" % kv for kv in sorted(c.items(), key=lambda x: -x[1])[:8])))

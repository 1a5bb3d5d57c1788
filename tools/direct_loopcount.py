#!/usr/bin/env python3
"""Instruction census of the direct kernel's mixed sample loop, stage by stage (compiles tools/direct_census.hip with one stage each).
   python tools/direct_loopcount.py [mode]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
extra = sys.argv[2:]
names = {1: "T0", 2: "T1", 4: "T2-4", 8: "T5", 16: "T6", 32: "T7"}
for mask, name in names.items():
    out = "/tmp/direct_lc_%d.s" % mask
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause",
                           "--cuda-device-only", "-S", "-DCENSUS_MODE=" + mode, "-DKLATT_DIRECT_STAGES=%d" % mask] + extra + [os.path.join(ROOT, "tools", "direct_census.hip"), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    # basic blocks with their loop depth comments; the mixed loop is the Depth=2 loop with the most f64 instructions
    loops = {}
    cur = None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            hdr = re.search(r"Loop Header: Depth=(\d+)", l)
            par = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            key = None
            if hdr and hdr.group(1) == "2":
                key = cur[2:]
            elif par and par.group(2) == "2":
                key = par.group(1)
            loops.setdefault(key, [])
            curkey = key
            continue
        if cur is None:
            continue
        par = re.search(r"; %bb\.\d+:\s+;\s+in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if par:
            curkey = par.group(1) if par.group(2) == "2" else None
            loops.setdefault(curkey, [])
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        loops.setdefault(curkey, []).append(t.split()[0])
    best = None
    for k, ins in loops.items():
        if k is None:
            continue
        f64 = sum(1 for i in ins if i.startswith("global_load_dwordx4"))      # the mixed loop holds the switch block's loads
        if best is None or f64 > best[1]:
            best = (k, f64, ins)
    k, _, ins = best
    f64 = sum(1 for i in ins if 'f64' in i)
    valu = sum(1 for i in ins if i.startswith("v_"))
    salu = sum(1 for i in ins if i.startswith("s_") and not i.startswith("s_waitcnt") and not i.startswith("s_nop"))
    lds = sum(1 for i in ins if i.startswith("ds_"))
    vmem = sum(1 for i in ins if i.startswith("global_") or i.startswith("buffer_") or i.startswith("scratch_"))
    cnd = sum(1 for i in ins if i.startswith("v_cndmask"))
    mov = sum(1 for i in ins if i.startswith("v_mov"))
    unroll = 2
    print("%-5s loop %s: %4d instructions (per sample at unroll %d: %5.1f): VALU %d (f64 %d, cndmask %d, mov %d), SALU %d, LDS %d, VMEM %d, waitcnt %d" % (
        name, k, len(ins), unroll, len(ins) / unroll, valu, f64, cnd, mov, salu, lds, vmem, sum(1 for i in ins if i.startswith("s_waitcnt"))))

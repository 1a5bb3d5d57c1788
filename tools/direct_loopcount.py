"""Instructions per sample of each lean direct stage's sample loop, from the census assembly (tools/direct_census.sh leaves /tmp/direct_census_MASK.s).

    python tools/direct_loopcount.py MODE [samples per trip]
For every stage mask: VALU / SALU / LDS / branch instructions in the innermost loops (Depth=2), the fade-start blocks (those with global loads) left out.
"""
import re
import sys

trip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for mask in (1, 64, 2, 4, 8, 16, 32):
    try:
        lines = open("/tmp/direct_census_%d.s" % mask).read().split("\n")
    except OSError:
        continue
    # basic blocks
    blocks, cur, name = [], [], "entry"
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
        if m:
            blocks.append((name, cur)); name, cur = m.group(1) + m.group(2), []
        elif re.match(r"^; %bb\.\d+:(.*)", ln):
            blocks.append((name, cur)); name, cur = ln, []
        elif "This Inner Loop Header: Depth=2" in ln:
            name += " Depth=2"
        elif ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;"):
            cur.append(ln.strip())
    blocks.append((name, cur))
    loops = {}
    for name, ins in blocks:
        m = re.search(r"Depth=2", name)
        if not m:
            continue
        hdr = re.search(r"Header=(BB\d+_\d+)", name)
        key = hdr.group(1) if hdr else re.match(r"\.L(BB\d+_\d+)", name).group(1)
        if any(i.startswith("global_load") for i in ins):
            continue
        c = loops.setdefault(key, {"valu": 0, "f64": 0, "salu": 0, "lds": 0, "br": 0})
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if "f64" in op:
                    c["f64"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                c["br"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
    for key, c in loops.items():
        print("mask %3d loop %-9s per sample: VALU %5.1f (f64 %5.1f)  SALU %4.1f  LDS %4.1f  branches %4.1f" % (
            mask, key, c["valu"] / trip, c["f64"] / trip, c["salu"] / trip, c["lds"] / trip, c["br"] / trip))

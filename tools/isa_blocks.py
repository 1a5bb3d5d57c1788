"""Per-basic-block instruction census of a gfx950 .s kernel (tools for DESIGN.md's instruction budgets)."""
import re
import sys


def census(path, minimum=1):
    blocks, name, cur = [], "entry", []
    for line in open(path):
        m = re.match(r"^(\.LBB[0-9_]+):", line) or re.match(r"^; %bb\.([0-9]+):", line)
        if m:
            blocks.append((name, cur)); name, cur = m.group(1), []
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    for name, ins in blocks:
        if len(ins) < minimum:
            continue
        c = lambda f: sum(1 for i in ins if f(i))
        print("%-12s n=%4d f64=%4d f32=%3d acc=%3d cnd=%3d ds=%3d vmem=%3d wait=%3d salu=%3d int/other-valu=%3d" % (
            name, len(ins), c(lambda i: "_f64" in i), c(lambda i: "_f32" in i and "cvt" not in i),
            c(lambda i: "accvgpr" in i), c(lambda i: "cndmask" in i), c(lambda i: i.startswith("ds_")),
            c(lambda i: i.startswith(("global_", "flat_", "buffer_", "scratch_"))), c(lambda i: i == "s_waitcnt"),
            c(lambda i: i.startswith("s_") and i != "s_waitcnt"),
            c(lambda i: i.startswith("v_") and "_f64" not in i and "accvgpr" not in i and "cndmask" not in i)))


if __name__ == "__main__":
    census(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)

"""Registers, scratch and LDS of every kernel in the built library's gfx950 code object (no GPU needed): the figures DESIGN.md
quotes, and the check that no batch kernel spills.

    python tools/kernel_resources.py [substring ...]       # --check: exit 1 if a kernel named by the substrings has scratch
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "nvspeechplayer_amd", "lib", "libspeechPlayer.so")


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [o.replace("klatt::", "").replace("(klatt::KernelArgs)", "").replace("(KernelArgs)", "") for o in out]


def kernels(obj=OBJ):
    with tempfile.TemporaryDirectory() as tmp:
        co, fat = os.path.join(tmp, "engine.co"), os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "copy")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + co, "--unbundle"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block)
        name = get("name").group(1)
        out.append(dict(name=name, agpr=int(block.split()[0]), vgpr=int(get("vgpr_count").group(1)), sgpr=int(get("sgpr_count").group(1)),
                        scratch=int(get("private_segment_fixed_size").group(1)), spill=int(get("vgpr_spill_count").group(1)),
                        lds=int(get("group_segment_fixed_size").group(1))))
    names = demangle([k["name"] for k in out])
    for k, n in zip(out, names):
        k["pretty"] = n
    return out


if __name__ == "__main__":
    subs = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    bad = 0
    print("%-86s %5s %5s %5s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spills"))
    for k in sorted(kernels(), key=lambda k: k["pretty"]):
        if subs and not any(s in k["pretty"] or s in k["name"] for s in subs):
            continue
        print("%-86s %5d %5d %5d %8d %6d" % (k["pretty"][:86], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["spill"]))
        bad += k["scratch"] > 0
    if check and bad:
        sys.exit(1)

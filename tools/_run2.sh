set -x
mkdir -p gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for N in 65536 16384; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $ROOT/gpurun_out/r2_icache_$N -- python3 $ROOT/bench.py $ARGS --utterances $N > /dev/null 2> $ROOT/gpurun_out/r2_icache_$N.err
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for n in (65536, 16384):
    f = glob.glob("gpurun_out/r2_icache_%d/**/*counter_collection.csv" % n, recursive=True)
    if not f: print("no file", n); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "klatt" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(n, k[:80])
        for c, v in sorted(cs.items()):
            print("    %-30s %.6g" % (c, sum(v) / len(v)))
PY

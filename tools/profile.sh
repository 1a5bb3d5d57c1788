#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats plus PMC passes (each in its own run, as the
# MI355X guide prescribes; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: tools/profile.sh <outdir> [bench.py args...]
# With PMC_KEY=<workload> PMC_JSON=<file> in the environment the per-launch VALU / HBM figures of this run are merged into
# that JSON (profiles/r2_pmc.json, which bench.py reads for roofline.traffic and roofline.valu) in the same go.
set -u
OUT=$(realpath -m "$1"); shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${PROF_STEPS:-20} --warmup 2 --no-cpu-baseline --no-extras $*"      # (>= 20 launches: VERDICT r3 -- five included a 9.6 ms outlier)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
PMC_ARGS="--steps 6 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_insts" -- python3 "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_insts.err"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d "$OUT/pmc_f64" -- python3 "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_f64.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_waits" -- python3 "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_waits.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_write.err"
if [ -n "${PMC_KEY:-}" ] && [ -n "${PMC_JSON:-}" ]; then
  python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$PMC_KEY" "$PMC_JSON" > "$OUT/summary.txt" 2>&1
else
  python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
fi
cat "$OUT/summary.txt"

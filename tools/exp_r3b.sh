#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_gputests.log 2>&1; tail -5 gpurun_out/r3b_gputests.log
timeout -k 10 500 python tools/ab_probe.py run base v2a +cfg2 +rot +jit +cfg4 > gpurun_out/r3b_exp.txt 2>&1
cat gpurun_out/r3b_exp.txt

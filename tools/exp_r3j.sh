#!/bin/bash
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 400 python tools/ab_probe.py run v2h v2h_e1 v2h_x4 +cfg2 +jit
} > gpurun_out/r3j.txt 2>&1
cat gpurun_out/r3j.txt

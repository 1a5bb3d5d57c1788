#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r3k.txt
cat gpurun_out/r3k.txt

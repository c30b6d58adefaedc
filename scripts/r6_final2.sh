#!/bin/bash
# final binary of round 6: the gate three times, then the evidence set (scripts/r6_final.sh)
set -u
bash scripts/gpu_gate.sh r6g4 3
for i in 1 2 3; do grep -E "graph replay|trajectory layered-bf16x2|unfiltered F32768 bf16x2 (grad y_emb|share|kink)|unfiltered F32768 bf16x3 kink|kink F2048 .* (flip rate|y_emb)" gpurun_out/r6g4/parity_report_$i.txt > gpurun_out/r6g4/bars_$i.txt; done
bash scripts/r6_final.sh

#!/bin/bash
# last call of the round: the driver's gate three times on the final binary, then the committed profile set once more (so that
# r05_bench_default.json carries the hipgraph_x8 entries and comes from the same box as the trace)
set -u
bash scripts/r5_gate.sh r5final3 3
bash scripts/profile_all.sh r05 step

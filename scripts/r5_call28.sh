#!/bin/bash
# final binary of the round: the driver's gate, then the committed profile set
set -u
bash scripts/r5_gate.sh r5final2 1
bash scripts/profile_all.sh r05 step

#!/bin/bash
# A/B of two builds of the library on ONE box, training step (BASELINE configs[4], 128 pairs): libfear_hip_prev.so against libfear_hip.so,
# alternating, three rounds.   gpurun --timeout 1200 -- bash tools/ab_train.sh
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for l in libfear_hip_prev.so libfear_hip.so; do
FEAR_LIB=feartracker_amd/$l python tools/train_prof.py 128 8 block 2>&1 | grep -E "ms/step wall" | sed "s|^|$l |" | cut -c1-150
done; done

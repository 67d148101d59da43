#!/bin/bash
# every tools/kb_c32_* binary (tools/kbench.hip built with -DFEAR_C32_ONLY and -DC32_* knobs), 256 crops, 30 launches
cd ${GRAFT_REPO_ROOT:-.}/tools
for k in kb_c32_*; do ./$k 256 30 | grep chain32 | tail -1 | sed "s|^|$k |"; done

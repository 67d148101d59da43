#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools
for k in kb_c32_*; do ./$k 256 30 | grep chain32 | tail -1 | sed "s|^|$k |"; done

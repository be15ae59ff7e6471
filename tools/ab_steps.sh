#!/bin/bash
# Same-box A/B of a training step between two builds of the library, alternated (boxes of the pool differ by +-5 %, so only
# numbers from one box compare):   tools/ab_steps.sh <reference .so> <qm9|rna|pdbbind> [steps] [alternations]
# The reference library is loaded through PAMNET_HIP_LIB (pamnet_amd/lib.py); the other run takes the in-tree build.
# Typical use: build HEAD into tools/probes/libref.so (git stash; python __graft_entry__.py; cp ...; git stash pop), then
#   gpurun -- 'bash tools/ab_steps.sh tools/probes/libref.so rna 200 3'
R=${GRAFT_REPO_ROOT:-/root/repo}
ref=$1; kind=${2:-rna}; steps=${3:-200}; alt=${4:-3}
for i in $(seq $alt); do
  echo "== reference ($ref)"; PAMNET_HIP_LIB=$(realpath $ref) python $R/tools/store_steps.py $kind $steps 2>&1 | tail -1
  echo "== in-tree build";    python $R/tools/store_steps.py $kind $steps 2>&1 | tail -1
done

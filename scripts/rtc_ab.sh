#!/bin/bash
# Same-box A/B of the run-time specialised kernels (option `specialise`) against the interpreter, with a hash of the five buffers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in N4chicken N4screw N4sixty7 N4smooth N4mandel; do
  for s in 0 1; do
    echo "== $w specialise=$s"
    timeout -k 10 300 python scripts/kbench.py $w specialise=$s hash=1 frames=${FRAMES:-48} 2>&1 | grep -v amdgpu.ids
  done
done

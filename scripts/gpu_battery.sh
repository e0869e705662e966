#!/bin/bash
# Run scripts/battery.sh on a GPU box and bring the results into profiles/<round>/ (run in the build container).
# Leaves the commit the library was built from in .build_commit (the box has no .git): the PMC files are stamped with it.
set -e
cd /root/repo
export RM_ROUND=${RM_ROUND:-r03}
(git rev-parse --short=12 HEAD; git status --porcelain -- cpu_raymarcher_amd include bench.py | grep -q . && echo "+uncommitted changes") | tr '\n' ' ' > .build_commit
python -c "import __graft_entry__ as g; g.build()" > /dev/null
hipcc_bin=/opt/rocm/bin/hipcc
mkdir -p scripts/bin && $hipcc_bin --offload-arch=gfx950 -O2 -o scripts/bin/valu_issue_bench scripts/valu_issue_bench.hip 2> /dev/null
/usr/local/graft/bin/gpurun --timeout 1150 -- "RM_ROUND=$RM_ROUND scripts/battery.sh $1 > gpurun_out/battery.log 2>&1; tail -40 gpurun_out/battery.log"
mkdir -p profiles/$RM_ROUND
for f in gpurun_out/$RM_ROUND/*; do case $f in *.err) ;; *) cp $f profiles/$RM_ROUND/ ;; esac; done
ls profiles/$RM_ROUND

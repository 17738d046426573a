#!/bin/bash
# round 3, call R: host-side profile of the training step (C1 is bound by it)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/host_profile.py STEP_METR-LA 100 > gpurun_out/r03r_host_profile_C1.log 2>&1
head -5 gpurun_out/r03r_host_profile_C1.log

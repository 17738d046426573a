#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_step.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3 > gpurun_out/r06_zi_tests.log
cat gpurun_out/r06_zi_tests.log
L=gpurun_out/r06_zi_edge_row.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; lib=$2; shift 2; STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
run "C2 old" libstep_hip_olddgl.so
run "C2 new" libstep_hip.so
run "C4 old" libstep_hip_olddgl.so --config STEP_PEMS07
run "C4 new" libstep_hip.so --config STEP_PEMS07
run "C5 old" libstep_hip_olddgl.so --config SYNTH_4096
run "C5 new" libstep_hip.so --config SYNTH_4096
done
(cd /tmp && export TMPDIR=/tmp && for lib in libstep_hip_olddgl.so libstep_hip.so; do STEP_HIP_LIB=$GRAFT_REPO_ROOT/step_amd/$lib STEP_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_zi -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 --no-prefetch --steps 10 --warmup 3 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_zi -name '*.db' | head -1) | grep "edge_bwd" | cut -c1-150 | sed "s/^/$lib /" >> $GRAFT_REPO_ROOT/$L; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_zi; done)
cat $L

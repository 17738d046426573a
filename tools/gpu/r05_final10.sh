#!/bin/bash
# round 5: the JSON line is the LAST line of stdout also when RCCL prints its banner (one rank under a process group)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 70 python bench.py --force-process-group --no-extras --no-cpu-baseline --no-pmc --steps 10 --warmup 3 > gpurun_out/r05zr_stdout.txt 2> gpurun_out/r05zr_stderr.txt
echo "rc $? lines $(wc -l < gpurun_out/r05zr_stdout.txt)"; tail -1 gpurun_out/r05zr_stdout.txt | head -c 120; echo; grep -c "RCCL version" gpurun_out/r05zr_stdout.txt gpurun_out/r05zr_stderr.txt

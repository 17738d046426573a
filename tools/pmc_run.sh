#!/bin/bash
# usage: tools/pmc_run.sh <name> <kernel-substring> -- <command...>   (GPU box); prints per-kernel PMC averages
name=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmc_${name}_a -o p -- "$@" > gpurun_out/pmc_${name}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmc_${name}_b -o p -- "$@" > gpurun_out/pmc_${name}_b.log 2>&1
python - "$name" "$kern" <<'PY'
import sqlite3, glob, sys
name, kern = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(f'gpurun_out/pmc_{name}_*/')):
    for db in glob.glob(d + '*.db'):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("""select s.kernel_name, p.name, sum(e.value), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like ? group by s.kernel_name, p.name""", (f'%{kern}%',)).fetchall()
        for r in rows:
            print(r[0][18:58], f"{r[1]:28s} per-dispatch {r[2]/r[3]:14.0f}  dispatches {r[3]}  avg_ns {r[4]:.0f}")
PY

#!/bin/bash
# round 6: counters of the second-version attention backward / forward (bench tool, T = 168 and 42)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_a2_$name -o p -- python tools/bench_pt_attention.py > gpurun_out/pmc_a2_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python - <<'PY' > gpurun_out/r06_w_attn2_pmc.txt
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/pmc_a2_*/')):
    for db in glob.glob(d + '**/*.db', recursive=True):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("""select s.kernel_name, d.grid_size_x, p.name, sum(e.value), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%attn2%' group by s.kernel_name, d.grid_size_x, p.name""").fetchall()
        for r in rows:
            print(r[0][18:50], r[1], f"{r[2]:26s} per-dispatch {r[3]/r[4]:16.0f}  n {r[4]}  avg_us {r[5]/1000:.0f}")
PY
cat gpurun_out/r06_w_attn2_pmc.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES -d gpurun_out/pmc_enc2 -o p -- python tools/bench_encoder.py 2456 336 2 > gpurun_out/pmc_enc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA -d gpurun_out/pmc_enc3 -o p -- python tools/bench_encoder.py 2456 336 2 > gpurun_out/pmc_enc3.log 2>&1
python - <<'PY'
import sqlite3, glob
for d in ['gpurun_out/pmc_enc2/', 'gpurun_out/pmc_enc3/']:
    for db in glob.glob(d + '*.db'):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("""select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%tsformer_encoder%' group by s.kernel_name, p.name""").fetchall()
        for r in rows:
            print(r[0][38:52], f"{r[1]:28s} per-dispatch {r[2]/r[3]:16.0f}")
PY

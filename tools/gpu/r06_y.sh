#!/bin/bash
# round 6: occupancy of the forecasting encoder per config (waves resident per SIMD = SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE / 8 * 1024))
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in STEP_PEMS04 STEP_PEMS07 STEP_METR-LA; do
  for park in x 1 0; do
  rm -rf gpurun_out/pmc_occ
  env $( [ $park != x ] && echo STEP_ENC_PARK=$park ) rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d gpurun_out/pmc_occ -o p -- python bench.py --pmc-child - --config $cfg --encoder-workgroups 0 > gpurun_out/pmc_occ.log 2>&1
  python - $cfg $park <<'PY' >> gpurun_out/r06_y_encoder_occupancy.txt
import sqlite3, glob, sys
for db in glob.glob('gpurun_out/pmc_occ/**/*.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("""select s.kernel_name, p.name, sum(e.value), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e
       join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%tsformer_encoder%' group by s.kernel_name, p.name""").fetchall()
    v = {r[1]: r[2] / r[3] for r in rows}
    if v:
        us = rows[0][4] / 1000
        occ = v['SQ_WAVE_CYCLES'] * 4 / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
        print(sys.argv[1], 'STEP_ENC_PARK=' + sys.argv[2], rows[0][0][18:75], f"{us:.0f} us  waves {v['SQ_WAVES']:.0f}  waves/SIMD {occ:.2f}  clock {v['GRBM_GUI_ACTIVE'] / 8 / us / 1000:.2f} GHz  VALU insts {v['SQ_INSTS_VALU'] / 1e6:.0f} M")
PY
  done
done
cat gpurun_out/r06_y_encoder_occupancy.txt

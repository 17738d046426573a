#!/bin/bash
# PMC passes over the encoder micro-benchmark (run on the GPU box via gpurun). Writes gpurun_out/pmc_enc_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_enc_$name -o p -- python tools/bench_encoder.py 2456 336 2 > gpurun_out/pmc_enc_$name.log 2>&1
}
if [ "$1" != "mem" ]; then
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
fi
run mem1 FETCH_SIZE
run mem2 WRITE_SIZE GRBM_GUI_ACTIVE
python - <<'PY'
import sqlite3, glob, json
for d in sorted(glob.glob('gpurun_out/pmc_enc_*/')):
    for db in glob.glob(d + '*.db'):
        con = sqlite3.connect(db); cur = con.cursor()
        try:
            rows = cur.execute("""select s.kernel_name, p.name, avg(e.value), count(*) from rocpd_pmc_event e
               join rocpd_info_pmc p on e.pmc_id = p.id
               join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               where s.kernel_name like '%tsformer_encoder%' group by s.kernel_name, p.name""").fetchall()
        except Exception as ex:
            print(db, "query failed", ex); continue
        for r in rows:
            print(d.split('/')[-2], r[0][20:60], r[1], r[2], r[3])
# HBM traffic record for bench.py's roofline object (dropout-on kernel = the one the training step launches)
vals = {}
for name in ("mem1", "mem2"):
    for db in glob.glob(f'gpurun_out/pmc_enc_{name}/*.db'):
        cur = sqlite3.connect(db).cursor()
        for kn, pn, v in cur.execute("""select s.kernel_name, p.name, avg(e.value) from rocpd_pmc_event e
               join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               where s.kernel_name like '%tsformer_encoder_kernel%Lb1E%' group by s.kernel_name, p.name""").fetchall():
            vals[pn] = v
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # rocprofv3 reports both in KiB-like units of 1024 B per the counter definition; FETCH_SIZE x2 on gfx950 (guide, HBM section)
    rec = {"STEP_PEMS04:B8": {"read_bytes": int(vals["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(vals["WRITE_SIZE"] * 1024),
                              "raw": vals, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_encoder.sh, S=2456 P=336 dropout on"}}
    json.dump(rec, open("gpurun_out/encoder_pmc.json", "w"), indent=1)
    print(rec)
PY

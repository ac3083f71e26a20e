#!/bin/bash
# PMC passes (counters only + kernel trace) for an arbitrary command.  Usage: bash tools/pmc_run.sh NAME KERNEL_SUBSTR cmd...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; PAT=$2; shift 2
OUT=$R/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
cd $R
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- "${CMD[@]}" > $OUT/$n.log 2>&1; }
CMD=("$@")
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
run p3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run p4 GRBM_GUI_ACTIVE
run p5 FETCH_SIZE
run p6 WRITE_SIZE
python tools/pmc_summarize.py $OUT "$PAT" > $R/gpurun_out/pmc_$NAME.txt
rm -rf $OUT
cat $R/gpurun_out/pmc_$NAME.txt

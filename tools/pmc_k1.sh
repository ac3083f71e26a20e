#!/bin/bash
# PMC passes for the isolated K1 forward kernel (bench.py --only_roofline); writes CSV under gpurun_out/pmc_k1/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_k1
mkdir -p $OUT
cd /tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python $R/bench.py --only_roofline > $OUT/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
run p3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run p4 GRBM_GUI_ACTIVE GRBM_COUNT
run p5 FETCH_SIZE
run p6 WRITE_SIZE
ls -R $OUT | head -40

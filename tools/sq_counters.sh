#!/bin/bash
# GPU box: SQ counters of every gsr:: kernel over a dozen eager forward + backward passes of the headline workload
# (bench.py --traffic-child) - or of the command given after the directory, relative to the repo root -, three rocprofv3 --pmc passes
# (no trace domains besides --kernel-trace).  usage: sq_counters.sh <outdir> [python-script args ...]
O=${1:-gpurun_out/sq}
mkdir -p $O
R=$GRAFT_REPO_ROOT
shift
if [ $# -gt 0 ]; then CMD="python $R/$*"; else CMD="python $R/bench.py --traffic-child train"; fi
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/p1 -o t -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $R/$O/p2 -o t -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $R/$O/p3 -o t -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_table.py $O/p1 $O/p2 $O/p3 > $O/table.md
cat $O/table.md
find $O -name "*.csv" -size +2M -delete

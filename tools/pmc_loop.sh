#!/bin/bash
# L2 (TCC) / L1 (TCP) counters of bk_main INSIDE the frame loop (tools/loop_clk.py), per variant library:
#   VARIANTS="main tpf0" bash tools/pmc_loop.sh <outdir>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
out=$ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp
{
echo "### TCC / TCP counters available:"; rocprofv3 -L 2>/dev/null | grep -o -E "TC[CP]_[A-Z0-9_]+" | sort -u | tr '\n' ' '; echo
for v in $VARIANTS; do
  for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_l
    echo "### $v: --pmc $pass"
    RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=$ROOT/build/variants/lib_$v.so timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_l -- python $ROOT/tools/loop_clk.py 12 2>&1 | grep "in-loop"
    python $ROOT/tools/pmc_report.py /tmp/pmc_l bk_main
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt

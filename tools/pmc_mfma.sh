#!/bin/bash
# SQ / GRBM counters of bk_main in both arithmetic modes (matrix-pipe busy cycles, wave cycles and their wait split,
# GPU-active cycles), kernel alone on the bench-shaped uniform launch: rocprofv3 --pmc in its own passes (kernel-trace only).
# -> gpurun_out/pmc_mfma/summary.txt (copied to profiles/r03_d_mfma_pmc.md with the command lines)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_mfma
mkdir -p $out
cd /tmp
{
echo "### kernel source hash (bench.source_hash): $(cd $ROOT && python -c 'import bench; print(bench.source_hash())')"
echo "### counters available (rocprofv3 -L | grep):"
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_MISC|SQ_ACTIVE_INST_LDS|SQ_ACTIVE_INST_VMEM|SQ_INSTS_VALU_MFMA_F16|SQ_INSTS_MFMA|SQ_INSTS_VALU_MFMA_MOPS_F16|SQ_INSTS_VALU|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE" | sort -u | tr '\n' ' '
echo
for prec in split f16; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc_m
    echo "### \$ RMNET_BANK_PRECISION=$prec rocprofv3 --pmc $pass --kernel-trace --output-format csv -- python tools/chunk_bench.py 8 21 36 21 36 5"
    RMNET_BANK_PRECISION=$prec timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_m -- python $ROOT/tools/chunk_bench.py 8 21 36 21 36 5 2>&1 | tail -1
    python $ROOT/tools/pmc_report.py /tmp/pmc_m bk_main
  done
done
} > $out/summary.txt 2>&1
cat $out/summary.txt

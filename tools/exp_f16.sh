# ad-hoc probe: chunk_bench + bk_clk (+ optional dump / accuracy / trace) for a list of variant libraries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{
W="${W:-8 21 36 21 36 5}"
export RMNET_BANK_PRECISION=${PREC:-f16}
for v in ${VARIANTS:-f16}; do
echo "== $v"; RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/bk_clk.py $W 2>&1 | tail -2
[ -n "$DUMP" ] && RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/bk_clk_dump.py $W 2>&1 | tail -6
done
[ "${ACC:-none}" != none ] && RMNET_HIP_LIB=build/variants/lib_${ACC}.so timeout 300 python tools/dbg_bank.py 2>&1 | tail -14
for t in ${TRACE}; do
echo "== trace $t"; RMNET_HIP_LIB=build/variants/lib_$t.so timeout 120 python tools/bk_trace.py 0.46 ${TT:-40} 2>&1 | tail -12
done
} > gpurun_out/t1/log_f16.txt 2>&1
cat gpurun_out/t1/log_f16.txt

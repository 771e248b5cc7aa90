cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{
export RMNET_BANK_PRECISION=f16
for v in $VARIANTS; do for i in 1 2 3; do
echo "== $v run $i"; RMNET_HIP_LIB=build/variants/lib_$v.so timeout 300 python tools/dbg_bank.py 2>&1 | grep -A1 "no=70\|WORST"
done; done
} > gpurun_out/t1/log_acc.txt 2>&1
cat gpurun_out/t1/log_acc.txt

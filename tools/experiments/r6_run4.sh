cd /root/repo
export TMPDIR=/tmp
export CRDR_PRECISION=bf16x6
bash tools/pmc_1x1.sh b6_d128k3 128 128 128 3 1 0 16 22
bash tools/pmc_1x1.sh b6_d256k1 256 128 128 1 1 0 16 22
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_b6_d128k3_l -- python3 tools/pmc_one.py 128 128 128 3 1 0 16 22 > /dev/null 2>&1
unset CRDR_PRECISION
python tools/pmc_summary.py b6_d128k3 77.3 269 > gpurun_out/r6_pmc_bf6_v4.txt 2>&1
python tools/pmc_summary.py b6_d256k1 17.2 403 >> gpurun_out/r6_pmc_bf6_v4.txt 2>&1
python - >> gpurun_out/r6_pmc_bf6_v4.txt 2>&1 <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_b6_d128k3_l/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "igemm_kernel" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        print("  ", r["Counter_Name"], r["Counter_Value"])
PY
cat gpurun_out/r6_pmc_bf6_v4.txt

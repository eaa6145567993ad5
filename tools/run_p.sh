export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_w1 -- python3 tools/pmc_one.py 128 128 128 3 1 0 16 -1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_w2 -- python3 tools/pmc_one.py 128 128 128 3 1 0 16 -1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
for d in ("pmc_w1","pmc_w2"):
    f = max(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows=[r for r in csv.DictReader(open(f)) if "wino_kernel" in r["Kernel_Name"]]
    ids=sorted({int(r["Dispatch_Id"]) for r in rows})[-5:]
    acc=collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]]+=float(r["Counter_Value"])/len(ids)
    for k,v in acc.items(): print(d,k,"%.4g"%v)
PY

#!/bin/bash
# counters of the fused GDN backward kernels (MFMA busy, clock, wave wait split)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gdnb_a $R/gpurun_out/pmc_gdnb_b $R/gpurun_out/pmc_gdnb_f $R/gpurun_out/pmc_gdnb_w
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_a -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_b -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_f -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_w -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(f"gpurun_out/pmc_gdnb_{d}/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("gdn_bwd_onepass_kernel", "gdn_dgamma_kernel", "gdn_fused_fwd_kernel"):
            if name in k:
                out.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v[-4:]) / len(v[-4:]) for c, v in d.items()} for k, d in out.items()}
a, b = load("a"), load("b")
f, w = load("f"), load("w")
for k in a:
    c = a[k]; cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k, "cycles %.0f  MFMA busy %.1f%%  wait_any %.0f%% wait_inst %.0f%% active %.0f%%" % (cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) * 100,
          c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] * 100, c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] * 100, c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"] * 100))
    print("   ", {n: round(v) for n, v in b.get(k, {}).items()})
    if k in f and k in w:   # HBM bytes as MI355X_MICROARCH.md prescribes: (2 * FETCH_SIZE + WRITE_SIZE) KiB on gfx950
        print("    HBM traffic %.1f MB per launch" % ((2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024 / 1e6))
PY

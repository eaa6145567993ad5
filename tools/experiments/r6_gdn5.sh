#!/bin/bash
# counters of the fused GDN backward kernels (MFMA busy, clock, wave wait split)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_a -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gdnb_b -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(f"gpurun_out/pmc_gdnb_{d}/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("gdn_fused_bwd_kernel<12, 0>", "gdn_fused_bwd_kernel<12, 1>", "gdn_fused_fwd_kernel", "wgrad_kernel"):
            if name in k:
                out.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v[-4:]) / len(v[-4:]) for c, v in d.items()} for k, d in out.items()}
a, b = load("a"), load("b")
for k in a:
    c = a[k]; cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k, "cycles %.0f  MFMA busy %.1f%%  wait_any %.0f%% wait_inst %.0f%% active %.0f%%" % (cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) * 100,
          c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] * 100, c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] * 100, c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"] * 100))
    print("   ", {n: round(v) for n, v in b.get(k, {}).items()})
PY

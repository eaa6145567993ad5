#!/bin/bash
# kernel-time decomposition of the fused GDN backward passes: builds with parts of the tile loop removed (_exp/lib_<bits>.so)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export CRDR_HIP_LIB=$R/_exp/lib_$v.so
  rm -rf $R/gpurun_out/gdnexp_$v
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gdnexp_$v -- python3 $R/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
done
cd $R
python - "$@" <<'PY'
import csv, glob, sys
for v in sys.argv[1:]:
    f = glob.glob(f"gpurun_out/gdnexp_{v}/**/*kernel_trace.csv", recursive=True)[0]
    t = {}
    for r in csv.DictReader(open(f)):
        for name in ("gdn_fused_bwd_kernel<12, 0>", "gdn_fused_bwd_kernel<12, 1>", "gdn_bwd_onepass_kernel", "gdn_fused_fwd_kernel"):
            if name in r["Kernel_Name"]:
                t.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("variant", v, {k: round(sum(x[-4:]) / len(x[-4:]), 1) for k, x in t.items()})
PY

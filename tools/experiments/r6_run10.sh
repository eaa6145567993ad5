cd /root/repo
export TMPDIR=/tmp
for o in 0 1; do
  export CRDR_W4_ORDER=$o
  bash tools/pmc_1x1.sh c192k5s2_o$o 192 128 192 5 2 0 > /dev/null 2>&1
  bash tools/pmc_1x1.sh c256k5s2_o$o 256 128 256 5 2 0 > /dev/null 2>&1
  bash tools/pmc_1x1.sh t192k5s2_o$o 192 64 192 5 2 1 > /dev/null 2>&1
  bash tools/pmc_1x1.sh d128k3_o$o 128 128 128 3 1 0 > /dev/null 2>&1
  bash tools/pmc_1x1.sh h320k5_o$o 320 16 4256 5 1 0 > /dev/null 2>&1
done
unset CRDR_W4_ORDER
{ for o in 0 1; do python3 tools/pmc_summary.py c192k5s2_o$o 120.80 255.3; python3 tools/pmc_summary.py c256k5s2_o$o 214.75 342.1; python3 tools/pmc_summary.py t192k5s2_o$o 120.8 192.0; python3 tools/pmc_summary.py d128k3_o$o 77.31 269.0; python3 tools/pmc_summary.py h320k5_o$o 278.92 211.1; done; } > gpurun_out/r6_pmc_order2.txt 2>&1
grep -v "wave cycles\|MFMA busy" gpurun_out/r6_pmc_order2.txt
for r in 1 0 1 0; do CRDR_REUSE_D_FORWARD=$r timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-140; done

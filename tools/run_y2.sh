for rep in 1 2; do
echo "== new"; timeout 300 python tools/bench_wino.py --wino-only 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-20,75-140
echo "== old"; CRDR_HIP_LIB=$PWD/crdr_amd/_lib/libcrdr_exp_OLD.so timeout 300 python tools/bench_wino.py --wino-only 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-20,75-140
done

S=dec128k3,k3_128to256,hoist4256
for v in nomfma nodma; do
  echo "=== bf16x3 $v"; CRDR_HIP_LIB=$PWD/crdr_amd/_lib_$v/libcrdr_hip.so timeout 600 python tools/sweep_conv.py --shapes $S --dump --top 3 --bf16x3 2>&1 | grep -v amdgpu.ids
done

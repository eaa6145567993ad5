S=dec128k3,enc96k3,up3T,hoist4256,charm224,k3_128to256
L=$PWD/crdr_amd/_lib_st2/libcrdr_hip.so
echo "=== bf16x3 ST=2 tap-major"; CRDR_K_CMAJOR=0 CRDR_HIP_LIB=$L timeout 600 python tools/sweep_conv.py --shapes $S --dump --top 3 --bf16x3 2>&1 | grep -v amdgpu.ids
echo "=== bf16x3 ST=2 channel-major"; CRDR_K_CMAJOR=1 CRDR_HIP_LIB=$L timeout 600 python tools/sweep_conv.py --shapes $S --dump --top 3 --bf16x3 2>&1 | grep -v amdgpu.ids
echo "=== fp32 channel-major"; CRDR_K_CMAJOR=1 CRDR_HIP_LIB=$L timeout 600 python tools/sweep_conv.py --shapes $S --top 3 2>&1 | grep -v amdgpu.ids

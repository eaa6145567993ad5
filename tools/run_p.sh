export TMPDIR=/tmp
bash tools/pmc_1x1.sh e96k3 96 128 96 3 1 0
bash tools/pmc_1x1.sh d128k3 128 128 128 3 1 0
{
python3 tools/pmc_summary.py e96k3 43.49 201.33
python3 tools/pmc_summary.py d128k3 77.31 269.0
} > gpurun_out/pmc_wino.txt 2>&1
cat gpurun_out/pmc_wino.txt

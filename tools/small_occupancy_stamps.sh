set -euo pipefail
export TMPDIR=/tmp
O=gpurun_out/pair9; mkdir -p $O
bash tools/build_stamps.sh > $O/build.txt 2>&1
{
for n in 64 100 128 200 256; do
echo "== single, $n trajectories"; SCN_SMALL_PAIRING=0 SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so timeout -k 10 120 python3 tools/small_stamps.py 400 $n 2>&1 | grep -v amdgpu.ids | tail -11 | grep -E "total|layer 2|backward of the last"
done
for n in 32 64 100 128; do
echo "== paired, $n trajectories"; SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so timeout -k 10 120 python3 tools/small_stamps.py 400 $n 2>&1 | grep -v amdgpu.ids | tail -11 | grep -E "total|layer 2|backward of the last"
done
} > $O/stamps.txt 2>&1
cat $O/stamps.txt

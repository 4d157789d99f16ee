set -euo pipefail
export TMPDIR=/tmp
O=gpurun_out/pair2; mkdir -p $O
bash tools/build_stamps.sh > $O/build.txt 2>&1
{
echo "== paired"; SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so timeout -k 10 120 python3 tools/small_stamps.py 400 100 2>&1 | grep -v amdgpu.ids | tail -12
echo "== single"; SCN_SMALL_PAIRING=0 SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so timeout -k 10 120 python3 tools/small_stamps.py 400 100 2>&1 | grep -v amdgpu.ids | tail -12
} > $O/stamps.txt 2>&1
cat $O/stamps.txt

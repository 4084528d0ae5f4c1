export TMPDIR=/tmp
T='tests/test_pce_mlp_gpu.py::test_fused_forward'
for v in "" nfb6 noil nfb6noil; do
  if [ -n "$v" ]; then export MK_LIB_OVERRIDE=$PWD/tools/ab/lib_$v.so; else unset MK_LIB_OVERRIDE; fi
  echo "== variant [$v]"; timeout -k 10 120 python -m pytest "$T" -x -q -k "True-384-768-384-424-1 or True-16-64-32 or True-73-384-384-1000" 2>&1 | grep -E "passed|failed|assert [0-9]" | head -5
done
export MK_LIB_OVERRIDE=$PWD/tools/ab/lib_stamps.so
timeout -k 10 120 python tools/mlp_stamps.py 0 > gpurun_out/r03f_stamps_fwd.txt 2>&1; sed -n 1,45p gpurun_out/r03f_stamps_fwd.txt

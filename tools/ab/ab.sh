export TMPDIR=/tmp
python tools/mlp_ablate.py 2>&1 | tail -1
for v in n4_3 n4_4 n6_4 n8_4 n12_4 n6_4_abl63; do MK_LIB_OVERRIDE=$PWD/tools/ab/lib_$v.so python tools/mlp_ablate.py 2>&1 | tail -1; done

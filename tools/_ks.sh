timeout -k 10 300 python tools/gemm_modes_check.py || exit 1
MODES=1,3,5,11 python tools/gemm_ksweep.py 8192 4096 2>&1 | grep -E "^mode"
RES=1 MODES=1,3,5,11 python tools/gemm_ksweep.py 8192 4096 2>&1 | grep -E "^mode"

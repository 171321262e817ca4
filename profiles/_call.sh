#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05x; mkdir -p $O
timeout 200 python profiles/catch_up_microbench.py > $O/mb_A.json 2> $O/mb_A.err
CLMGS_LIB_PATH=$R/clm_gs_amd/libclmgs_hip_varB.so timeout 200 python profiles/catch_up_microbench.py > $O/mb_B.json 2> $O/mb_B.err
CLMGS_LIB_PATH=$R/clm_gs_amd/libclmgs_hip_varC.so timeout 200 python profiles/catch_up_microbench.py > $O/mb_C.json 2> $O/mb_C.err
timeout 200 python profiles/catch_up_microbench.py > $O/mb_A2.json 2> $O/mb_A2.err
cat $O/mb_*.json; tail -3 $O/mb_A.err

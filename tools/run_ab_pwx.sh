mkdir -p gpurun_out/r4c
for v in r3 m3 m2 nor c8; do CIDNET_LIB_PATH=$PWD/tools/bin/libcidnet_pwx_$v.so timeout -k 10 200 python tools/micro_pwx.py > gpurun_out/r4c/pwx_$v.txt 2>&1 || exit 1; done
timeout -k 10 300 python tools/micro_pwx.py --fp32 > gpurun_out/r4c/pwx_new.txt 2>&1
tail -1 gpurun_out/r4c/pwx_*.txt

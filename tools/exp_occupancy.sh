#!/bin/bash
# Does the random-walk kernel gain from more resident waves per SIMD?  Builds three variants of the d = 2 lane
# kernels (ymc_kernels.hip, part 1) -- four waves per SIMD with the coefficients in SGPRs (95 VGPRs) or hoisted
# (128 VGPRs, 2 spills), eight waves per SIMD (64 VGPRs, 37 spills) -- next to the product's (at most two waves
# per SIMD, 158 VGPRs), and runs bench.py without helper waves at 2 / 4 / 8 waves' worth of chains per SIMD.
#   on the build host:  tools/exp_occupancy.sh build      (writes yagre-mcmc_amd/libymc_{w4s,w4h,w8s}.so)
#   on the GPU box:     tools/exp_occupancy.sh run        (prints one line per library and ensemble size)
set -e
cd "$(dirname "$0")/.."
CS=yagre-mcmc_amd/csrc
if [ "$1" = build ]; then
    make -C $CS > /dev/null
    OTHERS=$(cd $CS && ls *.o | grep -v "ymc_kernels_p1.o\|_san.o")
    for v in "4,4:-DYMC_EXP_SGPR:w4s" "4,4::w4h" "8,8:-DYMC_EXP_SGPR:w8s"; do
        IFS=: read w extra n <<< "$v"
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/opt/rocm/include \
            -DYMC_KERNELS_PART=1 $extra "-DYMC_EXP_WAVES=$w" -c $CS/ymc_kernels.hip -o /tmp/k1_$n.o
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libymc_$n.so /tmp/k1_$n.o $OTHERS -ldl )
    done
    exit 0
fi
mkdir -p gpurun_out
for K in 131072 262144 524288; do
    for lib in base w4s w4h w8s; do
        L=$PWD/yagre-mcmc_amd/libymc.so; [ $lib != base ] && L=$PWD/yagre-mcmc_amd/libymc_$lib.so
        YMC_SPLIT=0 YMC_LIBRARY=$L python bench.py --chains $K --transitions 4000 --no-cpu-baseline \
            > gpurun_out/occ_${lib}_$K.json 2> gpurun_out/occ_${lib}_$K.err
        python - gpurun_out/occ_${lib}_$K.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "steps/s %.4g" % d["value"], "kernel ms %.3f" % d["roofline"]["kernel_ms_avg"],
      "clock MHz %.0f" % d["sustained_clock_mhz"], flush=True)
PY
    done
done

#!/bin/bash
# Where the fused wide kernel's time goes at large ensembles: timing-only builds without one stage each (their results are wrong):
#   tools/exp_wide_stages.sh build     (here)          tools/exp_wide_stages.sh run   (on the GPU box)
cd "$(dirname "$0")/.." || exit 1
C=yagre-mcmc_amd/csrc
VARIANTS=${VARIANTS:-"NODRAW NOPROD NOPRIOR"}
if [ "$1" = build ]; then
  make -C $C >/dev/null || exit 1
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/opt/rocm/include -DYMC_WF_$v -c $C/ymc_wide_fused.hip -o /tmp/wf_$v.o || exit 1
    objs=$(cd $C && ls ymc_kernels_p?.o ymc_mlda2_d?.o ymc_mlda2_thin_d?.o ymc_nested.o ymc_nested_deep.o ymc_aem.o ymc_tile.o ymc_tile_aem.o ymc_tile_nested.o ymc_post.o ymc_api.o ymc_comm.o | sed "s#^#$C/#")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o yagre-mcmc_amd/libymc_exp_$v.so $objs /tmp/wf_$v.o -ldl || exit 1
  done
  exit 0
fi
echo "== full"; python tools/exp_wide_fused.py 2>&1 | grep -E "m=64 |m=512 "
for v in $VARIANTS; do
  echo "== $v"; YMC_LIBRARY=$PWD/yagre-mcmc_amd/libymc_exp_$v.so python tools/exp_wide_fused.py 2>&1 | grep -E "m=64 |m=512 "
done

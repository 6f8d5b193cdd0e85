#!/bin/bash
# One GPU-box pass: the -m gpu tests, then the bench lines of every workload and the N > 1 rehearsals.
#   tools/gpu_check.sh TAG      (output under gpurun_out/TAG_*)
set -o pipefail
TAG=${1:-chk}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1 || { tail -30 $OUT/${TAG}_pytest.log; exit 1; }
tail -3 $OUT/${TAG}_pytest.log
python bench.py > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err || { cat $OUT/${TAG}_bench_c2.err; exit 1; }
for wl in c3 c3m c3a c4 c5; do
    python bench.py --workload $wl --no-cpu-baseline > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err || { cat $OUT/${TAG}_bench_$wl.err; exit 1; }
done
python bench.py --workload iat --steps 3 --warmup 1 > $OUT/${TAG}_bench_iat.json 2> $OUT/${TAG}_bench_iat.err || { cat $OUT/${TAG}_bench_iat.err; exit 1; }
# N > 1 rehearsals on the one GPU: RCCL with a single rank; two ranks over the explicit TCP stand-in;
# two ranks WITHOUT it must fail loudly (RCCL refuses two ranks on one device)
YMC_BENCH_FORCE_COMM=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_rccl1.json 2> $OUT/${TAG}_rccl1.err || { cat $OUT/${TAG}_rccl1.err; exit 1; }
YMC_COMM=tcp YMC_BENCH_ALLOW_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 3 --warmup 1 --chains 32768 > $OUT/${TAG}_tcp2.json 2> $OUT/${TAG}_tcp2.err || { tail -20 $OUT/${TAG}_tcp2.err; exit 1; }
# the same two ranks started by bench.py itself (one plain process, no launcher environment)
env -u WORLD_SIZE -u RANK YMC_COMM=tcp YMC_BENCH_ALLOW_SHARED_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --chains 32768 --no-cpu-baseline \
    > $OUT/${TAG}_tcp2plain.json 2> $OUT/${TAG}_tcp2plain.err || { tail -20 $OUT/${TAG}_tcp2plain.err; exit 1; }
YMC_COMM_TIMEOUT=60 YMC_BENCH_ALLOW_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
    bench.py --gpus 2 --steps 3 --warmup 1 --chains 32768 > $OUT/${TAG}_rccl2_shared.json 2> $OUT/${TAG}_rccl2_shared.err
echo "two RCCL ranks on one GPU: exit code $? (non-zero expected)" | tee $OUT/${TAG}_rccl2_shared.rc
for f in c2 c3 c3m c3a c4 c5 iat rccl1 tcp2 tcp2plain; do python - $OUT/${TAG}_$( [ $f = rccl1 -o $f = tcp2 -o $f = tcp2plain ] && echo $f || echo bench_$f ).json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "region %.2fs" % d.get("timed_region_s", 0), "clock", d.get("sustained_clock_mhz"),
      r["bound"], "frac %.3f" % r["frac"], "kernel_ms", r.get("kernel_ms_avg"), "valu", (r.get("valu") or {}).get("frac"),
      "collective", d.get("collective"), d.get("nranks_seen"))
PY
done

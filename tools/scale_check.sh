#!/bin/bash
# The multi-GPU scaling lines of one node, as plain processes (bench.py starts its own ranks):
#   tools/scale_check.sh [TAG] [MAX_GPUS]
#   - c2 (BASELINE configs[1], 65 536 chains per GPU) at 1 / 2 / 4 / 8 GPUs (weak scaling),
#   - the 8-GPU totals of configs[3] and configs[4]: c4 at 16 384 and c5 at 131 072 chains per GPU.
# Prints value, collective, nranks_seen and gather_ms per line, then per rank its device, NUMA node, cores, power cap and
# the clock ITS GPU sustained in the timed region (a drop under the node's power envelope shows here first); a line whose
# collective is not "rccl" with nranks_seen == N (for N > 1) fails the script.  Needs as many GPUs as MAX_GPUS (default: all visible).
# Never run the N = 8 case on a box that is not yours to fill.
set -o pipefail
TAG=${1:-scale}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out
mkdir -p $OUT
HAVE=$(python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from yagremcmc_amd import _ffi
print(_ffi.device_count())
PY
) || exit 1
MAX=${2:-$HAVE}
echo "GPUs visible: $HAVE, scaling up to $MAX"
status=0
line() { # file, expected ranks
python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2])
print(f"{sys.argv[1].split('/')[-1]:28s} n_gpus {d['n_gpus']} value {d['value']:.4g} {d['unit']}  per-GPU {d['value'] / d['n_gpus']:.4g}  "
      f"ms/step {d['ms_per_step']:.2f}  collective {d.get('collective')}  nranks_seen {d.get('nranks_seen')}  gather_ms {d.get('gather_ms')}")
for r in d.get("ranks") or []:     # every rank's GPU: the clock it held in the timed region, its socket, its cores
    clk = r.get("sustained_clock_mhz")
    print(f"    rank {r['rank']} device {r['device']} NUMA node {r.get('numa_node')} cores {r.get('cpus')} power cap "
          f"{r.get('power_cap_w')} W  sustained clock {clk and round(clk)} MHz")
if d.get("ranks"):
    clocks = [r["sustained_clock_mhz"] for r in d["ranks"] if r.get("sustained_clock_mhz")]
    if clocks:
        print(f"    clock over the node: min {min(clocks):.0f} / max {max(clocks):.0f} MHz (one GPU alone: the n1 line's "
              f"sustained_clock_mhz)")
for name, blk in (d.get("workloads") or {}).items():
    print(f"    {name}: {blk['value']:.4g} steps/s  roofline {blk['roofline']['bound']} {blk['roofline']['frac']:.3f}")
ok = d["n_gpus"] == n and (n == 1 or (d.get("collective") == "rccl" and d.get("nranks_seen") == n))
sys.exit(0 if ok else 4)
PY
}
for n in 1 2 4 8; do
    [ $n -le $MAX ] || continue
    f=$OUT/${TAG}_c2_n$n.json
    python3 bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline > $f 2> $OUT/${TAG}_c2_n$n.err || { tail -20 $OUT/${TAG}_c2_n$n.err; status=1; continue; }
    line $f $n || status=1
done
if [ $MAX -ge 8 ]; then
    for wl in c4 c5; do
        f=$OUT/${TAG}_${wl}_n8.json
        python3 bench.py --gpus 8 --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $f 2> $OUT/${TAG}_${wl}_n8.err || { tail -20 $OUT/${TAG}_${wl}_n8.err; status=1; continue; }
        line $f 8 || status=1
    done
fi
exit $status

#!/bin/bash
# The N-rank launch paths of bench.py on a box with ONE GPU (--oversubscribe: rank r on device r mod visible), cold kernel
# cache, all three launchers, config 3 (ahead-of-time kernel) and config 5 (hiprtc): what a real `--gpus 8` does on the
# host side - 8 engines, 8 parameter set-ups, one hiprtc compile shared through the disk cache, the rendezvous - without
# the scaling (the ranks share the device, so per-rank time is ~N x the single-rank time). Output: one summary line per run.
#   tools/multi_gpu_dry_run.sh <out.txt> [ranks=8]
set -u
out=$1; N=${2:-8}
root=$PWD
summ() { python - "$1" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print("   NO JSON LINE:", open(sys.argv[1]).read()[-400:]); raise SystemExit
d = json.loads(line[-1])
rk = d["ranks"]
st = [r["startup"] for r in rk]
jit = [s["jit"] for s in st]
print(f"   launcher: {d['config']['launcher']}; ranks {len(rk)}; value {d['value']:.4g} {d['unit']}; ms_per_step {d['ms_per_step']:.2f}")
print("   per-rank ms_per_step  : " + " ".join(f"{r['ms_per_step']:.1f}" for r in rk))
print("   import_s              : " + " ".join(f"{s['import_s']:.2f}" for s in st))
print("   engine_s              : " + " ".join(f"{s['engine_s']:.2f}" for s in st))
print("   setup_to_first_launch : " + " ".join(f"{s['setup_to_first_launch_s']:.2f}" for s in st))
print("   wait_at_first_barrier : " + " ".join(f"{s['wait_at_first_barrier_s']:.2f}" for s in st))
print("   hiprtc compiles       : " + " ".join(str(j.get('compiles')) for j in jit) + "   disk hits: " + " ".join(str(j.get('disk_hits')) for j in jit) + "   memory hits: " + " ".join(str(j.get('memory_hits')) for j in jit))
PY
}
{
echo "# bench.py --gpus $N --oversubscribe on one MI355X: launch paths, cold kernel cache per run (tools/multi_gpu_dry_run.sh)"
for w in cfg3 cfg5; do
  python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 > /tmp/dry_single.json 2>/tmp/dry_single.err
  echo "== $w, 1 rank (reference)"; summ /tmp/dry_single.json
  for l in processes threads torchrun; do
    export MLGPU_CACHE_DIR=$(mktemp -d /tmp/mlgpu_cache_XXXXXX)
    t0=$(date +%s.%N)
    if [ $l = torchrun ]; then
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --oversubscribe --workload $w --no-cpu-baseline --steps 5 --warmup 2 > /tmp/dry.json 2>/tmp/dry.err
    else
      python bench.py --gpus $N --oversubscribe --launcher $l --workload $w --no-cpu-baseline --steps 5 --warmup 2 > /tmp/dry.json 2>/tmp/dry.err
    fi
    rc=$?; t1=$(date +%s.%N)
    echo "== $w, $N ranks, launcher $l: rc $rc, wall $(python -c "print(f'{$t1-$t0:.1f}')") s, cache files $(ls $MLGPU_CACHE_DIR | wc -l)"
    summ /tmp/dry.json
    [ $rc -ne 0 ] && tail -5 /tmp/dry.err
    rm -rf $MLGPU_CACHE_DIR; unset MLGPU_CACHE_DIR
  done
done
} > $out 2>&1

#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of bench.py for the given workloads.
#   tools/gpu_profile.sh <tag> cfg3 cfg4 cfg5 ...
# Writes gpurun_out/prof_<tag>_<w>/ (scratch) and gpurun_out/<tag>_<w>_{stats.csv,bench.json}; copy the ones
# to be judged into profiles/.
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
for w in "$@"; do
  out=$root/gpurun_out/prof_${tag}_$w
  rm -rf $out; mkdir -p $out
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --workload $w --no-cpu-baseline > $root/gpurun_out/${tag}_${w}_bench_under_rocprof.json 2> $out/stderr.log )
  f=$(find $out -name '*_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $root/gpurun_out/${tag}_${w}_kernel_stats.csv
  python $root/bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $root/gpurun_out/${tag}_${w}_bench.json
  echo "== $w"; head -3 $root/gpurun_out/${tag}_${w}_kernel_stats.csv; cat $root/gpurun_out/${tag}_${w}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
  find $out -name '*.csv' -size +2M -delete
done

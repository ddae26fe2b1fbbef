cd $GRAFT_REPO_ROOT
root=$PWD
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_events.py -x -q -m gpu 2>&1 | tail -3
prof() { # tag, env..., 
  tag=$1; shift
  out=$root/gpurun_out/r05/prof_$tag; rm -rf $out; mkdir -p $out
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --workload synthfused --steps 5 --warmup 2 > $out/bench.json 2> $out/stderr.log )
  f=$(find $out -name '*_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $root/gpurun_out/r05/${tag}_kernel_stats.csv && head -4 $f
  python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('block ms (under rocprof)', d['roofline']['kernel_ms'], 'wall/launch', d['ms_per_step']/d['config']['launches_per_step'])"
  find $out -name '*.csv' -size +2M -delete
}
prof sf_graph MLGPU_BENCH_MIXDOWN=graph
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g voice-samples/s  %.3f ms per block (wall %.3f ms per step of %d)' % (d['value'], r['kernel_ms'], d['ms_per_step'], d['config']['launches_per_step']))"; }
echo "## synthfused sum=graph: $(MLGPU_BENCH_MIXDOWN=graph timeout 300 python bench.py --no-cpu-baseline --workload synthfused 2>/dev/null | tail -1 | line)"

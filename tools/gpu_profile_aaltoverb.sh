set -u
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles_aalto; rm -rf $out; mkdir -p $out
s=/tmp/prof_aalto; rm -rf $s; mkdir -p $s
cmd="python $root/tools/aaltoverb_bench.py 65536 16 0"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $s/stats -- $cmd > $out/r01w_aaltoverb_bench_under_rocprof.txt 2> $s/stats.log )
f=$(find $s/stats -name '*_kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/aaltoverb_bench.py 65536 16 0   (the reference's examples/audio-and-midi/reverb.cpp through the shim, 65536 reverbs x 16 DSPVectors per launch, MI355X)"; [ -n "$f" ] && head -5 $f; } > $out/r01w_aaltoverb_kernel_stats.csv
for pass in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $s/pmc_$pass -- $cmd > /dev/null 2> $s/pmc_$pass.log )
done
python $root/tools/pmc_summary.py traffic $s/pmc_FETCH_SIZE $s/pmc_WRITE_SIZE $out/r01w_aaltoverb_traffic.json
$cmd > $out/r01w_aaltoverb_bench.txt
cat $out/r01w_aaltoverb_kernel_stats.csv | cut -c1-180; cat $out/r01w_aaltoverb_bench.txt; python -c "
import json; d=json.load(open('$out/r01w_aaltoverb_traffic.json'))
for k,v in d['kernels'].items():
    if 'graph' in k: print(k, v)"

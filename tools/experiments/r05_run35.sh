cd $GRAFT_REPO_ROOT
for f in fused sequence fused sequence; do
MLGPU_RT_FORM=$f timeout 300 python bench.py --workload rt --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['rt']; print('$f', {k:round(r[k],1) if isinstance(r[k],float) else r[k] for k in ('call_us_p50','call_us_p99','call_us_max','misses','output_peak')})"
done

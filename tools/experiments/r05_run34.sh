cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_processbuffer.py tests/test_gpu_sequence.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload rt --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['rt']; print({k:r[k] for k in ('form','call_us_p50','call_us_p99','call_us_max','misses','two_calls_us_p50','voice_kernel_us_free_running','output_peak')})"

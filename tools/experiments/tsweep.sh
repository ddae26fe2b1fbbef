cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], r['kernel_ms'], r['frac'], r.get('store_ceiling') or r.get('stream_ceiling'))"; }
for tl in "30 25" "50 15" "75 10" "150 5" "250 3" "30 25"; do set -- $tl; $B --workload cfg3 --vectors $1 --launches $2 2>/dev/null | tail -1 | p "cfg3 T=$1 L=$2"; done
for tl in "32 16" "64 8" "128 4"; do set -- $tl; $B --workload cfg4 --vectors $1 --launches $2 2>/dev/null | tail -1 | p "cfg4 T=$1 L=$2"; done
for tl in "16 0" "32 0" "64 0"; do set -- $tl; $B --workload cfg5 --vectors $1 2>/dev/null | tail -1 | p "cfg5 T=$1"; done

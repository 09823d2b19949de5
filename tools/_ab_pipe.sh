set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
root=$(pwd)
for pipe in 0 1 0 1; do
  d=$(mktemp -d /tmp/kst_XXXX)
  (cd /tmp && DES_E2_PIPE=$pipe rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 $root/bench.py --cpu-steps 0 --no-large-series --no-elide-compare --no-profile --no-ceiling > $root/gpurun_out/r04_j_ab_pipe${pipe}.json 2> /dev/null)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "DES_E2_PIPE=$pipe: $(python3 -c "
import csv,json,sys
rows=list(csv.DictReader(open('$f')))
k={r['Name'].split('(')[0].replace('void des_hip::','')[:40]:(float(r['AverageNs'])/1e3,int(r['Calls'])) for r in rows[:5]}
print({a:round(b[0],2) for a,b in k.items()}, json.loads(open('$root/gpurun_out/r04_j_ab_pipe${pipe}.json').read().splitlines()[-1])['ms_per_step'])
")"
  cp "$f" $root/gpurun_out/r04_j_kernel_stats_pipe${pipe}.csv
done
DES_PROFILE_OUT=gpurun_out DES_TRAFFIC_NAME=r04_j_pmc_traffic_2d.json python tools/measure_traffic.py --ndims 2 > gpurun_out/r04_j_pmc_traffic_2d.txt 2>&1
cat gpurun_out/r04_j_pmc_traffic_2d.txt

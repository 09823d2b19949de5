#!/bin/bash
# tools/collect_round_profiles.sh TAG -- the measurement record of a round, on the MI355X box (gpurun): the benchmark lines (3-D
# default, the driver's command, 2-D), rocprofv3 kernel statistics of both, PMC traffic of both, SQ counters of the 3-D step.
# Everything goes to gpurun_out/TAG_*; copy what is to be judged into profiles/.
set -e
tag=${1:-r05_zz}
out=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2> $out/${tag}_bench_driver_cmd.err
python bench.py --ndims 2 > $out/${tag}_bench_2d.json 2> $out/${tag}_bench_2d.err
echo "bench lines done"
root=$(pwd)
for nd in 3 2; do
  d=$(mktemp -d /tmp/kst_XXXX)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 $root/bench.py --ndims $nd --cpu-steps 0 --no-large-series --no-elide-compare --no-profile --no-ceiling > /dev/null 2> $root/$out/${tag}_kernel_stats_${nd}d.err)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${tag}_kernel_stats_${nd}d_200steps.csv
  echo "kernel stats ${nd}d done"
done
DES_PROFILE_OUT=$out DES_TRAFFIC_NAME=r05_pmc_traffic.json python tools/measure_traffic.py > $out/${tag}_pmc_traffic.txt 2>&1
echo "pmc 3d done"
DES_PROFILE_OUT=$out DES_TRAFFIC_NAME=r05_pmc_traffic_2d.json python tools/measure_traffic.py --ndims 2 > $out/${tag}_pmc_traffic_2d.txt 2>&1
echo "pmc 2d done"
python tools/pmc_passes.py $out/${tag}_pmc_counters_1M.txt "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum" > $out/${tag}_pmc_counters.log 2>&1
echo "sq counters done"

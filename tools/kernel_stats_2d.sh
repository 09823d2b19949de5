#!/bin/bash
# tools/kernel_stats_2d.sh TAG -- rocprofv3 kernel statistics of the 2-D benchmark (200 steps) into gpurun_out/TAG_kernel_stats_2d.csv,
# the benchmark line of the same run into gpurun_out/TAG_ks2d.json.  The environment passes through: an A/B of two libraries or two
# settings of a switch on ONE box is `bash tools/kernel_stats_2d.sh a && DES_HIP_LIB=$PWD/other.so bash tools/kernel_stats_2d.sh b`
# (boxes differ by 2-3 %: compare within one call only).
set -e
export TMPDIR=/tmp
tag=${1:-ks}
root=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/kst_XXXX)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 $root/bench.py --ndims 2 --cpu-steps 0 --no-large-series --no-elide-compare --no-profile --no-ceiling > $root/gpurun_out/${tag}_ks2d.json 2> $root/gpurun_out/${tag}_ks2d.err)
f=$(find $d -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats_2d.csv

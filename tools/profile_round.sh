#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Everything the judged numbers come from, into gpurun_out/<tag>_*: rocprofv3 --kernel-trace --stats of bench.py (config 2 and
# config 4), then the --pmc passes of tools/profile_pmc.sh (separate runs) summarised by tools/pmc_summary.py.
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in 2 4 5; do
    steps=20; [ $cfg = 4 ] && steps=4; [ $cfg = 5 ] && steps=2
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_c$cfg -- python3 $R/bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-secondary \
        > $R/gpurun_out/${tag}_bench_under_rocprof_c$cfg.json 2> $R/gpurun_out/${tag}_stats_c$cfg.err
    f=$(find $R/gpurun_out/${tag}_stats_c$cfg -name '*kernel_stats.csv' | head -1)
    cp "$f" $R/gpurun_out/${tag}_kernel_stats_c$cfg.csv
    echo "stats config $cfg done"
done
cd $R
tools/profile_pmc.sh ${tag}c2 --config 2
echo "pmc config 2 done"
tools/profile_pmc.sh ${tag}c4 --config 4
echo "pmc config 4 done"
tools/profile_pmc.sh ${tag}c5 --config 5
echo "pmc config 5 done"
python3 tools/pmc_summary.py --source "$tag: bench.py --config 2 --steps 3 (1024 x 65536, L0=256), rocprofv3 --pmc, separate passes" gpurun_out/pmc_${tag}c2_* > gpurun_out/${tag}_pmc_summary_c2.json
python3 tools/pmc_summary.py --source "$tag: bench.py --config 4 --steps 3 (1024 x 65536, 2 levels, 17 taps), rocprofv3 --pmc, separate passes" gpurun_out/pmc_${tag}c4_* > gpurun_out/${tag}_pmc_summary_c4.json
python3 tools/pmc_summary.py --source "$tag: bench.py --config 5 --steps 3 (128 x 65536, 3 levels), rocprofv3 --pmc, separate passes" gpurun_out/pmc_${tag}c5_* > gpurun_out/${tag}_pmc_summary_c5.json
# the raw counter CSVs are large: keep only the summaries and the kernel stats
rm -rf gpurun_out/pmc_${tag}c2_* gpurun_out/pmc_${tag}c4_* gpurun_out/pmc_${tag}c5_* gpurun_out/${tag}_stats_c2 gpurun_out/${tag}_stats_c4 gpurun_out/${tag}_stats_c5

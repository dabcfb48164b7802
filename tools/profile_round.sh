#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Everything the judged numbers come from, into gpurun_out/<tag>_*: rocprofv3 --kernel-trace --stats of bench.py (configs 2, 4, 5 and the
# hierarchical configs with the reference's default method, LoCOMP), then the --pmc passes of tools/profile_pmc.sh (separate runs)
# summarised by tools/pmc_summary.py, which stamps every summary with the sha256 of the kernel sources (tools/csrc_digest.py).
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in 2 4 5 4locomp 5locomp; do
    n=${cfg%locomp}; extra=""; [ "$n" != "$cfg" ] && extra="--method locomp"
    steps=20; [ $n = 4 ] && steps=4; [ $n = 5 ] && steps=3; [ $cfg = 5locomp ] && steps=2
    name=c$n; [ "$n" != "$cfg" ] && name=c${n}_locomp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_$name -- python3 $R/bench.py --config $n $extra --steps $steps --warmup 2 --no-cpu-baseline --no-secondary \
        > $R/gpurun_out/${tag}_bench_under_rocprof_$name.json 2> $R/gpurun_out/${tag}_stats_$name.err
    f=$(find $R/gpurun_out/${tag}_stats_$name -name '*kernel_stats.csv' | head -1)
    cp "$f" $R/gpurun_out/${tag}_kernel_stats_$name.csv
    t=$(find $R/gpurun_out/${tag}_stats_$name -name '*kernel_trace.csv' | head -1)
    # per-launch durations of the initial correlation in launch order (the same kernel on the same shape in configs 2 and 4)
    python3 - "$t" > $R/gpurun_out/${tag}_corr_init_launches_$name.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'corr_init_mfma_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('corr_init_mfma_kernel launches in order: duration ms, gap to the previous kernel end ms')
allk = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ends = {}
prev_end = None
for r in allk:
    if 'corr_init_mfma_kernel' in r['Kernel_Name']:
        ends[r['Dispatch_Id']] = prev_end
    prev_end = int(r['End_Timestamp'])
for r in rows:
    pe = ends.get(r['Dispatch_Id'])
    print('%.3f  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, '%.3f' % ((int(r['Start_Timestamp']) - pe) / 1e6) if pe else '-'))
PY
    echo "stats $name done"
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
rm -rf gpurun_out/pmc_${tag}c2_* gpurun_out/pmc_${tag}c4_* gpurun_out/pmc_${tag}c5_* gpurun_out/${tag}_stats_c2 gpurun_out/${tag}_stats_c4 gpurun_out/${tag}_stats_c5 gpurun_out/${tag}_stats_c4_locomp gpurun_out/${tag}_stats_c5_locomp

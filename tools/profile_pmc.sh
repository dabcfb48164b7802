#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_pmc.sh <tag> [bench.py args...]
# Separate rocprofv3 --pmc passes (counters only: never combined with the sys / runtime trace domains) over a short
# bench.py run; the per-kernel CSVs land in gpurun_out/pmc_<tag>_<pass>/ and are summarised by tools/parse_pmc2.py.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-transfers --no-secondary --profile-steps 1 "${BENCH_ARGS[@]}" > $R/gpurun_out/pmc_${tag}_$name.log 2>&1
}
BENCH_ARGS=("${@}")
run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run busy SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE

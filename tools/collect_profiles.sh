#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box through gpurun; writes gpurun_out/prof_*).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh bf16'
# Counters go in their own passes (never combined with trace domains); summaries are reduced on the box.
set -u
#   [r5] other configurations: bash tools/collect_profiles.sh bf16 _c4 --depth 101 --frames 16 --clips 16   (tag + extra bench.py flags; SQ pass skipped with SKIP_SQ=1)
DT=${1:-bf16}
TAG=${2:-}
shift; shift
X="$*"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$DT$TAG
mkdir -p $O
DT0=$DT
cd /tmp && export TMPDIR=/tmp
STEPS=6; [ "$DT" = f32 ] && STEPS=4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$DT$TAG -- python $R/bench.py --dtype $DT $X --steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs > $O/stats.log 2>&1
cp $(find /tmp/ks_$DT$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kn_$DT$TAG -- python $R/bench.py --dtype $DT $X --steps $STEPS --warmup 2 --no-cpu-baseline --no-other-configs --no-overlap > $O/stats_nooverlap.log 2>&1
cp $(find /tmp/kn_$DT$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats_nooverlap.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$DT$TAG -- python $R/bench.py --dtype $DT $X --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$DT$TAG -- python $R/bench.py --dtype $DT $X --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/write.log 2>&1
python $R/tools/pmc_summary.py /tmp/pf_$DT$TAG /tmp/pw_$DT$TAG > $O/pmc_summary.json 2> $O/pmc_summary.err
[ "${SKIP_SQ:-0}" = 1 ] || rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/sq_$DT$TAG -- python $R/bench.py --dtype $DT $X --no-cpu-baseline --no-other-configs --no-overlap --steps 1 --warmup 1 > $O/sq.log 2>&1
[ "${SKIP_SQ:-0}" = 1 ] || python $R/tools/sq_summary.py $O/sq_counters.json $(find /tmp/sq_$DT$TAG -name "*.db") > $O/sq_summary.txt 2>&1
tail -1 $O/stats.log | cut -c1-200

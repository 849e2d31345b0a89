#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the rocprofv3 summaries that profiles/ holds.
#   bash tools/collect_profiles.sh <round-tag>      -> gpurun_out/profiles/<tag>_*
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (batch mode, exact dot, torch_cpu ties)
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.log 2>&1
grep "kvc::\|^\"Name" /tmp/kt/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_batch.csv
# 1b. the same launches alone (tools/prof_driver.py makes nothing but the batched call): the per-kernel AVERAGE here is the
#     duration of one 32-layer launch (the file above mixes them with bench.py's per-call warm-up launches)
rm -rf /tmp/kt1b && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1b -- python3 $R/tools/prof_driver.py c2 torch_cpu 10 batch > /dev/null 2>&1
grep "kvc::\|^\"Name" /tmp/kt1b/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_batch_launches_only.csv
# 2. same, per-layer calls on one stream (what a strictly sequential caller sees)
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --mode calls --streams 1 --no-graph > /dev/null 2>&1
grep "kvc::\|^\"Name" /tmp/kt2/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_calls_single_stream.csv
# 3. canonical-tie mode and the mfma16 tolerance mode
rm -rf /tmp/kt3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --tie-mode canonical > /dev/null 2>&1
grep "kvc::\|^\"Name" /tmp/kt3/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_batch_canonical_ties.csv
rm -rf /tmp/kt4 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --dot-mode mfma16 > /dev/null 2>&1
grep "kvc::\|^\"Name" /tmp/kt4/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_batch_mfma16.csv
# 4. PMC passes (one counter group per run; batch launch = 32 layers)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/pmc_$n && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -- python3 $R/tools/prof_driver.py c2 torch_cpu 4 batch > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$n $OUT/${TAG}_pmc_batch_$n.csv > /dev/null
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -- python3 $R/tools/prof_driver.py c2 torch_cpu 6 calls > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcs_$c $OUT/${TAG}_pmc_$c.csv > /dev/null
done
ls -la $OUT

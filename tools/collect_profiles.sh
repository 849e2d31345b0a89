#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the rocprofv3 summaries that profiles/ holds.
#   bash tools/collect_profiles.sh <round-tag>      -> gpurun_out/profiles/<tag>_*
# Kernel-trace passes and PMC passes are separate runs (never combined with other trace domains).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <name> <prof_driver args...>: per-kernel averages of the compression launches alone
  local name=$1; shift
  rm -rf /tmp/kt_$name && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -- python3 $R/tools/prof_driver.py "$@" > /dev/null 2>&1
  grep "kvc::\|^\"Name" /tmp/kt_$name/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_$name.csv
  echo "== $name"; cat $OUT/${TAG}_kernel_stats_$name.csv
}
# 1. the default bench command under the profiler (its own summary line is kept too)
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.log 2>&1
grep "kvc::\|^\"Name" /tmp/kt/*/*_kernel_stats.csv > $OUT/${TAG}_kernel_stats_c2_bench_command.csv
# 2. the batched launches alone, per configuration (average = one 32-layer launch)
stats c2_batch_launches_only c2 torch_cpu 10 batch
stats c2_batch_canonical_ties c2 canonical 10 batch
stats c2_w32_batch c2_w32 torch_cpu 6 batch
stats c4_batch c4 torch_cpu 6 batch
stats c5_batch c5 torch_cpu 3 batch
stats c3_h2o c3 torch_cpu 2 calls
stats c2_calls_single_stream c2 torch_cpu 2 calls
# 3. PMC passes (one counter group per run) for c2 batch, c2_w32 batch and c3
for cfgname in c2 c2_w32 c3; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d" " -f1)
    suffix=$([ $cfgname = c2 ] && echo "" || echo "${cfgname}_")
    rm -rf /tmp/pmc_$n && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -- python3 $R/tools/prof_driver.py $cfgname torch_cpu 3 batch > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py /tmp/pmc_$n $OUT/${TAG}_pmc_batch_${suffix}$n.csv > /dev/null
  done
done
# 4. round 3: decode step (N1 / N3 / ThinK) and the N2 prototype under the kernel trace, the one-wave probes behind WaveHeapL
if [ "${2:-all}" != "nopmc" ]; then :; fi
rm -rf /tmp/kt_dec && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_dec -- python3 $R/tools/decode_timing.py > $OUT/${TAG}_decode_timing.log 2>&1
grep "kvc::\|^\"Name\|attention\|fmha\|Cijk\|cat\|index" /tmp/kt_dec/*/*_kernel_stats.csv | head -40 > $OUT/${TAG}_kernel_stats_decode_timing.csv
rm -rf /tmp/kt_n2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_n2 -- python3 $R/tools/n2_probe.py > $OUT/${TAG}_n2_probe.log 2>&1
head -25 /tmp/kt_n2/*/*_kernel_stats.csv > $OUT/${TAG}_n2_kernel_stats.csv
(cd $R/tools && for p in hop_probe step_probe; do [ -x ./$p ] && ./$p > $OUT/${TAG}_$p.txt 2>&1; done; [ -x ./heap_probe ] && ./heap_probe 7992 120 260 > $OUT/${TAG}_heap_probe.txt 2>&1)
# 5. round 3: the matrix-core probes behind h2o_fused_kernel (16x16x4 f32 accumulation order; sustained f32 MFMA rate), its phase shares
(cd $R/tools && for p in mfma16x4_probe mfma_peak_probe; do [ -x ./$p ] && ./$p > $OUT/${TAG}_$p.txt 2>&1; done)
[ -f $R/tools/diag/libkvc_hip_stamps.so ] && KVC_LIB_PATH=$R/tools/diag/libkvc_hip_stamps.so python3 $R/tools/h2o_fused_stamps.py > $OUT/${TAG}_h2o_fused_stamps.txt 2>&1
(cd $R && python3 bench.py --config c3 --steps 5 --warmup 2 > $OUT/${TAG}_bench_c3_exact.json 2> /dev/null; python3 bench.py --steps 30 --warmup 5 > $OUT/${TAG}_bench_c2.json 2> /dev/null)
ls -la $OUT

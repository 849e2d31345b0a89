#!/bin/bash
# Diagnostic builds of the library (never shipped, never loaded by the product or the tests):
#   tools/build_diag.sh noearlyclobber   -> tools/diag/libkvc_hip_noearlyclobber.so  (-DKVC_DIAG_NO_EARLYCLOBBER: round 1's hazard,
#                                            for tools/h2o_d16_stress.py)
#   tools/build_diag.sh stamps           -> tools/diag/libkvc_hip_stamps.so          (-DKVC_STAMPS: s_memtime phase stamps)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
  noearlyclobber) FLAG=-DKVC_DIAG_NO_EARLYCLOBBER ;;
  stamps) FLAG=-DKVC_STAMPS ;;
  stamps_exp1) FLAG="-DKVC_STAMPS -DKVC_FZ_EXP1" ;;
  *) echo "usage: $0 noearlyclobber|stamps"; exit 2 ;;
esac
B=$(mktemp -d)
cd "$R/kvcache_factory_amd/csrc"
FLAGS=$(make -s --eval 'pf:
	@echo $(HIPFLAGS)' pf)
for f in kvc_api kvc_score kvc_select kvc_select_exact kvc_gather kvc_h2o kvc_decode kvc_ragged kvc_l2norm kvc_merge kvc_think kvc_cam; do
  /opt/rocm/bin/hipcc $FLAGS $FLAG -c $f.hip -o $B/$f.o &
done
wait
mkdir -p "$R/tools/diag"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$R/tools/diag/libkvc_hip_$1.so" $B/*.o
rm -rf $B
ls -la "$R/tools/diag/libkvc_hip_$1.so"

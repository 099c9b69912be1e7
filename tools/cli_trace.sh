#!/bin/bash
# Where the SW command line spends its time on a config-4-sized file: the CLI run against the tuning build of
# the library (LD_LIBRARY_PATH beats the RUNPATH) with AGX_TRACE_CREATE / AGX_TRACE_CLI.   usage: tools/cli_trace.sh [pairs]
N=${1:-1048576}
mkdir -p /tmp/tl && ln -sf $PWD/accelerating-genomics_amd/libagx_tuning.so /tmp/tl/libagx.so
python3 - <<PY
import sys; sys.path.insert(0, "$PWD")
import accelerating_genomics_amd.synth as synth
synth.write_sw_file("/tmp/sw_big.in", synth.sw_pairs($N, 32, 512, seed=4))
PY
for rep in 1 2 3; do
  T0=$(date +%s.%N)
  env LD_LIBRARY_PATH=/tmp/tl AGX_TRACE_CREATE=1 AGX_TRACE_CLI=1 accelerating-genomics_amd/bin/antidiagonalSmithWaterman /tmp/sw_big.in 2>&1 >/dev/null | sed 's/^/  /'
  python3 -c "import time; print('wall %.3f s' % (time.time() - $T0))"; echo ---
done
for rep in 1 2; do
  T0=$(date +%s.%N)
  env AGX_TRACE_CLI=1 accelerating-genomics_amd/bin/antidiagonalSmithWaterman /tmp/sw_big.in 2>&1 >/dev/null | sed 's/^/  /'
  python3 -c "import time; print('wall %.3f s (shipped library)' % (time.time() - $T0))"
done

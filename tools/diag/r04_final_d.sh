#!/bin/bash
# r04 final (after the descriptor K/V fetch): build() + smoke() in one process, the whole GPU suite once, the measurement
# artefacts with the final library
O=${OUT_ROOT:-gpurun_out}/r04h; mkdir -p $O
timeout 600 python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build+smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
SUITE_DIR=r04h bash tools/diag/r04_suite.sh 1 final
bash tools/collect_profiles_r04.sh all

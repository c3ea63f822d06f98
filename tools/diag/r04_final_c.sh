#!/bin/bash
# r04 final: one more run of the whole GPU suite on the final tree, then the measurement artefacts with the final library
SUITE_DIR=r04f bash tools/diag/r04_suite.sh 1 final
bash tools/collect_profiles_r04.sh all

#!/bin/bash
# Counter passes over the gray K2 / fused kernels of tools/k2_variants_bench.py for one library.
# usage: tools/pmc_fused.sh <tag> <library.so>       (results: gpurun_out/pmc_<tag>_<pass>/, summary on stdout)
tag=$1; export OGDEWARP_LIB=$2
run() {
  p=$1; shift
  bash tools/pmc_py.sh ${tag}_$p "$@" -- tools/k2_variants_bench.py 8 "gray 6000x8000, cell 25, linear" || return 1
  python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_$p MapNodes
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU &&
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT &&
run c SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE GRBM_COUNT

#!/bin/bash
# VALU / SALU / LDS instruction counts of the tile kernel for the ablation builds (development aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in NOQUERY NOSCAN NOFINISH; do
  export PC_AMD_LIB=scratch/libpc_abl_$v.so
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d gpurun_out/pmc_abl_$v -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > gpurun_out/pmc_abl_$v.log 2>&1 || exit 1
  echo $v; python3 scratch/tools_pmc.py gpurun_out/pmc_abl_$v | grep k_tile_scan
done

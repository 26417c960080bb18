#!/bin/bash
# SQ counters of the tile kernel on the headline workload (one pass per counter group)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmc_sq1 -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > gpurun_out/pmc_sq1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD -d gpurun_out/pmc_sq2 -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pipeline > gpurun_out/pmc_sq2.log 2>&1 &&
python3 scratch/tools_pmc.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2

#!/bin/bash
# ablation builds of the library for scratch/pmc_lds.sh (staging only / + per-batch set-up / + filter and insertion)
cd "$(dirname "$0")/../point_cloud_processing_amd/csrc" || exit 1
mkdir -p ../../ablbuild
for v in ${ABL_VARIANTS:-NOQUERY NOSCAN NOFINISH}; do
  flag=-DABL_$v
  case $v in STAGE1) flag=-DABL_STAGE=1;; STAGE2) flag=-DABL_STAGE=2;; STAGE3) flag=-DABL_STAGE=3;; esac
  make -j8 EXTRA=$flag BUILD=/tmp/build_$v OUT=../../ablbuild/libpc_abl_$v.so 2>&1 | grep -E "error|Error"
done
ls -la ../../ablbuild

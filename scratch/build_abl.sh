#!/bin/bash
# ablation builds of the library for scratch/pmc_lds.sh (staging only / + per-batch set-up / + filter and insertion)
cd "$(dirname "$0")/../point_cloud_processing_amd/csrc" || exit 1
mkdir -p ../../ablbuild
for v in NOQUERY NOSCAN NOFINISH; do
  make -j8 EXTRA=-DABL_$v BUILD=/tmp/build_$v OUT=../../ablbuild/libpc_abl_$v.so 2>&1 | grep -E "error|Error"
done
ls -la ../../ablbuild

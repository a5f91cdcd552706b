#!/usr/bin/env bash
set -uo pipefail
bash tools/prof_step.sh r5g_on KVQ_FUSE_DROPRES=1 > /dev/null 2>&1; head -16 gpurun_out/r5g_on/breakdown.txt
bash tools/prof_step.sh r5g_off KVQ_FUSE_DROPRES=0 > /dev/null 2>&1; head -16 gpurun_out/r5g_off/breakdown.txt
bash tools/r5_stamps.sh > /dev/null 2>&1; tail -5 gpurun_out/r5s/stamps.txt

#!/bin/bash
# Profiles BASELINE config 4a (mixed lengths) on the GPU box: kernel trace + one PMC pass, then all configs.
#   gpurun -- 'bash tools/profile_config4a.sh'   -> gpurun_out/prof4a, gpurun_out/pmc4a, gpurun_out/configs_new.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4a -o c4a -- python3 $R/tools/run_config4a.py > $R/gpurun_out/prof4a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc4a -o c4a -- python3 $R/tools/run_config4a.py > $R/gpurun_out/pmc4a.log 2>&1
cd $R && timeout -k 10 600 python tests/tools/bench_configs.py > gpurun_out/configs_new.jsonl 2> gpurun_out/configs_new.err

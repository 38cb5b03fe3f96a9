#!/bin/bash
# The round-3 measurement passes, in one call on a GPU box (run from the repository root; outputs under gpurun_out/,
# merged into profiles/ afterwards by profiles/tools/r03_collect.py).  Every rocprofv3 line starts python3 itself.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1 ($(date +%T))"; }
step "default bench";      (cd $R && python3 bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err) || exit 1
step "kernel stats f16x3"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_main -- python3 $R/bench.py --steps 5 --warmup 1 --main-only > $O/r03_bench_f16x3.json 2> /dev/null || exit 1
step "kernel stats f32";   rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_main32 -- python3 $R/bench.py --steps 3 --warmup 1 --main-only --mode f32 > $O/r03_bench_f32.json 2> /dev/null || exit 1
step "pmc fetch";          rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > /dev/null 2>&1 || exit 1
step "pmc write";          rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > /dev/null 2>&1 || exit 1
step "pmc sq";             rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/r03_pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --main-only > /dev/null 2>&1 || exit 1
step "train stats";        rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_train -- python3 $R/profiles/tools/time_train.py f16 6 > $O/r03_time_train.txt 2> /dev/null || exit 1
step "train pmc fetch";    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_pmc_tfetch -- python3 $R/profiles/tools/time_train.py f16 3 > /dev/null 2>&1 || exit 1
step "train pmc write";    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_pmc_twrite -- python3 $R/profiles/tools/time_train.py f16 3 > /dev/null 2>&1 || exit 1
step "train pmc sq";       rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/r03_pmc_tsq -- python3 $R/profiles/tools/time_train.py f16 3 > /dev/null 2>&1 || exit 1
step done

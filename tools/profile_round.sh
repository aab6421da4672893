#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, HBM-traffic PMC passes.
# usage: bash tools/profile_round.sh r01      -> gpurun_out/...; then `python tools/summarize_profiles.py r01` condenses into profiles/
set -e
R=${1:-r01}
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${R}_train.json 2> gpurun_out/bench_${R}_train.err
python bench.py --mode eval --steps 20 --warmup 5 > gpurun_out/bench_eval.json 2> gpurun_out/bench_eval.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_train -- python bench.py --no-cpu-baseline --no-host-probe > gpurun_out/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_eval -- python bench.py --mode eval --no-cpu-baseline --no-host-probe > gpurun_out/prof_eval.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${R}_fetch -- python bench.py --no-cpu-baseline --no-host-probe --steps 4 --warmup 2 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${R}_write -- python bench.py --no-cpu-baseline --no-host-probe --steps 4 --warmup 2 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${R}_eval_fetch -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --steps 4 --warmup 2 > gpurun_out/pmc_efetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${R}_eval_write -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --steps 4 --warmup 2 > gpurun_out/pmc_ewrite.log 2>&1
echo profile set done

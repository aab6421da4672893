#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats, HBM-traffic PMC passes, MFMA-busy PMC pass.
# usage: bash tools/profile_round.sh r03      -> gpurun_out/...; then `python tools/summarize_profiles.py r03` condenses into profiles/
# (rocprofv3: the program itself directly after `--`; counters in their own passes, kernel-trace / stats only)
R=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 50 --warmup 10 > gpurun_out/bench_${R}_train.json 2> gpurun_out/bench_${R}_train.err
python bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${R}_eval.json 2> gpurun_out/bench_${R}_eval.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-eval-leg --emulate-world 8 > gpurun_out/bench_${R}_emu.json 2> gpurun_out/bench_${R}_emu.err
MHR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 8 --no-cpu-baseline --no-eval-leg --no-kernel-events --no-host-probe > gpurun_out/bench_${R}_dp2_gloo.json 2> gpurun_out/bench_${R}_dp2_gloo.err
MHR_FORCE_DP=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-eval-leg --no-kernel-events > gpurun_out/bench_${R}_dp1_rccl.json 2> gpurun_out/bench_${R}_dp1_rccl.err
echo bench lines done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_train -- python bench.py --no-cpu-baseline --no-host-probe --no-eval-leg --no-kernel-events --steps 20 --warmup 5 > gpurun_out/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_eval -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --no-kernel-events --steps 20 --warmup 5 > gpurun_out/prof_eval.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${R}_fetch -- python bench.py --no-cpu-baseline --no-host-probe --no-eval-leg --no-kernel-events --no-graph --steps 4 --warmup 4 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${R}_write -- python bench.py --no-cpu-baseline --no-host-probe --no-eval-leg --no-kernel-events --no-graph --steps 4 --warmup 4 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${R}_eval_fetch -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --no-kernel-events --steps 4 --warmup 2 > gpurun_out/pmc_efetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${R}_eval_write -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --no-kernel-events --steps 4 --warmup 2 > gpurun_out/pmc_ewrite.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${R}_mfma -- python bench.py --no-cpu-baseline --no-host-probe --no-eval-leg --no-kernel-events --no-graph --steps 4 --warmup 4 > gpurun_out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${R}_eval_mfma -- python bench.py --mode eval --no-cpu-baseline --no-host-probe --no-kernel-events --steps 4 --warmup 2 > gpurun_out/pmc_emfma.log 2>&1
echo profile set done

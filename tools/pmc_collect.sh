#!/bin/bash
# three separate counter passes (gpurun refuses --pmc combined with tracing); run from the repo root on the GPU box
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcC_FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcC_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcC_WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcC_w.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcC_MFMA -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcC_m.log 2>&1
echo mfma done

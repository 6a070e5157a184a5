#!/bin/bash
# three separate counter passes (gpurun refuses --pmc combined with tracing); run from the repo root on the GPU box
# usage: bash tools/pmc_collect.sh [compress|decompress] [output tag]
set -e
MODE=${1:-compress}
TAG=${2:-pmcC}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--mode $MODE --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
SGIC_BENCH_SHAPES=gpurun_out/${TAG}_shapes.json rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_FETCH_SIZE -- python3 bench.py $ARGS > gpurun_out/${TAG}_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_WRITE_SIZE -- python3 bench.py $ARGS > gpurun_out/${TAG}_w.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_MFMA -- python3 bench.py $ARGS > gpurun_out/${TAG}_m.log 2>&1
echo mfma done

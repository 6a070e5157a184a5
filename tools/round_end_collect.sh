#!/bin/bash
# Round-end evidence on one MI355X box (run from the repo root): the driver-format bench line, kernel traces of the compress and
# decompress steps, the three PMC passes of each + the per-shape join, attention counters, single-image latency, the 512^2 stress,
# the CLI rates (host / GPU JPEG decode) and the GPU test suite.  Everything lands in gpurun_out/r3F; copy what is cited into profiles/.
#   usage: bash tools/round_end_collect.sh [part ...]      parts: bench trace pmc attn misc tests (default: all)
PARTS=${@:-bench trace pmc attn misc tests}
O=gpurun_out/r3F
mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1; fi
if has trace; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_c -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/kt_c.json 2> $GRAFT_REPO_ROOT/$O/kt_c.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_d -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --mode decompress --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/kt_d.json 2> $GRAFT_REPO_ROOT/$O/kt_d.err || exit 1
  cd $GRAFT_REPO_ROOT
fi
if has pmc; then
  bash tools/pmc_collect.sh compress pmc3C || exit 1
  python tools/pmc_by_shape.py gpurun_out/pmc3C gpurun_out/pmc3C_shapes.json $O/round3_pmc_gemm_by_shape.json $O/round3_pmc_gemm_summary.json > $O/pmcC_summary.log 2>&1 || exit 1
  bash tools/pmc_collect.sh decompress pmc3D || exit 1
  python tools/pmc_by_shape.py gpurun_out/pmc3D gpurun_out/pmc3D_shapes.json $O/round3_pmc_decompress_gemm_by_shape.json $O/round3_pmc_decompress_gemm_summary.json unpack12_kernel > $O/pmcD_summary.log 2>&1 || exit 1
fi
if has attn; then
  bash tools/attn_pmc.sh || exit 1
  python tools/pmc_attn_summarise.py gpurun_out/pmcA1 gpurun_out/pmcA2 $O/round3_pmc_attn_s3_summary.json > $O/attn_summary.log 2>&1
fi
if has misc; then
  python tools/latency_b1.py > $O/latency_b1.log 2>&1
  python tools/stress_512.py > $O/stress_512.log 2>&1
  SGIC_GPU_JPEG=0 python tools/cli_throughput.py 1920 > $O/cli_host.log 2> $O/cli_host.err
  SGIC_GPU_JPEG=1 python tools/cli_throughput.py 1920 > $O/cli_gpu.log 2> $O/cli_gpu.err
  python tools/bench_jpeg.py > $O/bench_jpeg.log 2>&1
fi
if has tests; then python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; echo "pytest rc $?" > $O/gpu_tests_rc.txt; fi
tail -2 $O/latency_b1.log; tail -3 $O/stress_512.log; tail -1 $O/cli_host.log | cut -c1-200; tail -1 $O/cli_gpu.log | cut -c1-200; head -14 $O/pmcC_summary.log; cat $O/gpu_tests_rc.txt; tail -3 $O/gpu_tests.log

#!/usr/bin/env bash
# collect_profiles.sh -- the exact rocprofv3 commands behind profiles/rNN_* (run on an MI355X box, from the repo root).
# Counters are collected in their own passes (TCC has 4 slots: FETCH_SIZE needs 3, WRITE_SIZE 2), never together with
# --kernel-trace/--stats; the program after `--` is python3 itself (no env/bash -c hop).
#   usage: tools/collect_profiles.sh <out_dir> [round_tag]
set -euo pipefail
OUT=${1:?output directory}
TAG=${2:-r03}
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras"
# 1. per-kernel time of the default bench command (the `roofline.avg_launch_ms` of bench.py must agree with AverageNs)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 200 --warmup 20 --repeats 3 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/trace.err"
# 2. HBM traffic, one TCC counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps 5 --warmup 2 --repeats 1 > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps 5 --warmup 2 --repeats 1 > /dev/null 2> "$OUT/pmc_write.err"
# 3. instruction mix and clocks
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
          --output-format csv -d "$OUT/pmc_sq" -- $BENCH --steps 5 --warmup 2 --repeats 1 > /dev/null 2> "$OUT/pmc_sq.err"
# 4. config D diagnostic (staged evaluator)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_d" -- $BENCH --tree effects --voices 128 --partials 1024 --steps 100 --warmup 10 --repeats 3 \
          > "$OUT/${TAG}_configD_bench_under_rocprof.json" 2> "$OUT/trace_d.err"
# 4b. short blocks: the short-call kernel under the tracer (T = 64)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_short" -- $BENCH --frames 64 --steps 400 --warmup 40 --repeats 3 \
          > "$OUT/${TAG}_short64_bench_under_rocprof.json" 2> "$OUT/trace_short.err"
# 4c. the HBM-bound variant: voices whose w / amp are track rows (tools/track_bench.py), T = 4800: kernel time, then HBM bytes
TRACKS="python3 $R/tools/track_bench.py --frames 4800"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_tracks" -- $TRACKS --steps 40 --json > "$OUT/${TAG}_tracks_under_rocprof.txt" 2> "$OUT/trace_tracks.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_tracks" -- $TRACKS --steps 4 > /dev/null 2> "$OUT/pmc_fetch_tracks.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_tracks" -- $TRACKS --steps 4 > /dev/null 2> "$OUT/pmc_write_tracks.err"
# 5. micro-benchmarks (built by: hipcc --offload-arch=gfx950 -O3 ... see the headers of the .hip files)
[ -x "$R/tools/valu_rate" ] && "$R/tools/valu_rate" > "$OUT/${TAG}_valu_rate.txt" 2>&1 || true
[ -x "$R/tools/bank_bench" ] && "$R/tools/bank_bench" 64 12 4800 15 > "$OUT/${TAG}_bank_variants_4800.txt" 2>&1 || true
echo "done: $OUT"

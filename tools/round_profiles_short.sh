#!/bin/bash
# The round's measurement set without the PMC passes (tools/round_profiles.sh has them): bench lines, rocprofv3 kernel
# statistics of the default command and of the unstructured workload, the other elements / model, the two-rank gloo
# rehearsal of the self-launching bench.   usage: tools/round_profiles_short.sh <tag>   -> gpurun_out/profiles_<tag>/
set -u
TAG=$1
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOTDIR"
O=$ROOTDIR/gpurun_out/profiles_$TAG
mkdir -p "$O"
timeout -k 10 600 python3 bench.py > "$O/bench_n66_default.json" 2> "$O/bench_n66_default.err" || { echo "default bench failed"; tail -5 "$O/bench_n66_default.err"; exit 1; }
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/rocprof" -- python3 "$ROOTDIR/bench.py" --cpu-sample 0 --no-off-lattice > "$O/bench_n66_under_rocprof.json" 2> "$O/rocprof.err" ) || { echo "rocprof run failed"; tail -5 "$O/rocprof.err"; exit 1; }
find "$O/rocprof" -name "*kernel_stats.csv" -exec cp {} "$O/rocprof_kernel_stats_n66_default.csv" \;
echo "rocprof done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/rocprof_tetgen" -- python3 "$ROOTDIR/bench.py" --mesh tetgen > "$O/bench_tetgen_under_rocprof.json" 2> "$O/rocprof_tetgen.err" ) || { echo "rocprof tetgen run failed"; tail -5 "$O/rocprof_tetgen.err"; exit 1; }
find "$O/rocprof_tetgen" -name "*kernel_stats.csv" -exec cp {} "$O/rocprof_kernel_stats_tetgen_10m.csv" \;
echo "rocprof tetgen done"
timeout -k 10 300 python3 bench.py --quadratic --n 24 --cpu-sample 0 --no-newton > "$O/bench_tet10_n24.json" 2> "$O/bench_tet10.err" || echo "tet10 bench failed"
timeout -k 10 300 python3 bench.py --hex --n 40 --cpu-sample 0 --no-newton > "$O/bench_hex8_n40.json" 2> "$O/bench_hex8.err" || echo "hex8 bench failed"
timeout -k 10 300 python3 bench.py --model a5 --cpu-sample 0 --no-newton --no-tet10 --no-off-lattice > "$O/bench_a5_n66.json" 2> "$O/bench_a5.err" || echo "a5 bench failed"
FEAHIP_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --n 31 --steps 10 --warmup 3 --cpu-sample 0 > "$O/bench_gloo2_selflaunch.json" 2> "$O/bench_gloo2_selflaunch.err"; echo "gloo2 rc=$?" >> "$O/bench_gloo2_selflaunch.err"
rm -rf "$O/rocprof" "$O/rocprof_tetgen"
echo "all done"

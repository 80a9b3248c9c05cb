#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ for the current build.  Run ON THE GPU BOX from the repo root:
#   gpurun -- 'bash scripts/collect_profiles.sh r02a'
# Kernel-trace stats and each PMC group are separate runs (rocprofv3 cannot mix them; FETCH_SIZE and WRITE_SIZE do
# not fit one pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  The program itself follows `--` (python3 bench.py),
# never a wrapper: the profiler's preloaded library has initialised the GPU by then.
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
B="--no-extras --cpu-seconds 0 --frames-per-step 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 3 $B > "$out/bench_trace.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 bench.py --steps 6 --warmup 2 $B > "$out/bench_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 bench.py --steps 6 --warmup 2 $B > "$out/bench_write.log" 2>&1
echo "write done"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/tcc" -- python3 bench.py --steps 6 --warmup 2 $B > "$out/bench_tcc.log" 2>&1
echo "tcc done"
# the L2's memory-side read requests by size: 32 x n32 + 64 x n64 + 128 x n128 are the bytes read, whatever the access pattern (FETCH_SIZE
# tallies this chip's 128-byte requests at 64 bytes: MI355X_MICROARCH.md "HBM"; scripts/calib_fetch.hip holds both against a known byte count)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$out/rdreq" -- python3 bench.py --steps 6 --warmup 2 $B > "$out/bench_rdreq.log" 2>&1
echo "rdreq done"
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/calib_fetch scripts/calib_fetch.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/calib_fetch" -- /tmp/calib_fetch > "$out/calib_fetch.log" 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$out/calib_rdreq" -- /tmp/calib_fetch > "$out/calib_rdreq.log" 2>&1
echo "calibration done"
# cfg5 (outdoor shape, the scatter form of the update): kernel trace and the same counters for 40 frames of scripts/outdoor_bench.py
O="scripts/outdoor_bench.py --frames 40"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cfg5_trace" -- python3 $O > "$out/cfg5_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/cfg5_fetch" -- python3 $O > "$out/cfg5_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/cfg5_write" -- python3 $O > "$out/cfg5_write.log" 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/cfg5_tcc" -- python3 $O > "$out/cfg5_tcc.log" 2>&1
# (round 5) ... and its reads by request size, as for cfg2: FETCH_SIZE alone only brackets them on this chip
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$out/cfg5_rdreq" -- python3 $O > "$out/cfg5_rdreq.log" 2>&1
echo "cfg5 done"
# ... and the same trajectory kept as its DAG only (CHAD_FLAG_DAG_ONLY: the closing table straight to cluster words, the fold on the device)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cfg5dag_trace" -- python3 scripts/cfg5_dagonly.py --frames 110 > "$out/cfg5dag_trace.log" 2>&1
echo "cfg5 dag-only done"
# ... and a frame-sharded map's SPARSE exchange at cfg5's scale (one process, a silent second rank: scripts/exchange_trace.py)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cfg5xchg_trace" -- python3 scripts/exchange_trace.py --frames 110 > "$out/cfg5xchg_trace.log" 2>&1
echo "cfg5 exchange done"
python3 scripts/summarise_profiles.py "$out" "$tag"

#!/bin/bash
# HBM traffic of the bench's kernels from rocprofv3 TCC counters (run on the GPU
# box: gpurun -- scripts/collect_traffic.sh <tag> [latest] [bench args...]).  FETCH_SIZE and
# WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"), so
# they are collected in two separate --pmc runs with --kernel-trace only.
# Output: gpurun_out/traffic_<tag>/{fetch,write}/..., summary -> profiles/traffic_<tag>.json
# (and profiles/traffic_latest.json when the second argument is "latest").
set -e
tag=${1:-r02}; shift || true
latest=""; if [ "$1" = "latest" ]; then latest=latest; shift; fi
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/traffic_$tag
mkdir -p $out/fetch $out/write
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/fetch/bench.json 2> $out/fetch/err.log
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/write/bench.json 2> $out/write/err.log
python3 scripts/summarize_traffic.py $out $tag $latest

#!/bin/bash
# Collects the profiles kept under profiles/ (run on the GPU box from the repository root).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 bench.py > $O/bench.json
timeout -k 10 200 python3 bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats2.log
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1_under_rocprof.json 2> $O/stats1.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2> $O/fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2> $O/write.log
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2> $O/sq.log
# condensed on the box: the raw traces are too large to bring back
S=$O/summary && mkdir -p $S
cp $O/bench.json $S/r01_bench.json
cp $O/bench_streams1.json $S/r01_bench_streams1.json
cp $O/bench_under_rocprof.json $S/r01_bench_under_rocprof.json
cp $O/bench_streams1_under_rocprof.json $S/r01_bench_streams1_under_rocprof.json
python3 profiles/summarize.py stats $(find $O/stats2 -name "*kernel_stats.csv") $S/r01_kernel_stats.csv
python3 profiles/summarize.py stats $(find $O/stats1 -name "*kernel_stats.csv") $S/r01_kernel_stats_streams1.csv
python3 profiles/summarize.py pmc $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") 792 $S/r01_pmc_traffic.json
python3 profiles/summarize.py sq $(find $O/sq -name "*counter_collection.csv") $S/r01_pmc_sq.json
rm -rf $O/stats1 $O/stats2 $O/fetch $O/write $O/sq
ls -la $S

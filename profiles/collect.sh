#!/bin/bash
# Collects the profiles kept in this directory.  Run on the GPU box from the repository root:
#   bash profiles/collect.sh r03 [quick]
# One rocprofv3 run per kind of data (kernel trace; FETCH_SIZE; WRITE_SIZE; two passes of SQ counters): counters
# are never combined with traces, and every pass has a time limit of its own.
set -e
R=${1:-r03}
MODE=${2:-full}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O/summary
S=$O/summary
B="--kfdb-frames 0 --frames-4k 0 --fed-frames 0 --no-cpu-baseline"
if [ "$MODE" = full ]; then
  timeout -k 10 300 python3 bench.py > $S/${R}_bench.json
  timeout -k 10 200 python3 bench.py --streams 1 --no-cpu-baseline > $S/${R}_bench_streams1.json
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 bench.py $B > $S/${R}_bench_under_rocprof.json 2> $O/stats2.log
  python3 profiles/summarize.py stats $(find $O/stats2 -name "*kernel_stats.csv") $S/${R}_kernel_stats.csv
  rm -rf $O/stats2
fi
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --streams 1 $B > $S/${R}_bench_streams1_under_rocprof.json 2> $O/stats1.log
python3 profiles/summarize.py stats $(find $O/stats1 -name "*kernel_stats.csv") $S/${R}_kernel_stats_streams1.csv
rm -rf $O/stats1
P="--steps 2 --warmup 1 --streams 1 $B"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py $P > /dev/null 2> $O/fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py $P > /dev/null 2> $O/write.log
python3 profiles/summarize.py pmc $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") 792 $S/${R}_pmc_traffic.json
rm -rf $O/fetch $O/write
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/sq -- python3 bench.py $P > /dev/null 2> $O/sq.log
python3 profiles/summarize.py sq $(find $O/sq -name "*counter_collection.csv") $S/${R}_pmc_sq.json
rm -rf $O/sq
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- python3 bench.py $P > /dev/null 2> $O/sq2.log
python3 profiles/summarize.py sq $(find $O/sq2 -name "*counter_collection.csv") $S/${R}_pmc_sq_busy.json
rm -rf $O/sq2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --output-format csv -d $O/sq3 -- python3 bench.py $P > /dev/null 2> $O/sq3.log
python3 profiles/summarize.py sq $(find $O/sq3 -name "*counter_collection.csv") $S/${R}_pmc_sq_mem.json
rm -rf $O/sq3
ls -la $S

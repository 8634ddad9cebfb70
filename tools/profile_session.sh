#!/bin/bash
# rocprofv3 evidence for profiles/: tools/profile_session.sh <tag> [bench.py flags ...]
#   <tag>_bench_under_rocprofv3.json   the line bench.py printed under --kernel-trace --stats
#   gpurun_out/prof_<tag>/             kernel trace + stats        (tools/summarize_prof.py)
#   gpurun_out/pmc_<tag>_<group>/      one PMC pass per counter group (tools/summarize_pmc.py);
#                                      counters never share a run with --kernel-trace/--stats
cd "$(dirname "$0")/.." || exit 1
tag=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o run -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $O/${tag}_bench_under_rocprofv3.json 2> $O/prof_$tag.err
echo "trace run exit $?"; cat $O/${tag}_bench_under_rocprofv3.json | head -c 600; echo
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" \
           "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/pmc_${tag}_g$i -o run -- python3 $R/bench.py --steps 5 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $O/pmc_${tag}_g$i.err
  echo "pmc group $i ($grp) exit $?"
done
cd $R
python tools/summarize_prof.py $O/prof_$tag $O/${tag} | tail -8
LAUNCH_OFFSETS=$(python -c "import json; print(json.load(open('$O/${tag}_bench_under_rocprofv3.json'))['roofline']['launch_offsets'])") \
  python tools/summarize_pmc.py $O/${tag}_pmc.json $O/pmc_${tag}_g* | tail -40

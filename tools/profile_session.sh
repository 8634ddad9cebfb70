#!/bin/bash
# rocprofv3 evidence for profiles/: tools/profile_session.sh <tag> [bench.py flags ...]
# Raw rocprofv3 output stays in /tmp on the GPU box; gpurun_out/ receives only the summaries:
#   <tag>_bench_under_rocprofv3.json   the line bench.py printed under --kernel-trace --stats
#   <tag>_kernel_stats.csv, <tag>_dispatches.csv          (tools/summarize_prof.py)
#   <tag>_pmc.json                      one PMC pass per counter group (tools/summarize_pmc.py);
#                                       counters never share a run with --kernel-trace / --stats
cd "$(dirname "$0")/.." || exit 1
tag=$1; shift
R=$PWD; O=$R/gpurun_out; W=/tmp/prof_$tag; mkdir -p $O $W
export TMPDIR=/tmp
cd /tmp
# 1000 timed steps: the pre-roll and warm-up launches (the clock governor's ramp) are then < 30 % of the dispatches the
# --stats average is taken over, and <tag>_dispatches.csv keeps every one of them for a steady-state average
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o run -- python3 $R/bench.py --steps ${PROF_STEPS:-1000} --no-cpu-baseline --no-extras "$@" > $O/${tag}_bench_under_rocprofv3.json 2> $W/trace.err
echo "trace run exit $?"; head -c 400 $O/${tag}_bench_under_rocprofv3.json; echo
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" $EXTRA_PMC_GROUPS; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $W/pmc_g$i -o run -- python3 $R/bench.py --steps 5 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $W/pmc_g$i.err
  echo "pmc group $i ($grp) exit $?"
done
cd $R
python tools/summarize_prof.py $W/trace $O/${tag} | tail -6
LAUNCH_OFFSETS=$(python -c "import json; print(json.load(open('$O/${tag}_bench_under_rocprofv3.json'))['roofline']['launch_offsets'])") \
  python tools/summarize_pmc.py $O/${tag}_pmc.json $W/pmc_g* | tail -30
du -sh $O

#!/bin/bash
# Collect the per-round profile evidence on the GPU box:  scripts/profile_round.sh r01
#   gpurun_out/<tag>_bench.json                 bench.py JSON line (live hipEvent roofline numbers)
#   gpurun_out/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_kernel_trace_summary.txt   per (kernel, grid) averages from that trace (streams concurrent)
#   gpurun_out/<tag>_kernel_trace_serial.txt    same with CHORE_ENC_SERIAL=1 (one stream: clean per-kernel durations,
#                                               what bench.py's per-kernel numbers correspond to)
#   gpurun_out/<tag>_timeline.txt               span / busy / idle of an encoder pass (scripts/timeline.py)
#   gpurun_out/<tag>_pmc_pass<i>.txt            counter passes (separate runs, kernel-trace only)
tag=${1:-r02}
dt=${2:-fp16x3}          # precision mode: fp16x3 (bench default), bf16, fp32
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp
rm -rf /tmp/prof_$tag /tmp/profs_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- \
    python $repo/bench.py --mode query --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/prof_summary.py $f 60 > $out/${tag}_kernel_trace_summary.txt
[ -n "$f" ] && python $repo/scripts/timeline.py $f x > $out/${tag}_timeline.txt
CHORE_ENC_SERIAL=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/profs_$tag -o $tag --output-format csv -- \
    python $repo/scripts/enc_only.py $dt 8 > /dev/null 2>> $out/${tag}_rocprof.err
f=$(find /tmp/profs_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/prof_summary.py $f 60 > $out/${tag}_kernel_trace_serial.txt
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rm -rf /tmp/pmc_${tag}_$i
    CHORE_ENC_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pmc_${tag}_$i -o p --output-format csv -- \
        python $repo/scripts/enc_only.py $dt > /dev/null 2> $out/${tag}_pmc$i.err
    f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $repo/scripts/pmc_summary.py $f > $out/${tag}_pmc_pass$i.txt
done
cd $repo
# HBM-side traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE) of every kernel -> what bench.py's roofline.traffic reads
[ -f $out/${tag}_pmc_pass1.txt ] && [ -f $out/${tag}_pmc_pass2.txt ] && python scripts/pmc_traffic.py $out/${tag}_pmc_pass1.txt $out/${tag}_pmc_pass2.txt $out/pmc_traffic_${dt}.json > $out/${tag}_pmc_traffic_summary.txt

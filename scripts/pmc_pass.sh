#!/bin/bash
# one rocprofv3 counter pass over scripts/enc_only.py:  scripts/pmc_pass.sh <outfile> <counter> [<counter> ...]
out=$1; shift
repo=$(pwd)
export TMPDIR=/tmp
d=/tmp/pmc_$$
cd /tmp
rocprofv3 --kernel-trace --pmc "$@" -d $d -o p --output-format csv -- python $repo/scripts/enc_only.py bf16 2 > /dev/null 2> $d.err
f=$(find $d -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/pmc_summary.py $f conv_lds > $repo/$out || tail -5 $d.err

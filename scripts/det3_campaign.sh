#!/bin/bash
# scripts/det3_campaign.sh <runs> <label> [ENV=VAL ...]: how often, and in which backward operator first, do two concurrent
# processes' training passes differ from their own first pass (scripts/train_determinism3.py with TRACE_OPS=1)
runs=$1; label=$2; shift 2
tot=0; declare -A first
for k in $(seq $runs); do
  out=$(env TRACE_OPS=1 "$@" timeout 300 python scripts/train_determinism3.py 2 3 2>&1)
  n=$(echo "$out" | grep -c "tensors differ")
  tot=$((tot + n))
  for op in $(echo "$out" | grep "first differing backward" | sed 's/.*call #[0-9]* of [0-9]*: \([A-Za-z_]*\);.*/\1/'); do first[$op]=$(( ${first[$op]:-0} + 1 )); done
done
echo "$label: differing passes $tot of $((runs * 2 * 3)); first differing operator: $(for k in "${!first[@]}"; do echo -n "$k=${first[$k]} "; done)"

#!/bin/bash
# The table behind DESIGN 7.0 "the reference's regression configuration": test/regression_tests.jl's model, un-augmented
# (8 + 0, a normalised density with a known optimum: NLL >= 8 h(Beta(2,4)) = -2.90) and as written (8 + 8), trained with the
# reference's loop under each reading of the two third-party ingredients that are not in /root/reference -- the Lion rule
# (Optimisers.jl) and the Dense initialisation (Lux) -- and with Adam.  Runs on the GPU box:
#     bash tools/training_ablation.sh [epochs] [a|b|ab]   ->  gpurun_out/train_abl/*.json
set -u
E=${1:-300}
PART=${2:-ab}
OUT=gpurun_out/train_abl
mkdir -p $OUT
A=("8 0 lion glorot" "8 0 lion_optimisers glorot" "8 0 adam glorot" "8 0 lion lux_v1" "8 0 lion_optimisers lux_v1" "1 0 lion glorot")
B=("8 8 lion glorot" "8 8 lion_optimisers glorot" "8 8 adam glorot" "1 0 lion_optimisers glorot" "1 1 lion glorot" "1 0 adam glorot")
SPECS=()
[[ $PART == *a* ]] && SPECS+=("${A[@]}")
[[ $PART == *b* ]] && SPECS+=("${B[@]}")
for spec in "${SPECS[@]}"; do
  set -- $spec
  name=nv$1_na$2_$3_$4
  timeout -k 10 400 python tools/regression_example.py --nvars $1 --naugs $2 --opt $3 --init $4 --epochs $E --out $OUT/$name.json > $OUT/$name.log 2>&1
  echo "$name rc $?: $(tail -n 2 $OUT/$name.log | head -n 1 | cut -c1-300)"
done

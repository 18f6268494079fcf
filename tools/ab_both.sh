# usage: bash tools/ab_both.sh name... : frame stages + training loop with each variant library
for v in "$@"; do STAGES="${STAGES:-nt_encode_bwd nt_encode_fwd}" bash tools/ab_variants.sh $v | tail -1; bash tools/ab_train.sh $v | tail -1; done

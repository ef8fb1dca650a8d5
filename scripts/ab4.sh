export EESEN_OVERLAP=0
echo "--- L2_LOCAL=0"; bash scripts/ab_variants.sh "--steps 8 --warmup 2" b8
export EESEN_L2_LOCAL=1
echo "--- L2_LOCAL=1"; bash scripts/ab_variants.sh "--steps 8 --warmup 2" b4 b8

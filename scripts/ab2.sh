export EESEN_OVERLAP=0
bash scripts/ab_variants.sh "--steps 8 --warmup 2" f16 f18 f20 f22 f24
unset EESEN_OVERLAP
echo "--- overlap on (default)"
bash scripts/ab_variants.sh "--steps 10 --warmup 3" f20

# new default (wave-7 delayed poll) against the previous protocol, other configurations
for cfg in cfg4 cfg1; do
  echo "--- $cfg"
  bash scripts/ab_variants.sh "--config $cfg --steps 5 --warmup 2" old
done

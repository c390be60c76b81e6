# A/B of library builds on the same GPU box: tools/abtest.sh "<lib> <lib> ..."; per-lib caps via env CAPS_<i> = "cap4 cap8 cap22"
i=0
for rep in 1 2; do for L in ${1:-ab/libairlift_base.so}; do
  echo "$L:"; AIRLIFT_LIB=$PWD/$L PAIRS=2000000 bash tools/dbgrun.sh | grep total
done; done

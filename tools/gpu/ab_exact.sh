# interleaved A/B of library builds on one box with the exact kernels on the driver's pool: ab_exact.sh <reps> lib...
# (per library and repetition: the pool's second call in ms, then the median of seven isolated 4096-target batches)
reps=$1; shift
for rep in $(seq "$reps"); do for lib in "$@"; do
  PIK_LIB=$(realpath $lib) python tools/gpu/exact_only.py exact --single | tail -2 | tr '\n' ' ' | sed "s|^|$(basename $lib) |"; echo
done; done | sort

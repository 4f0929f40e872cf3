# interleaved A/B of library builds on one box with the exact kernels on the driver's pool: ab_exact.sh <reps> lib...
reps=$1; shift
for rep in $(seq "$reps"); do for lib in "$@"; do
  echo "$(basename $lib) $(PIK_LIB=$(realpath $lib) python tools/gpu/exact_only.py exact | tail -1)"
done; done | sort

# the exact kernels on the driver's pool under other schedules, final library of round 5
set -u
run() { echo "== $*"; for r in 1 2; do env "$@" python tools/gpu/exact_only.py exact 2>&1 | tail -1; done; }
run PIK_NOTHING=1
run PIK_PASSES=2,4,6,8,12,16,24,32,40,48,64,80
run PIK_PASSES=2,4,8,12,16,20,24,32,40,48,64,80
run PIK_PASSES=2,4,8,12,16,20,24,28,32,40,48,56,64,72,80
run PIK_PASSES=1,2,3,4,6,8,12,16,24,32,40,48,64,80
run PIK_PASSES=2,4,8,12,16,24,32,40,48,56,64,72,80,90
run PIK_OCC2=640
run PIK_OCC2=960
run PIK_OCC2=1280
run PIK_NOTHING=2

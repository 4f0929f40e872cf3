# the exact kernels on the driver's pool under other schedules (tools/gpu/exact_only.py prints ms and solves/s)
set -u
run() { echo "== $*"; env "$@" python tools/gpu/exact_only.py exact 2>&1 | tail -1; }
run PIK_NOTHING=1
run PIK_OCC2=128
run PIK_OCC2=320
run PIK_OCC2=480
run PIK_OCC2=0
run PIK_PASSES=2,4,8,12,16,24,32,40,48,64,80
run PIK_PASSES=2,4,8,16,24,32,48,64,80
run PIK_PASSES=1,2,4,8,12,16,20,24,32,40,48,64,80
run PIK_PASSES=2,4,6,8,12,16,24,32,40,56,72,88
run PIK_PASSES=3,6,10,16,24,32,40,48,64,80
run PIK_PASSES=2,4,8,12,16,20,24,28,32,40,48,56,64,72,80,90
run PIK_LPE_SCHED=0:1,12:2,20:4,32:8,64:16
run PIK_LPE_SCHED=0:1,16:4,40:8,80:16
run PIK_LPE_SCHED=0:1,16:2,24:4,40:16
run PIK_NOTHING=2

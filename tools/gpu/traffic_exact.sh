# HBM traffic of the exact kernels on the driver's pool (tools/gpu/exact_only.py: a warm-up pool and a timed pool,
# 163 840 problems) per library: FETCH_SIZE and WRITE_SIZE in their own PMC passes.  usage: traffic_exact.sh lib...
set -u
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  L=$(realpath $REPO/$lib); T=/tmp/traffic_$(basename $lib .so); rm -rf $T; mkdir -p $T
  PIK_LIB=$L rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $T/f -o p -- python $REPO/tools/gpu/exact_only.py exact > $T/f.log 2>&1
  PIK_LIB=$L rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $T/w -o p -- python $REPO/tools/gpu/exact_only.py exact > $T/w.log 2>&1
  python - $T $(basename $lib) <<'PY'
import glob, sqlite3, sys
root, name = sys.argv[1], sys.argv[2]
tot = {}
for db in glob.glob(root + "/*/*.db") + glob.glob(root + "/*/*/*.db"):
    con = sqlite3.connect(db)
    cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
    nc = "counter_name" if "counter_name" in cols else "name"
    for k, c, v in con.execute(f"select kernel_name, {nc}, sum(value) from counters_collection group by kernel_name, {nc}"):
        if k and "memetic_kernel" in k:
            tot[c] = tot.get(c, 0.0) + v
            key = (c, k.split("(")[0][-34:])
            tot[key] = tot.get(key, 0.0) + v
n = 163840.0
f, w = tot.get("FETCH_SIZE", 0.0) * 1024, tot.get("WRITE_SIZE", 0.0) * 1024
print(f"{name}: fetch x2 {2 * f / n / 1e3:.1f} KB + write {w / n / 1e3:.1f} KB = {(2 * f + w) / n / 1e3:.1f} KB per problem")
for key, v in sorted((k, v) for k, v in tot.items() if isinstance(k, tuple)):
    print(f"    {key[0]:11s} {key[1]:36s} {v * 1024 / n / 1e3:9.1f} KB per problem")
PY
done

export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/pmc_ln; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python bench.py --no_cpu_baseline --no_roofline --steps 2 --warmup 1 > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_ln"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    name = {}
    for row in csv.DictReader(open(f)):
        k = (row["Dispatch_Id"], row["Counter_Name"]); per[k] += float(row["Counter_Value"]); name[row["Dispatch_Id"]] = row["Kernel_Name"][:70]
    for (d, c), v in per.items():
        acc[name[d]][c].append(v)
rows = []
for k, cs in acc.items():
    f = cs.get("FETCH_SIZE", [0]); w = cs.get("WRITE_SIZE", [0])
    tot = (2 * sum(f) + sum(w)) * 1024 / 1e6 / 3   # MB per step (3 steps incl. warmup)
    rows.append((tot, k, len(f), 2 * max(f) * 1024 / 1e6, max(w) * 1024 / 1e6))
rows.sort(reverse=True)
with open(root + "/../ln_pmc_summary.txt", "w") as o:
    for tot, k, n, fm, wm in rows[:40]:
        o.write("%-70s calls %4d  MB/step %9.1f  max fetch %8.1f MB  max write %8.1f MB\n" % (k, n, tot, fm, wm))
PY
rm -rf $OUT
head -32 $R/gpurun_out/ln_pmc_summary.txt

"""Whole-step table: every kernel of the ONE-STREAM training step with >= 1 % of the step's kernel time: calls per
step, us per step, WRITE_SIZE + 2 x FETCH_SIZE MB per step (gfx950 correction per MI355X_MICROARCH.md), GB/s.  Three rocprofv3 runs of the
same bench command (kernel trace; FETCH_SIZE; WRITE_SIZE -- counters + kernel trace only).  python tools/step_traffic.py > profiles/rNN_step_traffic.txt"""
import collections, csv, glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARM = 6, 2
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "0", "--steps", str(STEPS), "--warmup", str(WARM), "--no_children",
       "--no_roofline", "--no_cpu_baseline", "--no_pmc", "--no_device_time"] + sys.argv[1:]
env = dict(os.environ, TMPDIR="/tmp")


def run(extra, tag):
    d = tempfile.mkdtemp(prefix="steptraffic_" + tag, dir="/tmp")
    subprocess.run(["rocprofv3"] + extra + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag, "--"] + CMD, env=env, cwd="/tmp",
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=False)
    return d


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "")[:64]


d = run([], "kt")
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
shutil.rmtree(d, ignore_errors=True)
byt = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = run(["--pmc", c], c)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            byt[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024 * (2.0 if c == "FETCH_SIZE" else 1.0)
    shutil.rmtree(d, ignore_errors=True)
n = STEPS + WARM                                            # (set-up kernels of the first step are counted in: < 2 %)
tot_us = sum(v[1] for v in dur.values()) / n
tot_b = sum(byt.values()) / n
print("# one-stream training step (bench.py --streams 0), per step over %d steps: kernel-time sum %.1f us, counter bytes %.2f GB = %.2f TB/s over "
      "the kernel time = %.3f of 8 TB/s" % (n, tot_us, tot_b / 1e9, tot_b / tot_us / 1e6, tot_b / tot_us / 1e6 / 8.0))
print("%-64s %6s %9s %6s %10s %8s" % ("kernel (>= 1 % of the kernel time)", "calls", "us/step", "%", "MB/step", "GB/s"))
rest_us = rest_b = 0.0
for k, (cnt, us) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    if us / n >= 0.01 * tot_us:
        print("%-64s %6.1f %9.1f %6.2f %10.1f %8.0f" % (k, cnt / n, us / n, 100 * us / n / tot_us, byt[k] / n / 1e6, byt[k] / us / 1e3))
    else:
        rest_us += us / n; rest_b += byt[k] / n
print("%-64s %6s %9.1f %6.2f %10.1f %8.0f" % ("(all kernels below 1 %)", "", rest_us, 100 * rest_us / tot_us, rest_b / 1e6, rest_b / max(rest_us, 1e-9) / 1e3))

"""Instruction histogram of one kernel of a `hipcc -S` listing -- whole kernel, or the stretch between the s_barrier before the first and
the s_barrier after the last line matching --mark (default: ds_read_b64_tr_b16 = the dW phase of cw_bwd_kernel).  Counts what round 6's
findings 56 / 57 were found with: scratch stores / reloads, exec-mask branches, s_waitcnt, MFMAs.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S tvqaplus_amd/csrc/cat3_bwd_dw.hip -o /tmp/cw.s
  python tools/isa_hist.py /tmp/cw.s 'cw_bwd_kernelILb1ELi3E' [--mark ds_read_b64_tr_b16] [--top 30]"""
import argparse, collections, re

ap = argparse.ArgumentParser()
ap.add_argument("listing"); ap.add_argument("kernel", help="regex matched against the mangled name of the kernel's label")
ap.add_argument("--mark", default=None); ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
lines = open(a.listing).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and re.search(a.kernel, l)][0]
end = start
while "s_endpgm" not in lines[end]:
    end += 1
k = lines[start:end + 1]
lo, hi = 0, len(k)
if a.mark:
    idx = [i for i, l in enumerate(k) if a.mark in l]
    lo, hi = idx[0], idx[-1]
    while lo > 0 and "s_barrier" not in k[lo]:
        lo -= 1
    while hi < len(k) - 1 and "s_barrier" not in k[hi]:
        hi += 1
c = collections.Counter()
for l in k[lo:hi]:
    t = l.strip()
    if t and not t.startswith((";", ".")) and not t.endswith(":"):
        c[t.split()[0]] += 1
tot = sum(c.values())
print("%s lines %d..%d: %d instructions | scratch st %d ld %d | exec branches %d | s_waitcnt %d | mfma %d | s_barrier %d" % (
    lines[start][:60], lo, hi, tot, sum(v for n, v in c.items() if n.startswith("scratch_store")),
    sum(v for n, v in c.items() if n.startswith("scratch_load")), c["s_cbranch_execz"] + c["s_cbranch_execnz"], c["s_waitcnt"],
    sum(v for n, v in c.items() if n.startswith("v_mfma")), c["s_barrier"]))
for n, v in c.most_common(a.top):
    print("  %-28s %d" % (n, v))

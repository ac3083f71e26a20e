#!/usr/bin/env python
"""Per-kernel register / spill / scratch summary of a HIP object file (build/*.o).  Usage: kernel_regs.py build/x.o [substr]"""
import glob, os, re, subprocess, sys, tempfile, shutil
llvm = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = tempfile.mkdtemp()
try:
    tmp = os.path.join(d, os.path.basename(obj)); shutil.copy(obj, tmp)
    subprocess.check_call([llvm + "/llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    b = glob.glob(tmp + ".*amdgcn*")[0]
    notes = subprocess.check_output([llvm + "/llvm-readelf", "--notes", b]).decode()
finally:
    shutil.rmtree(d)
for blk in notes.split("- .agpr_count:")[1:]:
    f = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = f("name").group(1)
    if pat not in name: continue
    print("%-100s vgpr %3s agpr %3s spill %3s scratch %4s lds %6s" % (name[:100], f("vgpr_count").group(1), blk.split()[0],
          f("vgpr_spill_count").group(1), f("private_segment_fixed_size").group(1), f("group_segment_fixed_size").group(1)))

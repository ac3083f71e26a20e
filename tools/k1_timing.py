import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tim = torch.zeros(8, dtype=torch.int64, device="cuda")
os.environ["STAGE_K1_TIM"] = str(tim.data_ptr())
import torch.nn.functional as F
from tvqaplus_amd import _lib
from tvqaplus_amd.synth import make_batch
lib = _lib.load()
N, NA, Li, Lqa, Lr, D = 16, 5, 300, 40, int(os.environ.get("LR", "20")), 128
g = torch.Generator().manual_seed(2018)
b = make_batch(N=N, Li=Li, Lr=Lr, Lw=2, Lqa=Lqa, wd_size=4, vfeat_size=4, seed=2018)
Cn = F.normalize(torch.randn(N, NA, Lqa, D, generator=g), dim=-1).cuda(); Q = torch.randn(N, Li, Lr, D, generator=g).cuda()
cm, qm = b.qas_mask.cuda().contiguous(), b.vid_mask.cuda().contiguous()
A = torch.empty(N, NA, Li, Lqa, D, device="cuda"); S = torch.empty(N, NA, Li, Lqa, Lr, device="cuda"); Sn = torch.empty_like(S)
st = torch.cuda.current_stream().cuda_stream
def launch(): lib.stage_str_attn_fwd(Cn.data_ptr(), Q.data_ptr(), cm.data_ptr(), qm.data_ptr(), A.data_ptr(), S.data_ptr(), Sn.data_ptr(), N, NA, Li, Lqa, Lr, D, 10.0, 0.0, 0, st)
for _ in range(3): launch()
torch.cuda.synchronize(); tim.zero_(); torch.cuda.synchronize()
reps = 10
for _ in range(reps): launch()
torch.cuda.synchronize()
t = tim.cpu().tolist()
names = ["staging", "wait_cf", "stage1", "softmax+Sstores", "stage2+Astores", "ticket/loop"]
tot = sum(t[:6])
print("per-wave avg cycles (100MHz counter?) per launch; total per wave", tot / reps / 2048)
for n_, v in zip(names, t): print("%-18s %12.1f  %5.1f%%" % (n_, v / reps / 2048, 100.0 * v / tot))

"""Do the kernels of the video attention at the stress shapes (Lr = 20, D = 256, bf16 storage) write outside their outputs?  Every output
is a slice of one sentinel-filled buffer with 4 KB guard zones on both sides; the guards are checked after each call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tvqaplus_amd import _lib
lib = _lib.load()
dev = "cuda"
N, NA, Li, Lqa, Lr, D = int(os.environ.get("N", 8)), 5, int(os.environ.get("LI", 150)), 40, int(os.environ.get("LR", 20)), 256
G = 4096
st = torch.cuda.current_stream().cuda_stream
class Arena:
    def __init__(self, nbytes):
        self.buf = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=dev); self.off = 0; self.slices = []
    def take(self, shape, dtype):
        n = 1
        for s in shape: n *= s
        nb = n * torch.empty((), dtype=dtype).element_size()
        self.off += G
        self.off = (self.off + 255) // 256 * 256
        t = self.buf[self.off:self.off + nb].view(dtype).view(shape)
        self.slices.append((self.off, nb))
        self.off += nb                      # (the next take() adds the guard; NOT rounded up: the byte right behind the tensor is guard)
        return t
    def check(self, what):
        mask = torch.ones_like(self.buf, dtype=torch.bool)
        for o, nb in self.slices: mask[o:o + nb] = False
        bad = ((self.buf != 0xA5) & mask).nonzero().flatten()
        if bad.numel():
            first = int(bad[0]); owner = [(i, o, nb) for i, (o, nb) in enumerate(self.slices) if o + nb <= first]
            print("GUARD OVERWRITTEN after %s: %d bytes, first at %d = %d bytes behind slice #%d" % (what, bad.numel(), first, first - (owner[-1][1] + owner[-1][2]), owner[-1][0]))
        else:
            print("guards intact after", what)
torch.manual_seed(1)
bf = torch.bfloat16
U = N * NA * Li * Lqa
A_ = Arena(U * D * 2 * 3 + U * Lr * 4 * 2 + U * 3 * D * 2 + 64 * G + (1 << 22))
Cn = torch.randn(N, NA, Lqa, D, device=dev).to(bf); Q = torch.randn(N, Li, Lr, D, device=dev).to(bf); Qn = torch.nn.functional.normalize(Q.float(), dim=-1).to(bf)
cm = torch.ones(N, NA, Lqa, device=dev); qm = (torch.rand(N, Li, Lr, device=dev) < 0.8).float(); qm[:, :, 0] = 1
A = A_.take((N, NA, Li, Lqa, D), bf); S = A_.take((N, NA, Li, Lqa, Lr), torch.float32); Sn = A_.take((N, NA, Li, Lqa, Lr), torch.float32)
_lib.check(lib.stage_str_attn_long_fwd(Cn.data_ptr(), Q.data_ptr(), Qn.data_ptr(), cm.data_ptr(), qm.data_ptr(), A.data_ptr(), S.data_ptr(), Sn.data_ptr(),
                                       N, NA, Li, Lqa, Lr, D, 10.0, 1, st), "long fwd")
torch.cuda.synchronize(); A_.check("stage_str_attn_long_fwd (slices: 0 A, 1 S, 2 Sn)")
# c2q: LayerNorm([a, b, a*b]) in bf16 + Linear 3D -> D
a = torch.randn(N * NA * Lqa, D, device=dev).to(bf)
gamma = torch.ones(3 * D, device=dev); beta = torch.zeros(3 * D, device=dev)
z = A_.take((U, 3 * D), bf); mean = A_.take((U,), torch.float32); rstd = A_.take((U,), torch.float32)
_lib.check(lib.stage_cat3_layernorm_fwd_bf16(a.data_ptr(), A.data_ptr(), gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             U, D, Li, Lqa, 1e-5, 0.1, 99, st), "cat3 ln bf16")
torch.cuda.synchronize(); A_.check("stage_cat3_layernorm_fwd_bf16 (3 z, 4 mean, 5 rstd)")
W = torch.randn(D, 3 * D, device=dev) * 0.05; bias = torch.zeros(D, device=dev)
y = A_.take((U, D), bf)
_lib.check(lib.stage_gemm_nt_bf16(z.data_ptr(), None, W.data_ptr(), bias.data_ptr(), None, y.data_ptr(), U, D, 3 * D, 1, st), "gemm nt bf16")
torch.cuda.synchronize(); A_.check("stage_gemm_nt_bf16 768 -> 256 (6 y)")

"""Host-side cost per custom-op call (tiny tensors, GPU idle): microseconds of CPU per forward / forward+backward call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvqaplus_amd import ops
dev = "cuda"
x = torch.randn(8, 128, device=dev, requires_grad=True); g = torch.ones(128, device=dev, requires_grad=True); b = torch.zeros(128, device=dev, requires_grad=True)
w = torch.randn(128, 128, device=dev, requires_grad=True)
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print("layernorm fwd        %.1f us" % t(lambda: ops.layernorm(x.detach(), g.detach(), b.detach())))
print("linear fwd           %.1f us" % t(lambda: ops.linear(x.detach(), w.detach(), b.detach(), relu=True)))
def fb_ln():
    y, _ = ops.layernorm(x, g, b); torch.autograd.grad(y.sum(), (x, g, b))
def fb_lin():
    y = ops.linear(x, w, b, relu=True); torch.autograd.grad(y.sum(), (x, w, b))
print("layernorm fwd+bwd    %.1f us" % t(fb_ln, 500))
print("linear fwd+bwd       %.1f us" % t(fb_lin, 500))
print("torch layer_norm fwd %.1f us" % t(lambda: torch.nn.functional.layer_norm(x.detach(), (128,), g.detach(), b.detach())))
print("torch linear fwd     %.1f us" % t(lambda: torch.relu(torch.nn.functional.linear(x.detach(), w.detach(), b.detach()))))
print("current_stream().cuda_stream %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream, 20000))
print("data_ptr()           %.2f us" % t(lambda: x.data_ptr(), 20000))

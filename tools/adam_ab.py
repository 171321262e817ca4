import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clm_gs_amd.clm_kernels import adam_rows
N = 28_000_000
dev = "cuda"
def run(cols_mode):
    ts = []
    for d in (3, 1, 3, 4):
        p = torch.randn(N, d, device=dev); g = torch.randn(N, d, device=dev)
        m = torch.zeros_like(p); v = torch.zeros_like(p)
        cols = d if cols_mode == "natural" else (4 if p.numel() % 4 == 0 else 1)
        lr = torch.full((cols,), 1e-3, device=dev)
        args = (p.view(-1, cols), g.view(-1, cols), m.view(-1, cols), v.view(-1, cols), None, lr, 0.9, 0.999, 1e-15, 1, True, 0.25, True)
        adam_rows(*args); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            adam_rows(*args)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
    return ts
for mode in ("natural", "vec4", "natural", "vec4"):
    t = run(mode)
    print(mode, [round(x, 3) for x in t], "sum", round(sum(t), 3), "ms;", round(N * 11 * 4 * 8 / sum(t) / 1e6, 0), "GB/s")
p = [torch.nn.Parameter(torch.randn(N, d, device=dev)) for d in (3, 1, 3, 4)]
opt = torch.optim.Adam(p, lr=1e-3, eps=1e-15, fused=True)
for q in p: q.grad = torch.randn_like(q)
opt.step(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): opt.step()
e.record(); torch.cuda.synchronize()
print("torch fused adam", round(s.elapsed_time(e) / 5, 3), "ms (+ /bsz and zeros passes separately)")

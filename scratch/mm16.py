"""Scratch: is an f16-in / f32-out library GEMM available through torch on this build, and how fast?"""
import time, torch
dev = torch.device('cuda:0')
P, K, N = 786432, 256, 256
def bench(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
a = torch.randn(P, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; g = torch.randn(P, N, device=dev)
print('f32 dX  (P,N)@(N,K): %.2f ms' % bench(lambda: g @ w))
print('f32 dW  (N,P)@(P,K): %.2f ms' % bench(lambda: g.t() @ a))
a16, w16, g16 = a.half(), w.half(), g.half()
print('f16->f16 dX: %.2f ms' % bench(lambda: g16 @ w16))
print('f16->f16 dW: %.2f ms' % bench(lambda: g16.t() @ a16))
try:
    r = torch.mm(g16, w16, out_dtype=torch.float32)
    print('out_dtype ok', r.dtype, float((r - g16.float() @ w16.float()).abs().max()))
    print('f16->f32 dX: %.2f ms' % bench(lambda: torch.mm(g16, w16, out_dtype=torch.float32)))
    print('f16->f32 dW: %.2f ms' % bench(lambda: torch.mm(g16.t(), a16, out_dtype=torch.float32)))
    g3 = torch.cat([g16, g16, g16], 1); w3 = torch.cat([w16, w16, w16], 0)
    print('f16->f32 dX K*3: %.2f ms' % bench(lambda: torch.mm(g3, w3, out_dtype=torch.float32)))
    gt3 = torch.cat([g16, g16, g16], 0); a3 = torch.cat([a16, a16, a16], 0)
    print('f16->f32 dW K*3: %.2f ms' % bench(lambda: torch.mm(gt3.t(), a3, out_dtype=torch.float32)))
except Exception as e:
    print('out_dtype failed:', repr(e)[:300])

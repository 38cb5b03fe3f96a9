"""Scratch: weight-volume decoder forward/backward, MIOpen vs batched-GEMM form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from humannerf_amd.network import MotionWeightVolumeDecoder, conv_transpose3d_k4s2p1
dev = torch.device('cuda:0')
dec = MotionWeightVolumeDecoder().to(dev)
pri = torch.rand(1, 25, 32, 32, 32, device=dev) + 0.01
def run_gemm():
    out = dec(motion_weights_priors=pri); out.sum().backward()
def run_miopen():
    h = dec.decoder.block_mlp(dec.const_embedding[None]).view(-1, 1024, 1, 1, 1)
    out = F.softmax(dec.decoder.block_conv(h) + torch.log(pri), dim=1); out.sum().backward()
def t(fn, n=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('GEMM form  fwd+bwd %.2f ms' % t(run_gemm))
print('MIOpen     fwd+bwd %.2f ms' % t(run_miopen))
with torch.no_grad():
    print('GEMM form  fwd %.2f ms' % t(lambda: dec(motion_weights_priors=pri)))
    print('MIOpen     fwd %.2f ms' % t(lambda: dec.decoder.block_conv(dec.decoder.block_mlp(dec.const_embedding[None]).view(-1, 1024, 1, 1, 1))))

"""Scratch: time of the head weight-gradient kernels (dY fp32 [P, 4|3] x blocked f16 activations [P, 256|128]) at the
training step's size, and the bandwidth that is.
    python profiles/tools/time_dw_head.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from humannerf_amd import ops
dev = torch.device('cuda:0')
P = 786432
for n_out, n_in in ((4, 256), (3, 128)):
    dY = torch.randn(P, n_out, device=dev)
    X = torch.randn(P, n_in, device=dev).clamp_(min=0).half()          # (any bytes do: the layout only matters to the result)
    for _ in range(3):
        ops.mlp_dw_h(dY, X, P=P, x_blocked=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        ops.mlp_dw_h(dY, X, P=P, x_blocked=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('head dW %d x %d over %d samples: %.4f ms per call (kernel + reduce), %.2f TB/s of operand reads' % (
        n_out, n_in, P, ms, P * (n_in * 2 + n_out * 4) / ms / 1e9))

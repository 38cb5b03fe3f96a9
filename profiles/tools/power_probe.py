"""Scratch: sample rocm-smi power / clocks while the f16x3 canonical kernel runs back to back."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import ops
from humannerf_amd.seeded import default_shapes, seeded_state
dev = torch.device('cuda:0')
st = seeded_state({k: v for k, v in default_shapes().items() if 'mlp' in k and 'decoder' not in k}, 0)
T = lambda a: torch.from_numpy(a).to(dev)
idx = [0, 2, 4, 6, 8, 10, 12, 14]
cw = [T(st[f'cnl_mlp.module.pts_linears.{i}.weight']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.weight'])]
cb = [T(st[f'cnl_mlp.module.pts_linears.{i}.bias']) for i in idx] + [T(st['cnl_mlp.module.output_linear.0.bias'])]
P = 32768 * 128
x = (torch.rand(P, 3, device=dev) * 2 - 1)
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--json'], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), 'ERR ' + repr(e)))
        time.sleep(0.3)
for mode in ('f16x3', 'f32'):
    cp = ops.canonical_pack(cw, cb, mode)
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    time.sleep(1.0)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6.0:
        for _ in range(20): ops.canonical(x, cp, mode)
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    time.sleep(0.5)
    stop = True; th.join()
    print(mode, 'kernel avg %.3f ms over %d launches' % (dt / n * 1e3, n))
    for t, s in samples[::3][:8]:
        print('  t=%.1f %s' % (t - t0, s[:600].replace('\n', ' ')))

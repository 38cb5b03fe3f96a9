"""The C ABI without Python in the process ("C-ABI layer beneath it"): tests/cabi/cabi_host.cpp
is a plain C++ program (hipMalloc'd buffers, one hipStream_t, no torch) linked against libhnrf.so; it renders one
ray chunk through hnrf_nonrigid_pack / hnrf_canonical_pack / hnrf_render_rays_fwd and exercises the error channel.
Here the same inputs are regenerated (its 64-bit LCG, vectorised) and pushed through the ctypes binding + the
step-by-step kernels: the two callers must agree bit for bit."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cabi', 'cabi_host')


class LCG:
    A, C = np.uint64(6364136223846793005), np.uint64(1442695040888963407)

    def __init__(self):
        self.state = np.uint64(0x9E3779B97F4A7C15)

    def uniform(self, n, lo, hi):
        with np.errstate(over='ignore'):
            a = np.cumprod(np.full(n, self.A, dtype=np.uint64))                    # a^1 .. a^n  (mod 2^64)
            s = np.cumsum(np.concatenate([[np.uint64(1)], a[:-1]]).astype(np.uint64))   # 1 + a + .. + a^(k-1)
            x = a * self.state + self.C * s
        self.state = x[-1]
        u = ((x >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.float32) / np.float32(16777216.0)
        return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
def test_cpp_host_through_the_c_abi(mode, tmp_path):
    from humannerf_amd import ops
    assert os.path.exists(HOST), 'tests/cabi/cabi_host missing: run __graft_entry__.build()'
    R, S, B, G = 333, 128, 24, 32
    out_file = str(tmp_path / 'out.bin')
    env = dict(os.environ)
    r = subprocess.run([HOST, str(R), str(S), str(ops.MLP_MODES[mode]), out_file], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'workspace' in r.stdout                       # message of the refused call came through hnrf_last_error
    got = np.fromfile(out_file, dtype=np.float32)
    assert got.size == R * 5

    g = LCG()
    dev = torch.device('cuda:0')
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rays_o = g.uniform(R * 3, -0.2, 0.2).reshape(R, 3); rays_o[:, 2] -= np.float32(3.0)
    rays_d = g.uniform(R * 3, -0.25, 0.25).reshape(R, 3); rays_d[:, 2] = 1.0
    near, far = g.uniform(R, 2.0, 2.3), g.uniform(R, 3.6, 4.0)
    Ts = g.uniform(B * 3, -0.1, 0.1).reshape(B, 3)
    Rs = np.zeros((B, 3, 3), np.float32)
    for b in range(B):
        a = np.float32(0.3) * (g.uniform(1, 0.0, 1.0)[0] - np.float32(0.5))
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))          # std::cos(float) / std::sin(float)
        Rs[b] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    vol = g.uniform((B + 1) * G ** 3, 0.0, 0.08).reshape(B + 1, G, G, G)
    cond = g.uniform(69, -0.3, 0.3)
    nr_shapes = [(128, 105), (128, 128), (128, 128), (128, 128), (128, 164), (128, 128), (3, 128)]
    cn_shapes = [(256, 63)] + [(256, 256)] * 4 + [(256, 319)] + [(256, 256)] * 2 + [(4, 256)]
    def layers(shapes):
        ws, bs = [], []
        for o, i in shapes:
            a = np.float32(np.sqrt(np.float32(6.0) / np.float32(i + o))) * np.float32(1.4)
            ws.append(T(g.uniform(o * i, -a, a).reshape(o, i)))
            bs.append(T(g.uniform(o, -0.05, 0.05)))
        return ws, bs
    nw, nb = layers(nr_shapes)
    cw, cb = layers(cn_shapes)
    bmin, bscale = T(np.full(3, -1.2, np.float32)), T(np.full(3, 2.0 / 2.4, np.float32))
    hann, bg = torch.ones(6, device=dev), T(np.array([255., 128., 0.], np.float32))
    nr_packed = ops.nonrigid_pack(nw, nb, T(cond), mode)
    cn_packed = ops.canonical_pack(cw, cb, mode)
    ref = ops.render_rays(T(rays_o), T(rays_d), T(near), T(far), None, T(Rs), T(Ts), T(vol), bmin, bscale, hann,
                          nr_packed, cn_packed, bg, S, mode)
    want = np.concatenate([ref['rgb'].cpu().numpy().ravel(), ref['alpha'].cpu().numpy(), ref['depth'].cpu().numpy()])
    # cos/sin of the 24 rotation angles may differ in the last bit between libm and numpy: allow 1e-6, expect equality
    assert np.isfinite(got).all() and got[:R * 3].std() > 1e-3
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)

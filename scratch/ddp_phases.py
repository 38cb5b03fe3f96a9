"""Scratch: phase timing of Trainer.backward_step with 2 gloo ranks sharing cuda:0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=2)
    from humannerf_amd import scene
    from humannerf_amd.config import cfg
    from humannerf_amd.network import Network
    from humannerf_amd.train import Trainer, image_loss
    from humannerf_amd.seeded import default_shapes, seeded_state
    dev = torch.device('cuda:0')
    net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()}); net = net.to(dev)
    fr = scene.synthetic_frame(H=512, W=512, focal_at_512=1700.0, pose_seed=rank)
    keys = ['rays', 'near', 'far', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec', 'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor']
    data = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in keys}
    idx = torch.arange(0, 6144, device=dev) * 37 % (512 * 512)
    tb = dict(data); tb['rays'] = data['rays'][:, idx].contiguous(); tb['near'] = data['near'][idx].contiguous(); tb['far'] = data['far'][idx].contiguous()
    tb['target_rgbs'] = torch.rand(6144, 3, device=dev)
    cfg.perturb, cfg.N_samples, cfg.train.lossweights.lpips = 1.0, 128, 0.0
    tr = Trainer(net, world_size=2)
    def sync(): torch.cuda.synchronize(); return time.perf_counter()
    for it in range(5):
        t0 = sync(); tr.network.train(); tr.optimizer.zero_grad(set_to_none=True)
        out = tr.network(**tb, iter_val=float(tr.iter)); t1 = sync()
        loss, _ = image_loss(out['rgb'][None, None], tb['target_rgbs'][None, None], None)
        loss.backward(); t2 = sync()
        tr.grad_sync.reduce(); t3 = sync()
        tr.optimizer_step(); t4 = sync()
        if rank == 0: print('it %d fwd %.1f bwd %.1f reduce %.1f opt %.1f ms' % (it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
    tb2 = dict(tb)
    ts = []
    torch.cuda.synchronize(); dist.barrier()
    for it in range(30):
        t0 = time.perf_counter(); tr.train_step(tb2); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if rank == 0: print('train_step walls (ms):', ' '.join('%.0f' % t for t in ts), flush=True)
    print('rank', rank, 'reserved GB', torch.cuda.memory_reserved() / 2**30, flush=True)
    dist.destroy_process_group()
if __name__ == '__main__':
    mp.spawn(w, args=(29544,), nprocs=2)

"""Scratch: what the gloo rehearsal of the 2-rank path pays per collective (both ranks on cuda:0)."""
import os, sys, time
import torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=2)
    dev = torch.device('cuda:0')
    x = torch.randn(840000, device=dev); v = torch.randn(25 * 32 ** 3, device=dev); c = torch.zeros(1, dtype=torch.float64, device=dev)
    for name, fn in (('allreduce 3.3MB', lambda: dist.all_reduce(x)), ('allreduce vol 3.3MB', lambda: dist.all_reduce(v)),
                     ('allgather f64', lambda: dist.all_gather([torch.empty_like(c), torch.empty_like(c)], c))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        if rank == 0: print(name, '%.2f ms' % ((time.perf_counter() - t0) * 200))
    dist.destroy_process_group()
if __name__ == '__main__':
    mp.spawn(w, args=(29533,), nprocs=2)

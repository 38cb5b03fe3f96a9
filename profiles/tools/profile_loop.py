"""cProfile of the main thread of run.run_movement (which call of the frame loop blocks on the GPU?)."""
import cProfile, io, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from humannerf_amd import dataset, run, scene
from humannerf_amd.config import cfg
from humannerf_amd.network import Network
from humannerf_amd.seeded import default_shapes, seeded_state
n = 16
d = tempfile.mkdtemp()
scene.write_synthetic_subject(d, n_frames=n, size=512, binary_mask=True)
cfg.N_samples, cfg.perturb, cfg.amd.diagnostics = 128, 0., False
dev = torch.device('cuda:0')
subj = dataset.Subject(d)
net = Network(); net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state(default_shapes(), 0).items()})
net = net.to(dev).eval()
out = tempfile.mkdtemp()
run.run_movement(net, subj, logdir=out, device=dev, test_num=3)
pr = cProfile.Profile()
pr.enable()
run.run_movement(net, subj, render_folder_name='timed', logdir=out, device=dev)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print(s.getvalue())

"""Not a test: times the full G+D train step at the C3 config on one GPU (debug / profiling aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histoGAN import Trainer

B = int(os.environ.get('B', 32)); S = int(os.environ.get('S', 256)); CAP = int(os.environ.get('CAP', 16))
tr = Trainer('probe', '/tmp/hg_results', '/tmp/hg_models', S, CAP, batch_size=B, hist_insz=150,
             hist_resizing='interpolation')
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
tr.init_GAN()
print('params G', sum(p.numel() for p in tr.GAN.G.parameters()), 'D', sum(p.numel() for p in tr.GAN.D.parameters()))
for i in range(3):
    tr.train(); torch.cuda.synchronize()
tr.steps = 1   # avoid GP/PL at steps%4==0 for the plain-step timing
t0 = time.perf_counter()
n = 6
for i in range(n):
    tr.steps = 1 + 4 * i
    tr.train()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'plain step {dt*1e3:.1f} ms  {B/dt:.1f} img/s  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
tr.steps = 32
t0 = time.perf_counter(); tr.train(); torch.cuda.synchronize()
print(f'GP+PL step {(time.perf_counter()-t0)*1e3:.1f} ms'); tr.print_log()

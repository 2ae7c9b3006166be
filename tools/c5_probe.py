"""BASELINE configs[4] on one GPU: HistoGAN 1024^2, capacity 16, batch 8, h = 128, discriminator attention on.
    python tools/c5_probe.py [attn layers, default 3,4]
(attention after discriminator block 1 at 1024^2 means q/k/v of 512 channels on 512^2 maps: 8.6 GB each for the
16-image [fake; real] pass -- out of memory at 288 GB together with the gradient penalty's double backward.)"""
import sys, os, tempfile, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ATTN = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '3,4').split(',') if v]
from histoGAN import Trainer
tmp = tempfile.mkdtemp()
tr = Trainer('c5', tmp+'/r', tmp+'/m', 1024, 16, batch_size=8, hist_bin=128, hist_insz=150, attn_layers=ATTN)
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src(pool=2)
for _ in range(10): tr.train()      # incl. the HG_GRAPH=auto decision (and the captures, if it takes the graph)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n): tr.train()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(workload='HistoGAN 1024^2 cap16 B=8 h=128 attn_layers=%s train step' % ATTN, ms_per_step=round(dt*1e3, 1), images_per_s=round(8/dt, 2),
      mem_gb=round(torch.cuda.max_memory_allocated()/2**30, 1), d=tr.d_loss, g=tr.g_loss,
      graph_replay=bool(getattr(tr, '_graph_auto', False)), host_ratio=[round(v, 3) for v in getattr(tr, '_host_ratio', [])])))

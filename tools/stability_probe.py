"""Numerical stability run: 160 HistoGAN and 80 ReHistoGAN train steps at 64x64 / capacity 8 on synthetic data (losses every 20 steps).
    python tools/stability_probe.py"""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from histoGAN import Trainer
from ReHistoGAN import recoloringTrainer
tmp = tempfile.mkdtemp()
tr = Trainer('s', tmp+'/r', tmp+'/m', 64, 8, batch_size=8, hist_bin=32, hist_insz=150)
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src(pool=8)
for i in range(161):
    tr.train(alpha=2)
    if i % 20 == 0: print('histogan', i, round(tr.d_loss,3), round(tr.g_loss,3), round(tr.h_loss,3), round(tr.last_gp_loss,3), tr.pl_mean)
tr2 = recoloringTrainer('s2', tmp+'/r', tmp+'/m', 64, 8, batch_size=8, hist_bin=32, hist_insz=150, variance_loss=True, skip_conn_to_GAN=True)
tr2.run_evaluate = tr2.run_save = False
tr2.set_synthetic_data_src(pool=8)
for i in range(81):
    tr2.train()
    if i % 20 == 0: print('rehistogan', i, round(tr2.d_loss,3), round(tr2.g_loss,3), round(tr2.r_loss,3), round(tr2.h_loss,3), round(tr2.var_loss,4))
out = tr.evaluate(num=0)
print('evaluate', tuple(out.shape), float(out.min()), float(out.max()))

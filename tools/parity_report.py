"""Not a test: prints parity errors for every golden case (debug aid, run on the GPU box)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from conftest import golden_names, load_golden, relmax
from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
from histogan_amd.hist import hellinger_loss

dev = torch.device('cuda:0')
for name in golden_names():
    g = load_golden(name)
    kw = dict(g['kwargs'])
    try:
        blk = RGBuvHistBlock(device='cuda', **kw)
        x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
        out = blk(x)
        ef = relmax(out.detach().cpu().numpy(), g['hist'])
        try:
            out.backward(torch.from_numpy(g['grad_out']).to(dev))
            eb = relmax(x.grad.cpu().numpy(), g['grad_x'])
        except Exception as e:
            eb = f'ERR {e}'
        print(f'{name:42s} fwd {ef:.3e}  bwd {eb if isinstance(eb,str) else format(eb,".3e")}', flush=True)
    except Exception:
        print(name, 'EXC'); traceback.print_exc()

import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from gpu_util import DEV, make_model
from parseq_amd.train import loss_and_grads
from oracle.synth import CONFIGS, synth_images
from test_training import CHARSET_94
cfg = CONFIGS['parseq']
m = make_model('parseq', 'bf16'); m.train_precision = 'bf16'
gen = torch.Generator().manual_seed(15)
B = int(os.environ.get('DIAG_B', '40'))
images = synth_images(B, cfg, seed=23).to(DEV)
lengths = torch.randint(1, 26, (B,), generator=gen).tolist(); lengths[3] = 25
labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
def run():
    m.rng = np.random.default_rng(8); torch.manual_seed(9)
    r = loss_and_grads(m, images, labels); torch.cuda.synchronize()
    return float(r.loss), {k: v.clone() for k, v in r.grads.items()}
la, ga = run()
os.environ['PARSEQ_TRAIN_ENC_PER_OP'] = '1'
lb, gb = run()
print('loss', la, lb)
for k in gb:
    if not k.startswith('encoder'): continue
    a, b = ga[k].double().flatten(), gb[k].double().flatten()
    if float(b.norm()) < 1e-9: continue
    print(f'{k:45s} rel {float((a-b).norm()/b.norm()):.3e} cos {float(a@b/(a.norm()*b.norm())):.5f}')

"""Time parseq_train_encoder_forward alone (batch 384, bf16-operand mode)."""
import os, sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from gpu_util import DEV, make_model
from parseq_amd import _native
from parseq_amd.train import _set_train_precision
from oracle.synth import CONFIGS, synth_images
cfg = CONFIGS['parseq']
m = make_model('parseq', 'bf16'); m.train_precision = 'bf16'
B = int(os.environ.get('B', '384'))
images = synth_images(B, cfg, seed=23).to(DEV)
lib = _native.lib()
native = m.model._sync_native().model
_set_train_precision(m, native)
nbytes = lib.parseq_train_encoder_workspace_bytes(native, B)
ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=DEV)
mem = torch.empty(B, 128, 384, dtype=torch.float32, device=DEV)
def fwd():
    _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(mem), _native.ptr(ws), nbytes, _native.stream_ptr(images)))
for _ in range(3): fwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fwd()
b.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get('PARSEQ_HIP_LIB', 'product')), 'B', B, 'forward ms', round(a.elapsed_time(b) / 10, 3))

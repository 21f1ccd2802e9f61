"""parseq_amd — MI355X-native PARSeq inference path (hand-written HIP for gfx950) behind the reference's Python surface.

    import torch
    model = torch.hub.load('/path/to/this/repo', 'parseq', source='local').eval().to('cuda')
    logits = model(images)                      # [N, L, 95], same contract as baudm/parseq
    labels, probs = model.tokenizer.decode(logits.softmax(-1))
"""
from .utils import InvalidModelError, create_model, load_from_checkpoint, parse_model_args  # noqa: F401

__version__ = '0.1.0'

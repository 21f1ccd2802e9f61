"""Factory functions with the reference's names and behaviour (`strhub/models/utils.py:10-104`):
`InvalidModelError`, `create_model`, `load_from_checkpoint`, `parse_model_args`, `get_pretrained_weights`."""
from __future__ import annotations

import torch

from .configs import get_config
from .model import init_weights  # noqa: F401  (re-exported like the reference module does)


class InvalidModelError(RuntimeError):
    """Exception raised for any model-related error (creation, loading)"""


# strhub/models/utils.py:14-20 (PARSeq and ViTSTR entries; ABINet / TRBA / CRNN are out of scope)
_WEIGHTS_URL = {
    'parseq-tiny': 'https://github.com/baudm/parseq/releases/download/v1.0.0/parseq_tiny-e7a21b54.pt',
    'parseq-patch16-224': 'https://github.com/baudm/parseq/releases/download/v1.0.0/parseq_small_patch16_224-fcf06f5a.pt',
    'parseq': 'https://github.com/baudm/parseq/releases/download/v1.0.0/parseq-bb5792a6.pt',
    'vitstr': 'https://github.com/baudm/parseq/releases/download/v1.0.0/vitstr-26d0fcf4.pt',     # utils.py:20
}


def _get_model_class(key: str):
    if 'parseq' in key:
        from .system import PARSeq as ModelClass
        return ModelClass
    if 'vitstr' in key:                                   # utils.py:58-59
        from .vitstr import ViTSTR as ModelClass
        return ModelClass
    raise InvalidModelError(f"Unable to find model class for '{key}' (PARSeq and ViTSTR are implemented here)")


def get_pretrained_weights(experiment: str):
    try:
        url = _WEIGHTS_URL[experiment]
    except KeyError:
        raise InvalidModelError(f"No pretrained weights found for '{experiment}'") from None
    return torch.hub.load_state_dict_from_url(url=url, map_location='cpu', check_hash=True)


def create_model(experiment: str, pretrained: bool = False, **kwargs):
    try:
        config = get_config(experiment, **kwargs)
    except FileNotFoundError:
        raise InvalidModelError(f"No configuration found for '{experiment}'") from None
    ModelClass = _get_model_class(experiment)
    model = ModelClass(**config)
    if pretrained:
        model.model.load_state_dict(get_pretrained_weights(experiment))
    return model


def load_from_checkpoint(checkpoint_path: str, **kwargs):
    if checkpoint_path.startswith('pretrained='):
        model_id = checkpoint_path.split('=', maxsplit=1)[1]
        return create_model(model_id, True, **kwargs)
    ModelClass = _get_model_class(checkpoint_path)
    # A Lightning checkpoint: {'state_dict': {'model.<key>': tensor, ...}, 'hyper_parameters': {...}} (train.py:86-92)
    ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    if 'state_dict' not in ckpt:                       # a bare inner-model state_dict, as released on GitHub
        raise InvalidModelError(f"'{checkpoint_path}' is not a Lightning checkpoint; use create_model(...).model.load_state_dict")
    hparams = dict(ckpt.get('hyper_parameters', {}))
    hparams.update(kwargs)
    model = ModelClass(**hparams)
    model.load_state_dict(ckpt['state_dict'])
    return model


def parse_model_args(args):
    kwargs = {}
    arg_types = {t.__name__: t for t in [int, float, str]}
    arg_types['bool'] = lambda v: v.lower() == 'true'
    for arg in args:
        name, value = arg.split('=', maxsplit=1)
        name, arg_type = name.split(':', maxsplit=1)
        kwargs[name] = arg_types[arg_type](value)
    return kwargs

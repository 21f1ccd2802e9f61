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
    """State dict of the released checkpoint of `experiment` (downloaded through torch.hub's cache)."""
    url = _WEIGHTS_URL.get(experiment)
    if url is None:
        raise InvalidModelError(f"No pretrained weights found for '{experiment}'")
    return torch.hub.load_state_dict_from_url(url=url, map_location='cpu', check_hash=True)


def create_model(experiment: str, pretrained: bool = False, **kwargs):
    """Build the system object of an experiment (`strhub/models/utils.py:73-83`); keyword arguments override its configuration."""
    try:
        config = get_config(experiment, **kwargs)
    except FileNotFoundError:
        raise InvalidModelError(f"No configuration found for '{experiment}'") from None
    system = _get_model_class(experiment)(**config)
    if pretrained:
        # utils.py:80-82: the released PARSeq files hold the INNER model's keys, every other released file (ViTSTR) the system's
        # ('model.'-prefixed) keys
        target = system.model if 'parseq' in experiment else system
        target.load_state_dict(get_pretrained_weights(experiment))
    return system


def load_from_checkpoint(checkpoint_path: str, **kwargs):
    """`pretrained=<experiment>` or the path of a Lightning checkpoint (`strhub/models/utils.py:86-93`).

    A Lightning checkpoint is {'state_dict': {'model.<key>': tensor, ...}, 'hyper_parameters': {...}} (written by
    train.py:86-92); the class is chosen from the file name, the hyper-parameters (overridden by `kwargs`) rebuild the system
    and the state dict is loaded with the system-level 'model.' prefix."""
    prefix = 'pretrained='
    if checkpoint_path.startswith(prefix):
        return create_model(checkpoint_path[len(prefix):], True, **kwargs)
    system_class = _get_model_class(checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    if 'state_dict' not in ckpt:                       # a bare inner-model state_dict, as released on GitHub
        raise InvalidModelError(f"'{checkpoint_path}' is not a Lightning checkpoint; use create_model(...).model.load_state_dict")
    hparams = {**ckpt.get('hyper_parameters', {}), **kwargs}
    system = system_class(**hparams)
    system.load_state_dict(ckpt['state_dict'])
    return system


_ARG_TYPES = {'int': int, 'float': float, 'str': str, 'bool': lambda text: text.lower() == 'true'}


def parse_model_args(args):
    """['name:type=value', ...] -> {name: typed value} with type in int / float / str / bool (`strhub/models/utils.py:96-104`)."""
    parsed = {}
    for item in args:
        key, text = item.split('=', maxsplit=1)
        name, type_name = key.split(':', maxsplit=1)
        parsed[name] = _ARG_TYPES[type_name](text)
    return parsed

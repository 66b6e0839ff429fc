"""Checkpoint / wire-format helpers.

The drop-in modules keep the reference's module tree, so ``state_dict()`` / ``load_state_dict()`` read
and write the reference's ``.pth`` files unchanged (codes/models/base_model.py:50-63; keys of
Appendix B).  What the reference does around them lives here:
* ``save_network`` — tensors moved to the CPU before ``torch.save`` (base_model.py:55-58);
* ``interpolate`` — the network-interpolation script (codes/scripts/net_interp.py:16-18);
* ``save_training_state`` / ``resume_training`` — the ``.state`` files (base_model.py:66-85);
  ``optim.FusedAdam.state_dict()`` emits torch.optim.Adam's layout, so resume files interchange."""
from collections import OrderedDict

import torch


def save_network(network, path):
    if isinstance(network, torch.nn.DataParallel):
        network = network.module
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in network.state_dict().items()), path)


def load_network(path, network, strict=True):
    if isinstance(network, torch.nn.DataParallel):
        network = network.module
    network.load_state_dict(torch.load(path, map_location='cpu'), strict=strict)


def interpolate(sd_a, sd_b, alpha):
    """(1 - alpha) * A + alpha * B over every key of A (net_interp.py:16-18)."""
    out = OrderedDict()
    for k, va in sd_a.items():
        out[k] = (1 - alpha) * va + alpha * sd_b[k]
    return out


def save_training_state(path, epoch, iter_step, optimizers, schedulers):
    state = {'epoch': epoch, 'iter': iter_step,
             'schedulers': [s.state_dict() for s in schedulers],
             'optimizers': [o.state_dict() for o in optimizers]}
    torch.save(state, path)


def resume_training(resume_state, optimizers, schedulers):
    ro, rs = resume_state['optimizers'], resume_state['schedulers']
    assert len(ro) == len(optimizers), 'Wrong lengths of optimizers'
    assert len(rs) == len(schedulers), 'Wrong lengths of schedulers'
    for o, sd in zip(optimizers, ro):
        o.load_state_dict(sd)
    for s, sd in zip(schedulers, rs):
        s.load_state_dict(sd)

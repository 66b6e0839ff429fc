"""Checkpoint / wire-format helpers.

The drop-in modules keep the reference's module tree, so ``state_dict()`` / ``load_state_dict()`` read
and write the reference's ``.pth`` files unchanged (codes/models/base_model.py:50-63; keys of
Appendix B).  What the reference does around them lives here:
* ``save_network`` — tensors moved to the CPU before ``torch.save`` (base_model.py:55-58);
* ``interpolate`` — the network-interpolation script (codes/scripts/net_interp.py:16-18);
* ``save_training_state`` / ``resume_training`` — the ``.state`` files (base_model.py:66-85);
  ``optim.FusedAdam.state_dict()`` emits torch.optim.Adam's layout, so resume files interchange;
* ``save_step`` / ``resume_step`` — what ``model.save(step)`` + ``model.save_training_state(epoch, step)``
  (codes/train.py:162-165) and the resume branch (train.py:27-28, 86-91; options.py:106-120) do for a
  ``train.ESRGANPlusStep``: ``{step}_G.pth`` / ``{step}_D.pth`` / ``{step}.state`` in the reference's layouts, taken
  behind whatever a pipelined step left in flight, plus — new, the reference trains in fp32 — the dynamic loss
  scaler's state, so that a resumed fp16 run continues bit for bit (tests/test_gpu_train_step.py)."""
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


def save_step(step, directory, epoch, iter_step, schedulers=()):
    """models/{iter}_G.pth, models/{iter}_D.pth, training_state/{iter}.state of a ``train.ESRGANPlusStep`` (the file
    names and dict layouts of base_model.py:50-74; `directory` plays opt['path']['models'] / ['training_state']).
    The step's side-stream tail (D's Adam, weight packs) is ordered first; the networks' ``state_dict()`` join it too.
    Returns the three paths."""
    import os
    step.finish()
    os.makedirs(directory, exist_ok=True)
    pg = os.path.join(directory, '%s_G.pth' % iter_step)
    pd = os.path.join(directory, '%s_D.pth' % iter_step)
    ps = os.path.join(directory, '%s.state' % iter_step)
    save_network(step.netG, pg)
    save_network(step.netD, pd)
    state = {'epoch': epoch, 'iter': iter_step,
             'schedulers': [s.state_dict() for s in schedulers],
             'optimizers': [step.optimizer_G.state_dict(), step.optimizer_D.state_dict()]}
    if step.scaler is not None:
        # fp16 path only (no counterpart in the reference): {scale, -, good steps, -, found flags}
        state['loss_scaler'] = step.scaler.state.detach().cpu()
    torch.save(state, ps)
    return pg, pd, ps


def resume_step(step, directory, iter_step, schedulers=()):
    """The inverse of ``save_step``: weights into the step's networks (their packed copies are refreshed at the next
    forward: load_state_dict marks them), Adam moments and step counts into its optimizers, schedulers, the loss
    scaler.  Returns (epoch, iter) as train.py:86-91 reads them."""
    import os
    step.finish()
    load_network(os.path.join(directory, '%s_G.pth' % iter_step), step.netG)
    load_network(os.path.join(directory, '%s_D.pth' % iter_step), step.netD)
    state = torch.load(os.path.join(directory, '%s.state' % iter_step), map_location='cpu')
    resume_training(state, [step.optimizer_G, step.optimizer_D], list(schedulers))
    if step.scaler is not None:
        if 'loss_scaler' in state:
            step.scaler.state.copy_(state['loss_scaler'].to(step.scaler.state.device))
        else:
            import warnings
            warnings.warn('resume_step: %s.state holds no "loss_scaler" (written by the reference, or by a run with a '
                          'static loss scale): the dynamic loss scale restarts from its initial value, so the first '
                          'steps after the resume are not bit-identical to the uninterrupted run' % iter_step,
                          RuntimeWarning, stacklevel=2)
    return state['epoch'], state['iter']

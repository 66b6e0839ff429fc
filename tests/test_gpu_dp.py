"""Data-parallel backward on the GPU: the segmented RRDBNet backward that feeds dp.GradExchange slice by
slice (one GPU: against the one-piece backward, with a recording stand-in for the all-reduce), and the real
exchange over RCCL on two GPUs (skipped on boxes with fewer)."""
import os
import socket
import sys

import pytest
import torch

from esrganplus_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


class _Recorder:
    """Stands in for dp.GradExchange inside the backward node: notes the slices it is handed."""

    def __init__(self, bucket_elems):
        self.bucket_elems = bucket_elems
        self.spans = []

    def __call__(self, t):
        self.spans.append((t.storage_offset(), t.numel()))
        return None


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
@pytest.mark.parametrize('bucket_elems', [1, 1 << 30])
def test_segmented_backward_equals_one_piece(dev, precision, bucket_elems):
    from esrganplus_amd import architecture as arch
    nb = 3
    sd = synth.rrdbnet_state_dict(nb=nb, seed=5)
    x = synth.image_batch(5, 2, 3, 16, 24, name='seg.x').to(dev)
    gy = synth.normal_like(5, 'seg.gy', (2, 3, 64, 96)).to(dev)
    grads = {}
    for seg in (False, True):
        net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval().set_precision(precision)
        net.load_state_dict(sd)
        rec = _Recorder(bucket_elems)
        if seg:
            net.attach_grad_sync(rec)
        (net(x) * gy).sum().backward()
        grads[seg] = {k: p.grad.clone() for k, p in net.named_parameters()}
        if seg:
            total = sum(p.numel() for p in net.parameters())
            spans = sorted(rec.spans)
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total      # slices tile the flat buffer
            assert all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:]))
            # tail, RRDB 2, 1, 0, first conv — merged into one when the bucket is larger than the net
            assert len(spans) == (nb + 2 if bucket_elems == 1 else 1)
    for k in grads[False]:
        a, b = grads[False][k], grads[True][k]
        # fp32 atomics: run-to-run rounding only
        assert (a - b).norm().item() <= 1e-5 * a.norm().clamp_min(1e-6).item(), k


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from esrganplus_amd import architecture as arch, dp
    try:
        assert dp.init_from_env('nccl') == world
        dev = torch.device('cuda', rank)
        sd = synth.rrdbnet_state_dict(nb=2, seed=9)
        x = synth.image_batch(9 + rank, 2, 3, 16, 16, name='dp.x').to(dev)      # every rank its own minibatch shard
        gy = synth.normal_like(9 + rank, 'dp.gy', (2, 3, 64, 64)).to(dev)

        def grads_of(attach):
            net = arch.RRDBNet(3, 3, 64, 2).to(dev).eval().set_precision('fp32')
            net.load_state_dict(sd)
            ex = dp.GradExchange(net, bucket_bytes=1 << 20, overlap=attach)
            assert ex.inline == attach
            (net(x) * gy).sum().backward()
            if attach:
                ex.start(); ex.wait()                       # no-ops: the gradients arrive averaged
            return torch.cat([p.grad.reshape(-1) for p in net.parameters()]), ex
        local, ex0 = grads_of(False)
        want = local.clone()
        dist.all_reduce(want)
        want /= world
        got, _ = grads_of(True)                             # in-backward exchange over RCCL
        assert (got - want).norm().item() <= 1e-5 * want.norm().item(), 'in-backward exchange'
        ex0.start(); ex0.wait()                             # after-backward exchange of the same gradients
        q.put((rank, 'ok'))
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_in_backward_exchange_two_ranks_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (RCCL)')
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res

"""The driver's command line end to end: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with
the contract's keys (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload) plus `roofline` and `cpu_baseline`; the train modes print theirs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), cwd=ROOT, check=True,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900).stdout.decode()
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_forward_line_follows_the_contract():
    d = _run('--gpus', '1', '--steps', '3', '--warmup', '1')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'fwd_bwd'):
        assert k in d, k
    assert (d['n_gpus'], d['steps'], d['warmup'], d['unit'], d['dtype'], d['scaling']) == (1, 3, 1, 'HR-Mpix/s', 'f16', 'weak')
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 16 * 512 * 512 / 1e6 / (d['ms_per_step'] / 1e3)) <= 1e-2 * d['value']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-3 and 0.2 < r['frac'] < 0.7
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'HR-Mpix/s' and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert 'settle_steps' not in d
    # round 5: the sustained dense fp16 rate of this box (register-resident MFMA loop, random operands) as context
    pr = d['mfma_sustained_probe']
    assert 800 < pr['tflops'] < 2600 and 1.0 < pr['shader_clock_ghz'] < 2.6
    assert d['roofline']['frac'] < d['roofline']['frac_of_sustained_probe'] < 1
    # BASELINE configs[2] / configs[4] ride in the same line
    t = d['train_step']
    assert t['ms_per_step'] > 0 and 0 < t['frac_of_f16_mfma_peak'] < 1 and t['roofline']['kernel'].startswith('rdb_')
    assert abs(t['tflops'] - 3.03e3 / t['ms_per_step']) <= 0.02 * t['tflops']
    # round 5: the synchronous figure next to the pipelined one, and HBM-side traffic of the train shape (PMC rows)
    assert t['ms_per_step_sync_log'] >= 0.9 * t['ms_per_step'] and t['roofline']['traffic'] and t['roofline']['traffic'] > 1e8
    gt = d['gtrain']
    assert gt['ms_per_step'] > 0 and set(gt['buckets']) == {'16x128^2', '8x192^2', '4x256^2'}
    assert abs(sum(b['ms'] for b in gt['buckets'].values()) - gt['ms_per_step']) <= 0.15 * gt['ms_per_step']


@pytest.mark.parametrize('mode', ['train', 'gtrain'])
def test_train_modes_print_one_line(mode):
    d = _run('--mode', mode, '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    assert d['steps'] == 2 and d['n_gpus'] == 1 and d['unit'] == 'HR-Mpix/s' and d['value'] > 0
    assert 'workload' in d['config']


def test_gtrain_bucket_loop_without_autograd_equals_the_autograd_loop():
    """bench.py's configs[4] loop drives the generator's launch lists directly (rrdbnet_train_forward / l1_raw /
    rrdbnet_train_backward, as train.ESRGANPlusStep does); ESR_GTRAIN_MANUAL=0 is the same loop through autograd.  Same
    kernels, same order, deterministic reductions: after three steps over three small buckets the weights and Adam
    moments are bit-identical."""
    code = r'''
import os, sys, types, torch
sys.path.insert(0, %r)
import bench
bench.GTRAIN_BUCKETS = ((3, 32), (2, 48), (1, 64))
bench.NB = 2
args = types.SimpleNamespace(precision='fp16')
dev = torch.device('cuda:0')
import esrganplus_amd.architecture as arch
keep = {}
orig = arch.RRDBNet.__init__
def init(self, *a, **k):
    orig(self, *a, **k); keep['net'] = self
arch.RRDBNet.__init__ = init
torch.manual_seed(7)
bench.measure_gtrain(args, 1, 0, dev, None, steps=2, warmup=1)
torch.cuda.synchronize()
sd = {k: v.detach().cpu() for k, v in keep['net'].state_dict().items()}
torch.save(sd, sys.argv[1])
''' % ROOT
    import tempfile
    import torch
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for manual in ('1', '0'):
            path = os.path.join(td, 'sd%s.pt' % manual)
            r = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, ESR_GTRAIN_MANUAL=manual),
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            outs.append(torch.load(path))
    assert outs[0].keys() == outs[1].keys()
    bad = [k for k in outs[0] if not torch.equal(outs[0][k], outs[1][k])]
    assert not bad, bad[:6]

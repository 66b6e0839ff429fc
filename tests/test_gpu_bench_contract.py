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
    # BASELINE configs[2] / configs[4] ride in the same line
    t = d['train_step']
    assert t['ms_per_step'] > 0 and 0 < t['frac_of_f16_mfma_peak'] < 1 and t['roofline']['kernel'].startswith('rdb_')
    assert abs(t['tflops'] - 3.03e3 / t['ms_per_step']) <= 0.02 * t['tflops']
    gt = d['gtrain']
    assert gt['ms_per_step'] > 0 and set(gt['buckets']) == {'16x128^2', '8x192^2', '4x256^2'}
    assert abs(sum(b['ms'] for b in gt['buckets'].values()) - gt['ms_per_step']) <= 0.15 * gt['ms_per_step']


@pytest.mark.parametrize('mode', ['train', 'gtrain'])
def test_train_modes_print_one_line(mode):
    d = _run('--mode', mode, '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    assert d['steps'] == 2 and d['n_gpus'] == 1 and d['unit'] == 'HR-Mpix/s' and d['value'] > 0
    assert 'workload' in d['config']

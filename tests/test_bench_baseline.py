"""bench.py's cpu_baseline: kind = "reference" comes from the record tools/gpu_jobs/with_reference.sh left of the UNMODIFIED reference
timed on the GPU box's host (profiles/r06_reference_cpu_on_gpu_box.json) — and only on a box whose fingerprint (CPU model + logical
cores + torch version) is the record's; anywhere else the live port timing is the baseline and says why the record was refused."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _port():
    return dict(value=2.0e5, unit='env_steps/s', cores=16, kind='port', sample='x', cpu_model='whatever')


def test_record_is_the_reference_and_complete():
    doc = json.load(open(os.path.join(REPO, 'profiles', 'r06_reference_cpu_on_gpu_box.json')))
    assert 'UNMODIFIED reference' in doc['what'] and doc['box']['cores_logical'] >= doc['box']['cores_physical'] >= 1
    for cfg in ('c1', 'c2', 'c3', 'c4'):                # C1, C2, the C3-policy and the C4-policy (models.Convolutional)
        s = doc['summary'][cfg]
        assert s['serial']['backend'] == 'serial' and s['serial']['value'] > 0 and s['best']['value'] >= s['serial']['value']
        assert 'pufferlib.vector.Serial' in s['serial']['what'] and 'c_gae.pyx' in s['serial']['what']
    assert doc['summary']['c2']['serial']['envs'] == 4096 and doc['summary']['c2']['serial']['horizon'] == 128
    assert all(r['reference_dir'] == '_refstage' for r in doc['runs'])      # timed from the staged copy on the GPU box


def test_fingerprint_gates_the_record(monkeypatch):
    import bench
    doc = json.load(open(os.path.join(REPO, 'profiles', 'r06_reference_cpu_on_gpu_box.json')))
    monkeypatch.setattr(bench, '_cpu_model', lambda: doc['box']['cpu_model'])
    monkeypatch.setattr(bench.os, 'cpu_count', lambda: doc['box']['cores_logical'])
    import torch
    monkeypatch.setattr(torch, '__version__', doc['box']['torch'])
    got = bench.reference_cpu_baseline('c2', _port())
    assert got['kind'] == 'reference' and got['value'] == doc['summary']['c2']['serial']['value'] and got['cores'] == 16
    assert got['port_live']['value'] == 2.0e5 and got['best_vectoriser']['backend'] == 'multiprocessing'
    assert got['source'].startswith('profiles/r06_reference_cpu_on_gpu_box.json')
    monkeypatch.setattr(bench.os, 'cpu_count', lambda: doc['box']['cores_logical'] // 2)        # another box
    got = bench.reference_cpu_baseline('c2', _port())
    assert got['kind'] == 'port' and got['value'] == 2.0e5 and 'was taken on' in got['reference_record_refused']
    assert bench.reference_cpu_baseline('nope', _port())['kind'] == 'port'                       # no such configuration in the record
    monkeypatch.setattr(bench.os, 'cpu_count', lambda: doc['box']['cores_logical'])
    c4 = bench.reference_cpu_baseline('c4', _port())                                             # the conv policy's baseline is the reference's too (round 6)
    assert c4['kind'] == 'reference' and 'models.Convolutional' in c4['sample'] and 0 < c4['value'] < 1e5
    monkeypatch.setattr(torch, '__version__', '0.0.0')                                           # another torch build: the record does not speak for it
    assert bench.reference_cpu_baseline('c2', _port())['kind'] == 'port'


def test_staging_list_is_the_hot_path_only():
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import stage_reference
    if not os.path.exists('/root/reference/demo.py'):
        import pytest
        pytest.skip('needs the reference checkout (build container)')
    files = stage_reference.files()
    assert {'demo.py', 'config.yaml', 'clean_pufferl.py', 'c_gae.pyx', 'pufferlib/vector.py', 'pufferlib/emulation.py',
            'pufferlib/environments/ocean/ocean.py', 'pufferlib/frameworks/cleanrl.py'} <= set(files)
    assert not any(f.startswith(('tests/', 'examples/', 'pufferlib/environments/atari')) for f in files) and len(files) < 40
    assert not os.path.exists(os.path.join(REPO, '_refstage'))          # scratch: never left behind
    ign = open(os.path.join(REPO, '.gitignore')).read()
    assert '_refstage/' in ign

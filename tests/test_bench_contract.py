"""The one-line JSON contract of bench.py, checked on the lines committed under profiles/ (the runs behind DESIGN.md's
tables): every key the driver and the judge read is present and consistent with the others."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.isfile(path):
        pytest.skip('%s not committed' % name)
    return json.load(open(path))


@pytest.mark.parametrize('name', ['r02_bench_c2.json', 'r02_bench_c4.json', 'r02_bench_c5.json', 'r03_bench_default.json'])
def test_bench_line_has_the_contract_keys(name):
    d = _load(name)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['unit'] == 'atom-selections/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    # value = selections of one step / time of one step
    assert abs(d['value'] - d['config']['selections_per_step'] * d['n_gpus'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-9 and 0.0 < r['frac'] <= 1.0
    c = d['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0


def test_config2_line_reports_what_section_8d_defines():
    d = _load('r02_bench_c2.json')
    assert d['dtype'] == 'f32' and d['config']['signals_per_gpu'] == 1024 and d['config']['T'] == 65536
    assert d['config']['K'] == 256 and d['config']['W'] == 64 and d['config']['L0'] == 256
    assert d['steps'] * d['ms_per_step'] >= 4000.0                          # >= 4 s of GPU time in the timed region
    assert all(d['config']['output_check'].values())
    assert 0 < d['value_incl_transfers'] <= d['value']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 157.3
    ks = {k['kernel'].split()[0]: k for k in r['all_kernels']}
    assert set(ks) == {'corr_init', 'iterate'}
    for k in ks.values():                                                    # achieved = algorithmic flop / measured duration
        assert abs(k['tflops'] - k['algorithmic_tflop'] / (k['ms'] * 1e-3)) <= 1e-6 * k['tflops']
    assert 0 < r['hbm_fraction'] < 0.1 and r['table_scan_equiv']['bytes_per_selection'] == 65536 * 256 * 4
    assert r['traffic'] is None or r['traffic'] > 0


def test_default_line_carries_the_hierarchical_configurations():
    """The default invocation (what the driver records) appends a bounded `secondary` object: BASELINE configs[3] with 17 and
    with the literal 16 level-1 taps, configs[4] at its per-GPU share -- per-level rooflines, the rate with the residual
    samples fetched, the reconstruction check."""
    d = _load('r03_bench_default.json')
    sec = d['secondary']
    assert set(sec) == {'config4_17taps', 'config4_16taps', 'config5', 'config4_17taps_locomp'}
    for name, v in sec.items():
        assert 'error' not in v and 'skipped' not in v, (name, v)
        # (three timed steps each: the two rates are separate measurements, within run-to-run noise of each other where the fetch is small)
        assert v['unit'] == 'atom-selections/s' and v['value'] > 0 and 0 < v['value_incl_residual_transfer'] <= 1.1 * v['value']
        assert v['signals_per_gpu'] == (128 if name == 'config5' else 1024)
        assert v['method'] == ('locomp' if name.endswith('locomp') else 'cmp')
        assert ('locomp' in v['levels'][0]['variant']) == (v['method'] == 'locomp')
        assert v['output_check']['device_energies_match_fetched_residuals'] and 'FAILED' not in v['output_check']
        assert v['levels'][0]['bound'] == 'mfma' and 0 < v['levels'][0]['loop_frac'] <= 1
        for l in v['levels'][1:]:
            assert l['bound'] == 'latency' and 0 < l['latency_model']['frac'] <= 1 and 0 < l['hbm']['frac'] < 1
    assert sec['config4_17taps']['output_check']['reconstructs'] is True and sec['config5']['output_check']['reconstructs'] is True
    # the reference script's own method (learn_mlcsc_dataset.py:108 takes the encoder's default, LoCOMP) within 3 x of the greedy method
    assert sec['config4_17taps_locomp']['output_check']['reconstructs'] is True
    assert sec['config4_17taps_locomp']['ms_per_step'] <= 3.0 * sec['config4_17taps']['ms_per_step']
    assert sec['config4_16taps']['output_check']['reconstructs'] is None          # (the reference itself reaches 2.9 dB on that hierarchy)


def test_no_tools_script_shadows_a_root_module():
    """tools/ holds stand-alone scripts; some of them put tools/ on sys.path.  A script there named like a root module (bench_hsc.py
    once was) would be imported in its place -- bench.py's secondary section lost `bench_hsc.compact` that way."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    top = {f for f in os.listdir(root) if f.endswith('.py')}
    tools = {f for f in os.listdir(os.path.join(root, 'tools')) if f.endswith('.py')}
    assert not (top & tools), top & tools
    import bench_hsc
    assert callable(bench_hsc.compact) and callable(bench_hsc.run)
    import tools_csrc_digest
    assert len(tools_csrc_digest.csrc_digest()) == 64

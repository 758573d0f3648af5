"""The ONE JSON line bench.py prints must fit the driver's reader (round 5's 20 kB line was cut at 8 000 characters and never
parsed): bench.compact_line on a full result of the real shape - a committed one - stays under bench.LINE_LIMIT, carries the
contract's keys, `roofline` and `cpu_baseline` with their fields, and refuses to emit a line that would not fit."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned():
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r5_bench_line_em_200k_100k_64.json')))
    full['parity_timed'] = {'argmax_identical': True, 'assignments_differing': 0, 'max_abs_posterior_diff': 3.1e-7, 'within_1e-5': True,
                            'barcodes': 200000, 'em_iterations': 45, 'against': 'exact_mode region, same start'}
    full['clocks_warm'] = dict(full['after_idle'])
    full['cold_start_5it'] = {'ms_per_step': 2.2, 'calls_ms': [11.0, 11.1, 11.2]}
    full['default_call_5it_ms_per_iteration'] = 2.36
    return full


def test_the_line_fits_and_parses():
    text = bench.compact_line(canned(), 'gpurun_out/bench_details_em_200k_100k_64_n1.json')
    assert len(text) < bench.LINE_LIMIT <= 4000, len(text)
    assert '\n' not in text
    line = json.loads(text)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert key in line, key
    assert line['vs_baseline'] is None and line['higher_is_better'] is True
    assert line['config']['workload'] == 'em_200k_100k_64' and 'model' not in line['config']
    for key in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'estep_ms'):
        assert key in line['roofline'], key
    assert abs(line['roofline']['frac'] - line['roofline']['achieved'] / line['roofline']['peak']) < 1e-4
    for key in ('value', 'unit', 'cores', 'host_cores', 'kind', 'sample'):
        assert key in line['cpu_baseline'], key
    for key in ('exact_mode_ms_per_step', 'after_idle_ms_per_step', 'default_call_5it_ms_per_iteration', 'parity_timed', 'details'):
        assert key in line, key
    assert line['value'] == pytest.approx(canned()['value'], rel=1e-5)


def test_a_line_that_would_not_fit_is_refused():
    full = canned()
    full['config']['estep_mode'] = 'x' * 5000
    with pytest.raises(AssertionError, match='bench line unusable'):
        bench.compact_line(full)


def test_multi_rank_line_has_the_contract_keys_without_a_cpu_baseline():
    full = canned()
    full['n_gpus'], full['cpu_baseline'] = 8, None
    full['weak'] = {'value': 1.2e9, 'ms_per_step': 1.3, 'barcodes_total': 1600000, 'exchange_ms_per_step': 0.2, 'note': 'n' * 300}
    line = json.loads(bench.compact_line(full))
    assert line['cpu_baseline'] is None and line['n_gpus'] == 8 and line['weak']['value'] == 1.2e9

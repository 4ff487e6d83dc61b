"""bench.py's roofline block is derived from committed evidence: the PMC summary of the
same command (profiles/pmc_*.json) and the in-kernel clock (profiles/clock.json).  These
CPU tests pin the arithmetic of that derivation and the presence / sanity of the files."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_figure_of_the_survey(bench):
    # S*U*W*2^d*T + S*(2T + 4 nu); config 4 = 2.199e12 + 3.36e8 bytes per sweep
    S, U, W = 256 ** 3, 64, 32
    b = bench.algorithmic_bytes(S, float(S) * U * W, 3, 8, 1)
    assert b == S * U * W * 8 * 8 + S * 20
    assert abs(b - 2.1994e12) / 2.1994e12 < 1e-3
    # fp32 config 5: 32 B per lattice cell
    assert bench.algorithmic_bytes(512 ** 3, float(512 ** 3) * U * W, 3, 4, 1) == 512 ** 3 * U * W * 32 + 512 ** 3 * 12


def test_peaks_are_the_spec_numbers(bench):
    # 78.6 TFLOP/s fp64 vector = 1024 SIMD x 16 lanes x 2 flop x 2.4 GHz  <=>  one wave64
    # instruction per 4 clocks per SIMD
    assert bench.FP64_ISSUE_PEAK == 256 * 4 * 2.4e9 / 4 == 6.144e11
    assert abs(bench.FP64_ISSUE_PEAK * 64 * 2 / 1e12 - 78.6) < 0.1
    assert bench.FP32_ISSUE_PEAK == 2 * bench.FP64_ISSUE_PEAK and bench.HBM_PEAK_GBS == 8000.0


def test_committed_pmc_summary_of_the_headline_workload(bench):
    pmc, path = bench.load_pmc('synth256_f64_column')
    assert pmc is not None and path == os.path.join('profiles', 'pmc_synth256_f64_column.json')
    c = pmc['counters_mean_per_dispatch']
    n = c['SQ_INSTS_VALU_ADD_F64'] + c['SQ_INSTS_VALU_MUL_F64'] + c['SQ_INSTS_VALU_FMA_F64']
    assert pmc['valu_wave_instr'] == n
    analytic = 256 ** 3 * 64 * 32 * 6 / 64.0                 # six operations per lattice cell
    assert analytic <= n <= 1.25 * analytic                  # the count cannot be below the floor
    assert pmc['hbm_bytes'] == (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
    assert 0.47e9 <= pmc['hbm_bytes'] <= 10e9                # at least the compulsory traffic
    # a frac computed from the committed duration stays below 1 against the spec peak
    frac = n / (pmc['avg_kernel_ms_trace_pass'] * 1e-3) / bench.FP64_ISSUE_PEAK
    assert 0.3 < frac < 1.0
    # (that file is the long way's, `bench.py --no-filter`, from round 3; the default kernels' counter files are round 6's)
    for key in ('synth256_f64_column_filter', 'ar1_f64_column_filter', 'searev_f64_column_filter', 'synth512f32_f32_column_filter',
                'noisy256_f64_column_filter', 'reservoirs_f64_lead_filter', 'coupled256_f64_column', 'inventory1d_fine_f64_line_filter'):
        got = bench.load_pmc(key)[0]
        assert got is not None and got['tag'].startswith('r06'), key


def test_committed_clock_probe(bench):
    clock = bench.load_clock()
    assert 1.5 < clock['sweep_kernel_ghz'] <= 2.5         # (stamps within ~1 % of the 2.4 GHz peak clock)
    assert 1.5 < clock['long_way_kernel']['sweep_kernel_ghz'] <= 2.4
    assert 4.0 <= clock['fp64_clk_per_wave_instr_measured'] < 5.0
    with open(os.path.join(ROOT, 'profiles', 'clock.json')) as f:
        assert json.load(f)['method'].startswith('s_memtime')


def test_every_workload_names_a_model_builder(bench):
    from stodynprog_amd import models
    for name, (builder, kw, dtype, cfg, label) in bench.WORKLOADS.items():
        assert hasattr(models, builder) and dtype in ('float64', 'float32')
        assert cfg in (None, 1, 2, 3, 4)
    assert bench.WORKLOADS['synth256'][3] == 3               # the metric's configuration

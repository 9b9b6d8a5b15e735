"""CPU: bench.py's helpers that need no GPU — the code-object hash that ties roofline.traffic to the build, and the
argument surface (--scaling, --dry-rccl)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_code_object_hash_identifies_the_streaming_kernels():
    h = bench.code_object_hash()
    assert h and len(h) == 16 and h == bench.code_object_hash()
    assert bench.code_object_hash(b'rollout_kernel') not in (None, h)          # another translation unit, another code object
    assert bench.code_object_hash(b'no_such_kernel_anywhere') is None


def test_traffic_is_refused_when_measured_on_other_kernels(tmp_path, monkeypatch):
    """roofline.traffic comes from a PMC pass of an earlier build: it is reported only if profiles/traffic.json names the
    hash of THIS library's code object; otherwise None plus the reason."""
    prof = tmp_path / 'profiles'
    prof.mkdir()
    key = 'caltech_N65536_project1_compact'
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    (prof / 'traffic.json').write_text(json.dumps({key: {'hbm_bytes_per_launch': 123, 'source': 's', 'code_object_sha256': 'deadbeef'}}))
    val, why = bench.lookup_traffic('caltech', 65536, True, 'compact')
    assert val is None and 're-run tools/profile.sh' in why
    (prof / 'traffic.json').write_text(json.dumps({key: {'hbm_bytes_per_launch': 123, 'source': 's',
                                                         'code_object_sha256': bench.code_object_hash()}}))
    assert bench.lookup_traffic('caltech', 65536, True, 'compact') == (123, 's')


def test_traffic_is_per_step_and_tied_to_the_launch_form(tmp_path, monkeypatch):
    """A pipelined step (bench.py --pipeline 2) is two launches: the entry holds bytes per launch and how many launches make a
    step; the figure is refused for the other form."""
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    (prof / 'traffic.json').write_text(json.dumps({'caltech_N65536_project1_compact': {
        'hbm_bytes_per_launch': 100, 'launches_per_step': 2, 'source': 's', 'code_object_sha256': bench.code_object_hash()}}))
    assert bench.lookup_traffic('caltech', 65536, True, 'compact', 2) == (200, 's')
    val, why = bench.lookup_traffic('caltech', 65536, True, 'compact', 1)
    assert val is None and '2 launch(es) per step' in why


def test_committed_traffic_entry_matches_or_is_refused():
    val, why = bench.lookup_traffic('caltech', 65536, True, 'compact', 2)
    assert val is None or (isinstance(val, int) and val > 5e7)


def test_pipeline_arguments():
    a = bench.parse_args([])
    assert a.pipeline == 2 and not a.no_single_launch
    assert bench.parse_args(['--pipeline', '1', '--no-single-launch']).pipeline == 1


def test_argument_surface():
    a = bench.parse_args(['--gpus', '8', '--scaling', 'strong'])
    assert a.scaling == 'strong' and a.global_envs == 65536 and a.envs_per_gpu == 65536
    assert bench.parse_args([]).scaling == 'weak' and not bench.parse_args([]).dry_rccl
    assert bench.parse_args(['--dry-rccl']).dry_rccl and bench.parse_args(['--episodes', 'real']).episodes == 'real'


# ---- the driver's contract: ONE compact JSON line on stdout (a 20 KB line left BENCH_r05.parsed null) ----
def _canned_full_record():
    """Round 5's full 20 KB record (the line the driver could not parse), with the round-6 roofline keys added."""
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r5z_bench_driver.json')))
    rec['roofline'].update({'frac_hbm': 0.49, 'frac_survey': 0.74, 'frac_steady': 0.37, 'frac_survey_steady': 0.85,
                            'frac_hbm_timed_window': 0.43, 'algorithmic_bytes_per_env_step': 1017.0,
                            'survey_bytes_per_env_step': 2309, 'mean_entries_per_env': 5.0,
                            'window': 'achieved / frac / frac_survey: the timed region of ms_per_step'})
    rec['config']['launches_per_step'] = 2
    return rec


def test_headline_line_is_compact_and_complete(tmp_path, capsys):
    full = _canned_full_record()
    assert len(json.dumps(full)) > 16000                       # the record that broke the contract
    out = tmp_path / 'sub' / 'bench_full.json'
    line = bench.emit(full, str(out))
    cap = capsys.readouterr()
    # exactly one stdout line, it is the returned headline, it parses, and it is small
    assert cap.out == line + '\n' and '\n' not in line
    assert len(line) < bench.HEADLINE_MAX_BYTES <= 4096 < 8192
    head = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'ranks_seen', 'roofline', 'cpu_baseline', 'full_record'):
        assert key in head, key
        if key in full:
            assert head[key] == full[key] or isinstance(full[key], dict)
    assert head['config']['workload'] == full['config']['workload'] and 'model' not in head['config']
    for key in ('bound', 'kernel', 'frac_hbm', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_survey', 'avg_kernel_ms',
                'step_period_ms', 'single_launch'):
        assert key in head['roofline'], key
    assert list(head['roofline'])[:3] == ['bound', 'kernel', 'frac_hbm']          # the counter-based utilisation leads
    for key in ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread_value'):
        assert key in head['cpu_baseline'], key
    assert 'secondary' not in head and 'per_rank' not in head and 'episode_metrics' not in head
    assert head['secondary_scalars']['gmm_caltech_us_per_step'] == round(full['secondary']['gmm_caltech']['ms_per_step'] * 1e3, 2)
    # the full record: in the file the headline names, and on stderr behind a prefix no JSON-line parser mistakes for the headline
    assert json.load(open(out)) == full
    assert cap.err.startswith('bench.py full record: {') and json.loads(cap.err[len('bench.py full record: '):]) == full


def test_headline_survives_missing_legs_and_overlong_strings():
    full = _canned_full_record()
    full.update({'roofline': None, 'cpu_baseline': None, 'secondary': None, 'n_gpus': 8, 'ranks_seen': 8,
                 'per_rank': {'value': [1.0] * 8}, 'strong_scaling': {'global_envs': 65536, 'ms_per_step': 0.01, 'value': 1.0, 'scaling': 'strong'}})
    full['config']['workload'] = 'x' * 6000
    head = bench.headline_line(full, None)
    assert len(json.dumps(head)) <= bench.HEADLINE_MAX_BYTES
    assert head['roofline'] is None and head['cpu_baseline'] is None and head['n_gpus'] == 8 and head['value'] == full['value']


def test_algorithmic_floor_is_below_the_survey_figure_and_what_the_layout_moves():
    n, k = 54, 36
    assert bench.algorithmic_bytes_per_env_step(n, k) == 2309
    f0, f8 = bench.layout_floor_bytes_per_env_step(n, k, 0.0), bench.layout_floor_bytes_per_env_step(n, k, 8.0)
    assert f0 == 216 + 584 + 33 + 64 and f8 == f0 + 192 and f8 < 2309

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

"""SB3VecEnv's per-environment info objects (sustaingym_amd.envs._LazyInfo) on the CPU: a dict that reads the step's batch arrays on
access — the keys SB3 and its monitors look at, `.copy()` to a real dict, extra keys a wrapper stores, aliasing to the current step."""
import numpy as np

from sustaingym_amd.envs import _LazyInfo, _StepInfoSource


def _source():
    s = _StepInfoSource()
    s.max_profit = np.array([1.5, 2.5, 3.5])
    s.breakdown = np.arange(9.0).reshape(3, 3)
    s.done = np.array([False, True, False])
    s.final = {'timestep': np.ones((3, 1), np.float32), 'demands': np.arange(6, dtype=np.float32).reshape(3, 2)}
    return s


def test_items_come_from_the_batch_arrays():
    s = _source()
    infos = [_LazyInfo(s, i) for i in range(3)]
    assert all(isinstance(i, dict) for i in infos)
    assert infos[0]['max_profit'] == 1.5 and infos[2]['reward_breakdown'] == {'profit': 6.0, 'carbon_cost': 7.0, 'excess_charge': 8.0}
    assert infos[0].get('TimeLimit.truncated', True) is False and infos[0].get('episode') is None
    assert 'terminal_observation' in infos[1] and 'terminal_observation' not in infos[0]
    assert np.array_equal(infos[1]['terminal_observation']['demands'], [2.0, 3.0])
    assert sorted(infos[1].keys()) == ['TimeLimit.truncated', 'max_profit', 'reward_breakdown', 'terminal_observation']
    assert len(infos[0]) == 3 and dict(infos[0].items())['max_profit'] == 1.5
    try:
        infos[0]['terminal_observation']
        raise AssertionError('a running environment has no terminal observation')
    except KeyError:
        pass


def test_copy_is_a_real_dict_and_extra_keys_stay():
    s = _source()
    info = _LazyInfo(s, 1)
    c = info.copy()
    assert type(c) is dict and set(c) == {'max_profit', 'reward_breakdown', 'TimeLimit.truncated', 'terminal_observation'}
    c['episode'] = {'r': 1.0}                              # what VecMonitor does
    assert 'episode' not in info
    info['custom'] = 7
    assert info['custom'] == 7 and 'custom' in info and 'custom' in info.keys() and len(info) == 5
    # the object describes the CURRENT step: the source moves on, the object follows; the copy does not
    s.max_profit = np.array([9.0, 8.0, 7.0]); s.done = None; s.final = None
    assert info['max_profit'] == 8.0 and 'terminal_observation' not in info and c['max_profit'] == 2.5

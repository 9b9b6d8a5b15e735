"""Pipelined halves (evc_set_pipeline, include/evcharge.h): the batch stepped as two half launches on two internal
streams must leave exactly the state and outputs the single launch leaves — the environments are independent, only the
ORDER of launches changes — and every other entry point must see the halves joined."""
import numpy as np
import pytest

from sustaingym_amd.hostio import to_device, to_host
from helpers import make_workload

pytestmark = pytest.mark.gpu

N = 32768          # the smallest batches the engine splits: a half must still give every wavefront of the grid 4 quads


def _engine(net, wl, project, pipeline, N=N):
    from sustaingym_amd.engine import StepEngine
    P = len(wl['n_sessions'])
    eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=P, max_sessions=wl['sessions'].shape[1],
                     moer_days=wl['moer'].shape[0])
    eng.upload_moer(wl['moer'])
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
    eng.set_autoreset_stride(7)
    if pipeline == 2:
        eng.set_pipeline(2)
    return eng


def _state(eng):
    return eng.get_state()          # scalars, entries as station rows, reward breakdown accumulators


def _assert_same(a, b, tag):
    for k in a:
        assert np.array_equal(a[k], b[k]), (tag, k, np.argwhere(np.asarray(a[k]) != np.asarray(b[k]))[:4])


@pytest.mark.parametrize('site,project,busy,drain', [
    ('caltech', True, False, '1'),        # queue drained inside the half launches (the default only after a day of short queues)
    ('jpl', True, True, '1'),             # ... many queued environments per workgroup
    ('jpl', True, True, '0'),             # the slow kernel: each half has its own queue and its own slow launch behind it
    ('caltech', True, True, 'auto'),      # the engine's own choice, changing over the run
    ('caltech', False, False, '1'),       # no projection: no queue at all
])
def test_pipelined_halves_equal_the_single_launch(site, project, busy, drain, monkeypatch):
    import torch
    from sustaingym_amd.network import site_str_to_site
    monkeypatch.setenv('EVC_DRAIN', drain)
    net = site_str_to_site(site)
    n = net.num_stations
    wl = make_workload(net, N, bank_slots=1024, seed=5, busy=busy, moer_days=4)
    one, two = _engine(net, wl, project, 1), _engine(net, wl, project, 2)
    assert np.array_equal(to_host(one.reset()), to_host(two.reset()))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(3)
    ring = [torch.rand((N, n), dtype=torch.float32, device='cuda', generator=gen) for _ in range(6)]
    ring[2][:] = 1.0                                     # saturated period: rows bind, the slow path runs inside the halves
    step1, out1 = one.make_stepper()
    step2, out2 = two.make_stepper()
    T = 330                                              # past the episode boundary: autoreset inside pipelined launches
    for t in range(T):
        step1(ring[t % 6].data_ptr())
        step2(ring[t % 6].data_ptr())                    # no join in between: consecutive steps of the two halves overlap
    assert two.pipelined_steps() == T and one.pipelined_steps() == 0
    two.join()
    torch.cuda.synchronize()
    for k in out1:
        assert torch.equal(out1[k], out2[k]), k
    _assert_same(_state(one), _state(two), 'after the pipelined run')
    m1, m2 = one.read_metrics(), two.read_metrics()
    for k in m1:                                         # sums over the batch by float atomics: equal up to their order
        np.testing.assert_allclose(m1[k], m2[k], rtol=1e-12, err_msg=k)
    assert one.last_slow_count() == two.last_slow_count()          # the halves' queues add up to the single launch's

    # entry points in between join by themselves: a debug step (one launch), a partial reset, pipelined steps again, the
    # synchronous step() API (joins before it returns its tensors)
    ids = np.arange(100, 900, 3, dtype=np.int32)
    for eng in (one, two):
        eng.reset(env_ids=ids, slots=(ids % 1024).astype(np.int32))
    for t in range(5):
        step1(ring[t].data_ptr())
        step2(ring[t].data_ptr())
    a = ring[4]
    g1 = {k: to_host(v).copy() for k, v in one.step(a).items()}
    g2 = {k: to_host(v).copy() for k, v in two.step(a).items()}          # no explicit join
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k
    d1 = {k: to_host(v).copy() for k, v in one.step(torch.randint(0, 5, (N, n), device='cuda'), bins=5).items()}
    d2 = {k: to_host(v).copy() for k, v in two.step(torch.randint(0, 5, (N, n), device='cuda'), bins=5).items()}
    assert d1.keys() == d2.keys()                        # discrete steps are never split (different draws: only the plumbing)
    assert two.pipelined_steps() == T + 5 + 1
    one.close()
    two.close()


def test_pipelined_halves_against_the_oracle(monkeypatch):
    """The pipelined form on its own against the CPU oracle (not only against the single launch)."""
    import torch
    monkeypatch.setenv('EVC_DRAIN', '1')
    from oracle import binding as ob
    from sustaingym_amd.network import caltech_acn
    net = caltech_acn()
    n = net.num_stations
    wl = make_workload(net, N, bank_slots=512, seed=9, moer_days=3)
    eng = _engine(net, wl, True, 2)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'], wl['moer'], autoreset_stride=7)
    assert np.array_equal(to_host(eng.reset()), bat.reset())
    rng = np.random.default_rng(2)
    acts = [rng.random((N, n), dtype=np.float32) for _ in range(4)]
    dev = [to_device(a) for a in acts]
    step, out = eng.make_stepper()
    o = None
    for t in range(40):
        step(dev[t % 4].data_ptr())
        o = bat.step(acts[t % 4], debug=False)
    assert eng.pipelined_steps() == 40
    eng.join()
    torch.cuda.synchronize()
    g = {k: to_host(v) for k, v in out.items()}
    assert np.array_equal(g['terminated'], o['terminated'])
    assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:])
    np.testing.assert_allclose(g['obs'][:, :n], o['obs'][:, :n], rtol=2e-7)
    np.testing.assert_allclose(g['reward'], o['reward'], rtol=1e-9, atol=1e-13)
    rem, dep, est = eng.station_state()
    orem, odep, oest = bat.station_state()
    assert np.array_equal(dep, odep) and np.array_equal(est, oest)
    np.testing.assert_allclose(rem, orem, rtol=1e-12, atol=1e-12)
    eng.close()


def test_rollout_as_a_loop_of_steps_is_pipelined_too(monkeypatch):
    """evc_rollout replaying a pre-staged ring as a loop of evc_step launches (EVC_ROLLOUT_FUSED=0) goes through the same
    launch path: with the pipelined mode on its periods are half launches, and StepEngine.rollout joins before it returns."""
    import torch
    from sustaingym_amd.network import caltech_acn
    monkeypatch.setenv('EVC_DRAIN', '1')
    monkeypatch.setenv('EVC_ROLLOUT_FUSED', '0')
    net = caltech_acn()
    wl = make_workload(net, N, bank_slots=512, seed=12, moer_days=3)
    one, two = _engine(net, wl, True, 1), _engine(net, wl, True, 2)
    assert np.array_equal(to_host(one.reset()), to_host(two.reset()))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(8)
    ring = torch.rand((4, N, net.num_stations), dtype=torch.float32, device='cuda', generator=gen)
    o1 = one.rollout(actions=ring, steps=60, accumulate_returns=False)      # (returns are a debug-kernel output: never split)
    o2 = two.rollout(actions=ring, steps=60, accumulate_returns=False)
    torch.cuda.synchronize()
    assert two.pipelined_steps() == 60
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    _assert_same(_state(one), _state(two), 'after the looped rollout')
    one.close()
    two.close()


def test_host_step_waits_for_both_halves(monkeypatch):
    """evc_step_host after evc_set_pipeline(2) splits its launch like evc_step does; the copies back to
    the host must wait for BOTH half launches (they run on side streams the engine's stream does not see by itself), and a
    device-policy step (random actions: a kernel on the engine's stream that reads the environments' scalars) issued while
    halves are pending must be ordered behind them.  Host outputs of 40 pipelined numpy steps == the single-launch
    engine's, step by step, from pinned and from pageable action arrays."""
    import torch
    from sustaingym_amd.network import caltech_acn
    monkeypatch.setenv('EVC_DRAIN', '1')
    net = caltech_acn()
    n = net.num_stations
    wl = make_workload(net, N, bank_slots=512, seed=14, moer_days=3)
    one, two = _engine(net, wl, True, 1), _engine(net, wl, True, 2)
    assert np.array_equal(one.reset(host=True), two.reset(host=True))
    rng = np.random.default_rng(6)
    pinned = torch.empty((N, n), dtype=torch.float32).pin_memory().numpy()
    for t in range(40):
        a = rng.random((N, n), dtype=np.float32)
        if t % 2:
            pinned[:] = a
            a = pinned                                    # page-locked: the h2d copy leaves the engine's stream idle at once
        g1 = {k: v.copy() for k, v in one.step(a).items()}
        g2 = {k: v.copy() for k, v in two.step(a).items()}
        for k in g1:
            assert np.array_equal(g1[k], g2[k]), (t, k)
    assert two.pipelined_steps() == 40 and one.pipelined_steps() == 0
    # pipelined device steps left pending, then a random-policy step: its action kernel reads t / episode of every environment
    for eng in (one, two):
        eng.set_policy_seed(5, env_id_base=0)
    dev = to_device(rng.random((N, n), dtype=np.float32))
    s1, _ = one.make_stepper()
    s2, _ = two.make_stepper()
    for t in range(6):
        s1(dev.data_ptr())
        s2(dev.data_ptr())
    r1 = {k: np.array(v) for k, v in one.step_policy('random').items()}
    r2 = {k: np.array(v) for k, v in two.step_policy('random').items()}
    for k in r1:
        assert np.array_equal(r1[k], r2[k]), k
    _assert_same(_state(one), _state(two), 'after host steps and a random-policy step')
    one.close()
    two.close()


def test_closed_loop_with_per_half_policies_equals_the_single_launch(monkeypatch):
    """evc_pipeline_half: a policy that reads the step's observation (the caller's greedy, sign(demands)), enqueued per
    half on that half's stream, under pipelined steps WITHOUT any join — 300 steps across the autoreset boundary leave
    the state and outputs of the same closed loop run as one launch per step on one stream."""
    import torch
    from sustaingym_amd.network import caltech_acn
    monkeypatch.setenv('EVC_DRAIN', '1')
    net = caltech_acn()
    n = net.num_stations
    wl = make_workload(net, N, bank_slots=1024, seed=17, busy=True, moer_days=4)
    one, two = _engine(net, wl, True, 1), _engine(net, wl, True, 2)
    assert np.array_equal(to_host(one.reset()), to_host(two.reset()))
    s1, o1 = one.make_stepper()
    s2, o2 = two.make_stepper()
    a1 = torch.zeros((N, n), dtype=torch.float32, device='cuda')
    a2 = torch.zeros((N, n), dtype=torch.float32, device='cuda')
    halves = two.pipeline_halves()
    assert halves[0][0].start == 0 and halves[0][0].stop == halves[1][0].start and halves[1][0].stop == N
    torch.cuda.synchronize()
    for t in range(300):                                  # the reference loop first: the engines share torch's current stream,
        torch.sign(o1['obs'][:, :n], out=a1)              # and work pending there is what a pipelined step orders itself behind
        s1(a1.data_ptr())
    torch.cuda.synchronize()
    for t in range(300):
        for sl, st in halves:
            with torch.cuda.stream(st):
                torch.sign(o2['obs'][sl, :n], out=a2[sl])
        s2(a2.data_ptr())
    split, ordered = two.pipelined_steps(ordered=True)
    assert split == 300 and ordered <= 2                  # every step split; (almost) none had to wait for the engine's stream
    two.join()
    torch.cuda.synchronize()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(a1, a2)
    _assert_same(_state(one), _state(two), 'closed loop')
    one.close()
    two.close()


def test_closed_loop_per_half_policies_on_a_batch_the_engine_does_not_split(monkeypatch):
    """with 2 048 environments a step is ONE launch on the engine's stream although evc_set_pipeline(2) is on and
    evc_pipeline_half hands out the side streams.  The per-half contract must still hold (policy of step k+1 after step k's
    outputs, step k+1 after the policy): a slow policy on the side streams (a spin kernel in front of it) would otherwise
    lose the race for the action buffer."""
    import torch
    from sustaingym_amd.network import caltech_acn
    monkeypatch.setenv('EVC_DRAIN', '1')
    n_small = 2048
    net = caltech_acn()
    n = net.num_stations
    wl = make_workload(net, n_small, bank_slots=512, seed=19, busy=True, moer_days=4)
    one, two = _engine(net, wl, True, 1, n_small), _engine(net, wl, True, 2, n_small)
    assert np.array_equal(to_host(one.reset()), to_host(two.reset()))
    s1, o1 = one.make_stepper()
    s2, o2 = two.make_stepper()
    a1 = torch.zeros((n_small, n), dtype=torch.float32, device='cuda')
    a2 = torch.zeros((n_small, n), dtype=torch.float32, device='cuda')
    halves = two.pipeline_halves()
    torch.cuda.synchronize()
    for t in range(120):
        torch.sign(o1['obs'][:, :n], out=a1)
        s1(a1.data_ptr())
    torch.cuda.synchronize()
    for t in range(120):
        for sl, st in halves:
            with torch.cuda.stream(st):
                torch.cuda._sleep(200_000)                # ~0.1 ms: the step must wait for it
                torch.sign(o2['obs'][sl, :n], out=a2[sl])
        s2(a2.data_ptr())
    assert two.pipelined_steps() == 0                     # never split at this size
    two.join()
    torch.cuda.synchronize()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(a1, a2)
    _assert_same(_state(one), _state(two), 'closed loop, unsplit')
    one.close()
    two.close()

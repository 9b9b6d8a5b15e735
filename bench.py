#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched 54-station EVChargingEnv step() on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver
launches one rank per GPU with torch.distributed.run; a plain `python bench.py --gpus N` starts its N
ranks itself (one process per GPU, RCCL); a launcher world that is not N ranks is refused.  A "step" is
one pass of the hot path (EVChargingEnv.step: projection -> pilots -> ACN-Sim charge / event pass ->
observation -> reward) over one batch of environments with the actions already resident in HBM.
Environments are independent, so they shard over ranks with no data-path collective (weak scaling: fixed
envs per GPU); only the final metrics are all-gathered.

Steady state: the cost of a step follows the time of day of the simulated episodes (empty network at
night, busiest in the afternoon).  So that ANY K steps measure the average over a day rather than
whichever hours they happen to cover, the untimed set-up staggers the episode phases uniformly
(the four environments of wavefront q are q mod 288 periods into their day; `--phase sync` starts all episodes together
instead, as a freshly reset vector env does).

The JSON line carries, beside the headline: `roofline` (dominant kernel, per-launch HIP events),
`cpu_baseline` (the oracle's C restatement on the host cores, rank 0 at N = 1), and `secondary` —
the other regimes of BASELINE.json's configs, each outside the headline's timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec peak
EPISODE = 288


# algorithmic HBM bytes per env-step (station-shaped accounting), Caltech n=54, k=36:
#   read action 4n = 216; read+write station state {remaining f64, dep i16, est i16} 12n*2 = 1296;
#   event cursor + next event ~32; MOER row (k+1)*4 = 148; write obs (2n+k+2)*4 = 584;
#   write reward/done/breakdown 8+1+24 = 33   => 2309 B
def algorithmic_bytes_per_env_step(n: int, k: int) -> int:
    return 4 * n + 12 * n * 2 + 32 + (k + 1) * 4 + (2 * n + k + 2) * 4 + 33


def layout_floor_bytes_per_env_step(n: int, k: int, mean_entries: float) -> float:
    """What the COMPACT layout cannot avoid moving per env-step: the action row in (4n), the observation row out
    (4(2n+k+2)), reward / done / breakdown out (33), the two 16-byte scalar records in and out (64), and the plugged-in EVs'
    entries (float64 remaining demand + 4-byte entry word = 12 B) read and written back: 24 per entry.  The MOER row and the
    arriving session record are shared / L2-resident and not counted.  A floor: a rate computed on it cannot exceed the HBM rate
    by accounting alone (the SURVEY figure above counts 1 296 B of station-shaped state this layout never touches)."""
    return 4 * n + 4 * (2 * n + k + 2) + 33 + 64 + 24.0 * mean_entries


HEADLINE_MAX_BYTES = 4096       # the driver keeps a bounded tail of stdout: round 5's 20 KB line did not parse


def _pick(d, keys):
    return {k2: d[k2] for k2 in keys if isinstance(d, dict) and k2 in d and d[k2] is not None}


def headline_line(full: dict, full_path: str | None) -> dict:
    """The ONE stdout line of the contract, cut down from the full record to what the driver and the judge read first: metric,
    value, window, config, `roofline`, `cpu_baseline` and a handful of scalars.  Everything else (`secondary`, `per_rank`,
    `episode_metrics`, notes) is in the file `full_record` names and on stderr.  Never more than HEADLINE_MAX_BYTES."""
    cfg = full.get('config') or {}
    roof = full.get('roofline') or {}
    cpu = full.get('cpu_baseline') or {}
    line = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling'))
    line['vs_baseline'] = full.get('vs_baseline')
    line.update(_pick(full, ('dtype', 'data')))
    line['config'] = _pick(cfg, ('workload', 'envs_per_gpu', 'global_envs', 'parallelism', 'launches_per_step', 'settle_steps', 'host_wait'))
    line.update(_pick(full, ('ranks_seen', 'env_steps_timed')))
    if full.get('n_gpus', 1) > 1:
        line['per_rank_value'] = (full.get('per_rank') or {}).get('value')
        if full.get('strong_scaling'):
            line['strong_scaling'] = _pick(full['strong_scaling'], ('global_envs', 'ms_per_step', 'value'))
    r = _pick(roof, ('bound', 'kernel', 'frac_hbm', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'window', 'frac_steady',
                     'frac_survey', 'frac_survey_steady', 'frac_hbm_timed_window', 'algorithmic_bytes_per_env_step',
                     'survey_bytes_per_env_step', 'mean_entries_per_env', 'avg_kernel_ms', 'step_period_ms', 'launches_per_step',
                     'traffic_over_algorithmic'))
    if roof:
        r.setdefault('traffic', roof.get('traffic'))
        if isinstance(roof.get('single_launch'), dict):
            r['single_launch'] = _pick(roof['single_launch'], ('ms_per_step', 'avg_kernel_ms', 'frac', 'frac_survey'))
        line['roofline'] = r
    else:
        line['roofline'] = None
    line['cpu_baseline'] = _pick(cpu, ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread_value')) or None
    sec = full.get('secondary') or {}
    scal = {}
    for key in ('action_ring_32', 'gmm_caltech', 'gmm_jpl', 'real_caltech'):
        if isinstance(sec.get(key), dict) and 'ms_per_step' in sec[key]:
            scal[key + '_us_per_step'] = round(sec[key]['ms_per_step'] * 1e3, 2)
    for key in ('rollout_greedy_65536_gmm', 'rollout_random_65536_gmm'):
        if isinstance(sec.get(key), dict) and 'env_steps_per_s' in sec[key]:
            scal[key + '_env_steps_per_s'] = sec[key]['env_steps_per_s']
    va = sec.get('vector_env_api') or {}
    for key in ('torch', 'torch_pipeline2', 'torch_caller_greedy', 'torch_policy_greedy', 'torch_pipeline2_policy_greedy'):
        if isinstance(va.get(key), dict) and 'ms_per_step' in va[key]:
            scal['vector_env_' + key + '_us_per_step'] = round(va[key]['ms_per_step'] * 1e3, 2)
    if scal:
        line['secondary_scalars'] = scal
    line['full_record'] = full_path
    # belt and braces: whatever a future edit adds, the line stays inside the budget
    for drop in ('secondary_scalars', 'per_rank_value', 'strong_scaling'):
        if len(json.dumps(line)) <= HEADLINE_MAX_BYTES:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > HEADLINE_MAX_BYTES:
        line['config'] = _pick(line['config'], ('envs_per_gpu', 'global_envs', 'parallelism'))
        line['config']['workload'] = str(cfg.get('workload', ''))[:200]
        if line.get('cpu_baseline'):
            line['cpu_baseline']['sample'] = str(line['cpu_baseline'].get('sample', ''))[:120]
    assert len(json.dumps(line)) <= HEADLINE_MAX_BYTES
    return line


def emit(full: dict, out_path: str | None) -> str:
    """Writes the full record to `out_path` (and, prefixed so that no line-oriented parser takes it for the headline, to stderr),
    then prints the compact headline as the LAST and ONLY stdout line.  Returns that line."""
    text = json.dumps(full)
    written = None
    if out_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
            with open(out_path, 'w') as fh:
                fh.write(text + '\n')
            written = os.path.relpath(out_path, ROOT) if os.path.abspath(out_path).startswith(ROOT) else out_path
        except OSError as exc:
            sys.stderr.write(f'bench.py: could not write {out_path}: {exc}\n')
    sys.stderr.write('bench.py full record: ' + text + '\n')
    sys.stderr.flush()
    line = json.dumps(headline_line(full, written))
    sys.stdout.write(line + '\n')
    sys.stdout.flush()
    return line


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=576)
    p.add_argument('--warmup', type=int, default=288)
    p.add_argument('--envs-per-gpu', type=int, default=65536)
    p.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                   help="'weak' (default): --envs-per-gpu environments on every GPU; 'strong': --global-envs environments "
                        'split over the GPUs (north_star: 65 536 batched environments on 8 GPUs).  An N > 1 weak run also '
                        "times the strong form afterwards and reports it as 'strong_scaling'")
    p.add_argument('--global-envs', type=int, default=65536, help='total environments of the strong-scaling form')
    p.add_argument('--dry-rccl', action='store_true',
                   help="initialise the 'nccl' (= RCCL) process group with world size 1 on cuda:0, all-gather the metrics "
                        'vector, print one JSON line and exit (run in a child process by the N = 1 bench)')
    p.add_argument('--site', default='caltech', choices=['caltech', 'jpl'])
    p.add_argument('--no-project', action='store_true', help='project_action_in_env=False')
    p.add_argument('--bank', type=int, default=8192, help='distinct episodes resident in HBM')
    p.add_argument('--ring', type=int, default=8, help='distinct action batches resident in HBM')
    p.add_argument('--pipeline', type=int, default=2, choices=[1, 2],
                   help='2 (default): the batch stepped as two half-batch launches on two streams whose tails overlap the '
                        "next launches (evc_set_pipeline; the action ring is resident, nothing reads an output between steps); "
                        '1: one launch per step, every step ordered on one stream')
    p.add_argument('--no-single-launch', action='store_true',
                   help='skip the one-launch-per-step comparison leg of a pipelined run (profiling passes: keeps every dispatch '
                        'of the streaming kernel a half launch)')
    p.add_argument('--phase', default='stagger', choices=['stagger', 'sync'],
                   help="'stagger' (default): episode phases spread uniformly over the day, every step costs the "
                        "day's average; 'sync': all episodes start together")
    p.add_argument('--battery', default='continuous', choices=['continuous', 'stepwise'],
                   help="acnportal Linear2StageBattery(charge_calculation=...); 'continuous' is acnportal's default")
    p.add_argument('--busy', action='store_true',
                   help='congested variant of the workload (30-60 long sessions per day); not the headline')
    p.add_argument('--episodes', default='synthetic', choices=['synthetic', 'gmm', 'real'],
                   help="'gmm': the bank is generated on the device from the reference's GMM (Summer 2019); 'real': every "
                        'ACN-Data day of Summer 2021 (RealTraceBank); not the headline (secondary records)')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help="process-group backend for --gpus > 1 ('nccl' = RCCL; 'gloo' only to exercise the N > 1 "
                        'logic with several ranks on ONE GPU, see --single-device)')
    p.add_argument('--single-device', action='store_true',
                   help='testing aid: every rank uses cuda:0 (needs --backend gloo)')
    p.add_argument('--settle-steps', type=int, default=EPISODE,
                   help='untimed steps at the end of the set-up, in front of the --warmup steps: SURVEY 8(d) defines the metric in steady '
                        'state "excluding one warm-up episode", and with a short --warmup (the driver: 5) the timed window would otherwise '
                        "begin on a GPU that has idled through the set-up's host work (clocks down).  Default: one episode")
    p.add_argument('--host-wait', default='auto', choices=['spin', 'auto'],
                   help="how the rank's host thread waits in torch.cuda.synchronize(): the runtime's choice (default) or 'spin' "
                        '(hipDeviceScheduleSpin; measured: no difference, profiles/r6_spin_ab2.txt)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-secondary', action='store_true', help='skip the secondary records (GMM days, multi-agent, battery)')
    p.add_argument('--leg-budget-s', type=float, default=60.0, help='time box of one secondary record')
    p.add_argument('--secondary-budget-s', type=float, default=240.0, help='time box of all secondary records together')
    p.add_argument('--cpu-envs', type=int, default=8192)
    p.add_argument('--cpu-steps', type=int, default=96, help='minimum timed steps of the cpu_baseline sample')
    p.add_argument('--full-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_full.json'),
                   help="file that receives the full record (secondary legs, per-rank data, notes); the stdout line names it; '' = none")
    p.add_argument('--kernel-timing-steps', type=int, default=288,
                   help='launches timed one by one (HIP events on the engine stream) for the roofline leg')
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# self-spawn: `python bench.py --gpus N` without a launcher
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    """Starts n copies of this command, one rank per GPU, and relays rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        sys.stderr.write(f'bench.py: ranks exited with {rcs}\n')
        return 1
    last = None
    for ln in out0.splitlines():            # the process-group backend may chat on stdout; the contract is ONE JSON line
        if ln.startswith('{') and ln.rstrip().endswith('}'):
            last = ln
    if last is not None:
        sys.stdout.write(last + '\n')
    return 0


# ------------------------------------------------------------------------------------------------
# one EV-charging workload on one GPU
# ------------------------------------------------------------------------------------------------
class EvWorkload:
    """Engine + episode bank + action ring for N environments on `dev`, phases set up as asked."""

    def __init__(self, site, N, dev_index, rank, project=True, episodes='synthetic', bank=8192, ring=8, busy=False,
                 phase='stagger', battery='continuous', seed_base=1000, pipeline=2):
        import torch
        from sustaingym_amd.engine import StepEngine
        from sustaingym_amd.network import site_str_to_site
        from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
        self.torch = torch
        self.net = net = site_str_to_site(site)
        self.site, self.N, self.n, self.k, self.project = site, N, net.num_stations, 36, project
        self.dev = torch.device('cuda', dev_index)
        self.P = P = min(bank, max(N, 1))
        self.moer_days = 32
        self.moer = synthetic_moer(self.moer_days, seed=7)
        real = None
        if episodes == 'real':            # every day of the packaged period, with the period's own MOER matrices
            from datetime import timedelta
            from sustaingym_amd.event_generation import RealTraceBank
            real = RealTraceBank(site, 'Summer 2021')
            self.P = P = real.num_days_in_date_range
            self.moer_days = P
            self.moer = np.stack([real.moer_loader.retrieve(real.date_range[0] + timedelta(days=d)) for d in range(P)])
        self.eng = eng = StepEngine(net, N, moer_forecast_steps=self.k, project_action=project, autoreset=True,
                                    device=dev_index, bank_slots=P, max_sessions=128 if episodes in ('gmm', 'real') else 64,
                                    moer_days=self.moer_days, charge_calculation=battery)
        eng.upload_moer(self.moer)
        if real is not None:
            self.bank = (real.n_sessions, real.sessions, real.requested, real.moer_day)
            eng.upload_episodes(*self.bank)
        elif episodes == 'gmm':
            from sustaingym_amd.event_generation import gmm_device_tables
            eng.upload_gmm(dict(gmm_device_tables(site, 'Summer 2019'), num_days=self.moer_days))
            eng.generate_episodes(0, P, seed_base + rank, 0)
            self.bank = eng.download_episodes(0, P)[:4]                     # for the cpu_baseline leg
        else:
            kw = dict(min_sessions=30, max_sessions=60, max_arrival=120, min_duration=40, max_duration=160) if busy else {}
            self.bank = synthetic_episodes(P, self.n, seed=seed_base + rank, stride=64, moer_days=self.moer_days, **kw)
            eng.upload_episodes(*self.bank)
        eng.set_autoreset_stride(1)
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(1234 + rank)
        self.ring = [torch.rand((N, self.n), dtype=torch.float32, device=self.dev, generator=gen) for _ in range(ring)]
        self.ptrs = [t.data_ptr() for t in self.ring]
        eng.reset()
        self.step, self.out = eng.make_stepper()
        # pipelined halves (evc_set_pipeline): the action ring is staged once, nothing reads an output between steps, so the
        # two half-batches may run ahead of each other; every reader below joins first (any engine call does)
        self.pipeline = pipeline
        if pipeline == 2:
            eng.set_pipeline(2)
        self._i = 0
        if phase == 'stagger':
            self._stagger()

    def run(self, steps: int) -> None:
        ptrs, step = self.ptrs, self.step
        for _ in range(steps):
            step(ptrs[self._i % len(ptrs)])
            self._i += 1

    def _stagger(self) -> None:
        """Untimed set-up: after 287 steps in which group s — the quads (4 consecutive environments = one
        wavefront of the streaming kernel) with quad index = s mod 288 — is reset after step s, the environments
        of quad q are 287 - (q mod 288) periods into their episodes: every period of the day is present in every
        launch from here on (and stays so: all episodes last 288 periods), each wavefront's four environments share
        a phase as they do in a synchronised vector env, and the quads a wavefront visits (stride 512) are spread
        over the day — so a launch costs the average over a synchronised day."""
        N = self.N
        quad = np.arange(N, dtype=np.int64) // 4
        for s in range(1, EPISODE):
            self.run(1)
            ids = np.flatnonzero(quad % EPISODE == s).astype(np.int32)
            if len(ids):
                self.eng.reset(env_ids=ids, slots=(ids % self.P).astype(np.int32))

    def time_kernels(self, launches: int) -> dict:
        """Per-launch begin-to-end durations (ms) of the streaming and the slow kernel (evc_enable_timing)."""
        eng = self.eng
        eng.enable_timing(True)
        main_ms, slow_ms, slow_cnt = [], [], []
        for _ in range(launches):
            self.run(1)
            a, b = eng.last_step_ms()
            main_ms.append(a)
            slow_ms.append(b)
            if self.project:
                slow_cnt.append(eng.last_slow_count())
        eng.enable_timing(False)
        res = {'main_ms': np.array(main_ms), 'slow_ms': np.array(slow_ms),
               'slow_envs': float(np.mean(slow_cnt)) if slow_cnt else 0.0, 'period_ms': None, 'half_ms': None}
        before = eng.pipelined_steps()
        self.run(8)
        if eng.pipelined_steps() - before == 8:
            # Pipelined halves: a step is two launches that overlap the neighbouring steps' — what the GPU needs per step is
            # the PERIOD of the launch train, measured with HIP events on the stream that joins it (windows of M = 288 steps;
            # the first record makes the stream busy, so the window's first step is ordered behind it like a caller's
            # actions would be), and a half launch's own begin-to-end time under that overlap (what a kernel trace shows).
            torch = self.torch
            M = 288
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            periods, halves = [], []
            for _ in range(max(3, launches // M)):
                e0.record()
                self.run(M)
                eng.join()
                e1.record()
                e1.synchronize()
                periods.append(e0.elapsed_time(e1) / M)
            eng.enable_timing(True)
            for _ in range(24):
                self.run(6)
                halves.extend(eng.last_half_ms())
            eng.enable_timing(False)
            res['period_ms'], res['half_ms'] = np.array(periods), np.array(halves)
        return res

    def wall_ms_per_step(self, steps: int) -> float:
        torch = self.torch
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        self.run(steps)
        self.host_issue_ms_per_step = (time.perf_counter() - t0) / steps * 1e3      # how long the host took to enqueue them
        self.eng.join()
        torch.cuda.synchronize(self.dev)
        return (time.perf_counter() - t0) / steps * 1e3

    def close(self) -> None:
        self.eng.close()


def code_object_hash(symbol: bytes = b'step_kernel_cquad') -> str | None:
    """sha256 (first 16 hex digits) of the gfx950 code object of the built library that holds `symbol`: what ties a
    PMC-derived traffic figure to the kernels that produced it."""
    import hashlib
    import struct
    from sustaingym_amd import _lib
    try:
        data = open(_lib.LIB_PATH, 'rb').read()
    except OSError:
        return None
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    at = data.find(magic)
    while at >= 0:
        n_entries = struct.unpack_from('<Q', data, at + len(magic))[0]
        pos = at + len(magic) + 8
        for _ in range(n_entries):
            off, size, tlen = struct.unpack_from('<QQQ', data, pos)
            triple = data[pos + 24:pos + 24 + tlen]
            pos += 24 + tlen
            if b'gfx950' in triple:
                blob = data[at + off:at + off + size]
                if symbol in blob:
                    return hashlib.sha256(blob).hexdigest()[:16]
        at = data.find(magic, at + 1)
    return None


def lookup_traffic(site, N, project, layout, launches_per_step=1):
    """HBM bytes per launch of the streaming kernel from the PMC passes of tools/profile.sh (profiles/traffic.json:
    FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs, corrected as DESIGN.md §6 describes).  The entry records
    the hash of the code object it was measured on: a library built from different kernels gets `None` (and the
    reason) instead of a stale figure."""
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        tj = json.load(open(tpath))
        key = f'{site}_N{N}_project{int(project)}' + ('' if layout == 'dense' else f'_{layout}')
        ent = tj.get(key, {})
        have = code_object_hash()
        if ent and ent.get('code_object_sha256') != have:
            return None, f"profiles/traffic.json was measured on code object {ent.get('code_object_sha256')}, this library is {have}: re-run tools/profile.sh"
        per_launch, lps = ent.get('hbm_bytes_per_launch'), ent.get('launches_per_step', 1)
        if per_launch is None:
            return None, ent.get('source')
        if lps != launches_per_step:
            return None, f'profiles/traffic.json was measured with {lps} launch(es) per step, this run uses {launches_per_step}: re-run tools/profile.sh'
        return per_launch * lps, ent.get('source')        # bytes per STEP (= per launch when a step is one launch)
    except Exception:
        return None, None


def roofline_record(w: EvWorkload, timed: dict, survey_bytes_per_env_step: int, window_ms: float | None = None,
                    mean_entries: float | None = None) -> dict:
    """The streaming kernel against the 8 TB/s HBM peak, per STEP of w.N environments, in three accountings:

    `frac_hbm`    what the PMC counters say moved (profiles/traffic.json, tied to this code object) over the steady step
                  period — the bandwidth utilisation;
    `frac`        ALGORITHMIC bytes over the same window `ms_per_step` is timed in.  Algorithmic = what the compact layout
                  cannot avoid moving (layout_floor_bytes_per_env_step; round 6): a lower bound on the bytes, so
                  `frac` cannot pass 1 by accounting alone and `traffic / algorithmic` >= 1 shows the re-reads;
    `frac_survey` SURVEY §8(d)'s station-shaped 2 309 B per env-step on that window, the `frac` of rounds 1-5, kept for
                  continuity: it counts 1 296 B of station-shaped state this layout never touches — a throughput index
                  (env-steps/s x 2 309 B / 8 TB/s), not a bandwidth."""
    layout = os.environ.get('EVC_LAYOUT', 'compact')          # engine default
    pipelined = timed.get('period_ms') is not None
    # one launch per step: the launch's own duration.  Pipelined halves: the period of the launch train (a step's two
    # launches overlap the neighbouring steps', so a launch's own duration is not what the GPU needs per step)
    avg = float(timed['period_ms'].mean()) if pipelined else float(timed['main_ms'].mean())
    window = avg if window_ms is None else window_ms
    survey = survey_bytes_per_env_step * w.N
    floor_b = layout_floor_bytes_per_env_step(w.n, w.k, mean_entries) if (mean_entries is not None and layout == 'compact') \
        else float(survey_bytes_per_env_step)
    alg = floor_b * w.N

    def frac(nbytes, ms):
        return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    traffic, source = lookup_traffic(w.site, w.N, w.project, layout, 2 if pipelined else 1)
    rec = {'bound': 'hbm', 'kernel': 'evc::step_kernel_cquad' if layout == 'compact' else 'evc::step_kernel_quad',
           'frac_hbm': frac(traffic, avg) if traffic else None,
           'achieved': round(alg / (window * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': frac(alg, window),
           'traffic': traffic,
           'window': ('achieved / frac / frac_survey: the timed region of ms_per_step' if window_ms is not None
                      else 'achieved / frac / frac_survey: steady step period') + '; frac_hbm / frac_steady: steady step period (HIP events)'
                     + "; frac_survey prices SURVEY's station-shaped 2309 B per env-step, more than this layout moves: it can pass 1"
                     + '; traffic / frac_hbm are L2 <-> fabric bytes, part of them served by the 256 MB Infinity Cache (config.actions)',
           'frac_steady': frac(alg, avg),
           'frac_survey': frac(survey, window), 'frac_survey_steady': frac(survey, avg),
           'frac_hbm_timed_window': frac(traffic, window) if traffic else None,
           'algorithmic_bytes_per_env_step': round(floor_b, 1), 'survey_bytes_per_env_step': survey_bytes_per_env_step,
           'mean_entries_per_env': None if mean_entries is None else round(mean_entries, 3),
           'traffic_gbs': round(traffic / (avg * 1e-3) / 1e9, 2) if traffic else None,
           'traffic_over_algorithmic': round(traffic / alg, 4) if traffic else None,
           'traffic_over_survey': round(traffic / survey, 4) if traffic else None,
           'traffic_source': source,
           'state_layout': layout, 'avg_kernel_ms': round(avg, 5),
           'kernel_ms_min_max': [round(float(timed['main_ms'].min()), 5), round(float(timed['main_ms'].max()), 5)],
           'kernel_launches_timed': int(len(timed['main_ms'])),
           'solver_kernel_ms': round(float(timed['slow_ms'].mean()), 5),
           'slow_queue_envs_per_step': round(timed['slow_envs'], 1),
           'algorithmic_bytes_per_launch': int(alg)}
    if pipelined:
        h = timed['half_ms']
        rec.update({
            'launches_per_step': 2, 'algorithmic_bytes_per_launch': int(alg) // 2, 'algorithmic_bytes_per_step': int(alg),
            'basis': 'A step is two half-batch launches on two streams (evc_set_pipeline) that overlap the neighbouring steps\' '
                     'launches; step_period_ms = HIP events on the joining stream over windows of 288 steps; half_launch_ms = one '
                     'launch\'s own begin-to-end time under that overlap (hipExtLaunchKernel events; what a kernel trace reports '
                     'per dispatch); step_alone_ms = first begin to last end of a step issued alone',
            'step_period_ms': round(avg, 5), 'avg_kernel_ms': round(float(h.mean()), 5),
            'kernel_ms_min_max': [round(float(h.min()), 5), round(float(h.max()), 5)], 'kernel_launches_timed': int(len(h)),
            'half_launch_ms': round(float(h.mean()), 5),
            'step_alone_ms': round(float(timed['main_ms'].mean()), 5),
            'step_period_ms_min_max': [round(float(timed['period_ms'].min()), 5), round(float(timed['period_ms'].max()), 5)]})
    return rec


def mean_entries_per_env(w: EvWorkload) -> float:
    """Plugged-in EVs per environment right now, from the step's own observation rows (demands > 0, env.py:381-394): the Ā of
    layout_floor_bytes_per_env_step.  With staggered phases every period of the day is present, so this is the day's average."""
    w.eng.join()
    w.torch.cuda.synchronize(w.dev)
    return float((w.out['obs'][:, :w.n] > 0).sum(dim=1).double().mean().item())


def cpu_baseline_record(args, w: EvWorkload, acts) -> dict:
    """The oracle (scalar C restatement, oracle/) on the host cores: bounded sample of the same workload."""
    from oracle import binding as ob
    cn, cs = min(args.cpu_envs, w.N), args.cpu_steps
    ns, sess, req, day = w.bank
    bat = ob.OracleBatch(ob.OracleNetwork(w.net), cn, w.k, w.project, args.battery)
    bat.set_bank(ns, sess, req, day, w.moer, autoreset_stride=1)
    bat.reset(np.arange(cn, dtype=np.int32) % w.P)
    try:
        quota = open('/sys/fs/cgroup/cpu.max').read().split()
        cgroup_cpus = None if quota[0] == 'max' else round(int(quota[0]) / int(quota[1]), 2)
    except Exception:
        cgroup_cpus = None
    host = {'cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)), 'cgroup_cpus': cgroup_cpus,
            'omp_max_threads': ob.max_threads()}
    cores = ob.default_threads()        # one thread per CPU this process may really use (cgroup quota)
    # skip the empty early-morning periods so that the sample has plugged-in EVs; their rate sizes the timed
    # sample to ~3 s of wall time on whatever host this is (bounded: --cpu-steps .. 2304 steps = 8 days)
    t1 = time.perf_counter()
    for i in range(96):
        bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=cores)
    rate0 = cn * 96 / (time.perf_counter() - t1)
    cs = int(min(2304, max(cs, 3.0 * rate0 / cn)))
    t1 = time.perf_counter()
    for i in range(cs):
        bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=cores)
    dt = time.perf_counter() - t1
    t1 = time.perf_counter()                # the same sample continued on one thread (SURVEY §8d asks for both), ~2 s
    for i in range(4):
        bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=1)
    c1 = int(min(96, max(4, 2.0 / ((time.perf_counter() - t1) / 4))))
    t1 = time.perf_counter()
    for i in range(c1):
        bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=1)
    dt1 = time.perf_counter() - t1
    return {'value': round(cn * cs / dt, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{cn} envs x {cs} steps (from period 97 on, across autoresets) of the same workload, {dt:.1f} s, '
                      f'oracle/ C restatement ({args.battery} battery), OpenMP over envs with {cores} threads',
            'single_thread_value': round(cn * c1 / dt1, 1),
            'single_thread_sample': f'{cn} envs x {c1} steps, {dt1:.1f} s, 1 thread', 'host': host}


# ------------------------------------------------------------------------------------------------
# secondary records: the other regimes BASELINE.json's configs name (rank 0, outside the timed region)
# ------------------------------------------------------------------------------------------------
def secondary_days(site, episodes, dev_index, battery) -> dict:
    """65 536 environments on the reference's own episode distributions: episodes = 'gmm' — days sampled on the device
    from the packaged GMM (Summer 2019 model, GMMsTraceGenerator) — or 'real' — every ACN-Data day of Summer 2021
    (RealTraceGenerator, sequential; real MOER).  Busier than the headline's synthetic days: pods and feeders bind
    around midday, the slow path takes part."""
    w = EvWorkload(site, 65536, dev_index, 0, project=True, episodes=episodes, phase='sync', battery=battery)
    w.run(EPISODE)                                   # one untimed day; the timed one starts at an episode boundary
    wall = w.wall_ms_per_step(EPISODE)               # one whole synchronised day (what a vector env plays)
    split = w.eng.pipelined_steps()
    w.eng.set_pipeline(1)                            # the same day as one launch per step, and its kernels by time of day
    wall1 = w.wall_ms_per_step(EPISODE)
    timed = w.time_kernels(EPISODE)
    # plugged-in EVs per environment, averaged over the day (24 samples of one more day): the A of the layout's byte floor
    abar = float(np.mean([(w.run(12), mean_entries_per_env(w))[1] for _ in range(EPISODE // 12)]))
    survey = algorithmic_bytes_per_env_step(w.n, w.k)
    alg = layout_floor_bytes_per_env_step(w.n, w.k, abar)
    what = 'device-generated GMM days' if episodes == 'gmm' else f'the {w.P} ACN-Data days of Summer 2021 (RealTraceBank), real MOER'
    rec = {'workload': f'65536 x {w.n}-station ({site}) on {what}, projection on, U[0,1) actions, synchronised episodes, mean over one whole day',
           'ms_per_step': round(wall, 5), 'env_steps_per_s': round(65536 / wall * 1e3, 1),
           'pipelined_steps_of_the_day': int(split - EPISODE) if split >= EPISODE else int(split),
           'single_launch': {'ms_per_step': round(wall1, 5), 'env_steps_per_s': round(65536 / wall1 * 1e3, 1),
                             'note': 'evc_set_pipeline(1): one launch per step; the kernel times below are of this form'},
           'kernel_us': round(float(timed['main_ms'].mean()) * 1e3, 2),
           'kernel_us_by_4h': [round(float(x.mean()) * 1e3, 1) for x in np.array_split(timed['main_ms'], 6)],
           'solver_kernel_us': round(float(timed['slow_ms'].mean()) * 1e3, 2),
           'solver_kernel_us_by_4h': [round(float(x.mean()) * 1e3, 1) for x in np.array_split(timed['slow_ms'], 6)],
           'slow_queue_envs_per_step': round(timed['slow_envs'], 1),
           'roofline': {'bound': 'hbm', 'algorithmic_bytes_per_env_step': round(alg, 1), 'mean_entries_per_env': round(abar, 3),
                        'survey_bytes_per_env_step': survey,
                        'achieved': round(alg * 65536 / (wall * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(alg * 65536 / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        'frac_survey': round(survey * 65536 / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        'note': 'on the whole step (streaming + slow kernel), not on one kernel; algorithmic = the compact layout\'s '
                                'byte floor (layout_floor_bytes_per_env_step), frac_survey = the station-shaped SURVEY 8d figure of rounds 1-5'}}
    w.close()
    return rec


def secondary_rollout(policy, dev_index, battery, episodes='synthetic', site='caltech') -> dict:
    """Whole episodes under a device-resident policy (BaseAlgorithm.run over GreedyAlgorithm / RandomAlgorithm,
    algorithms/base.py:63-88, baselines.py:22-51): 65 536 environments x 288 periods in ONE launch of the fused rollout
    kernel (csrc/evc_rollout.h: state in registers, no per-period action read / observation write), beside the same
    rollout as a loop of evc_step launches and the oracle's C episode loop on a bounded sample of the same episodes."""
    import torch
    from oracle import binding as ob
    from sustaingym_amd.hostio import to_host
    N = 65536
    w = EvWorkload(site, N, dev_index, 0, project=True, episodes=episodes, phase='sync', battery=battery)
    eng = w.eng
    eng.set_policy_seed(7)
    dev = w.dev

    def episodes_per_call(fused, reps):
        os.environ['EVC_ROLLOUT_FUSED'] = '1' if fused else '0'
        try:
            for _ in range(8 if fused else 1):                         # untimed: first launches, bank wrap, and — synchronised calls,
                eng.rollout(policy=policy, steps=EPISODE)              # like an evaluation loop's — what the engine needs to pick the
                torch.cuda.synchronize(dev)                            # faster register budget of the kernel (evc_last_rollout_waves)
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.rollout(policy=policy, steps=EPISODE)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / reps
        finally:
            os.environ.pop('EVC_ROLLOUT_FUSED', None)
    fused = episodes_per_call(True, 5)
    waves_chosen = eng.last_rollout_waves()
    eng.enable_timing(True)
    eng.rollout(policy=policy, steps=EPISODE)
    kernel_ms = eng.last_step_ms()[0]
    eng.enable_timing(False)
    loop = episodes_per_call(False, 2)
    # CPU: the oracle's episode loop on the first 4096 episodes of the bank
    cn = 4096
    ns, sess, req, day = w.bank
    bat = ob.OracleBatch(ob.OracleNetwork(w.net), cn, w.k, True, battery)
    bat.set_bank(ns, sess, req, day, w.moer, autoreset_stride=1)
    obs0 = bat.reset(np.arange(cn, dtype=np.int32) % w.P)
    cores = ob.default_threads()
    t0 = time.perf_counter()
    bat.rollout(policy, obs0, steps=EPISODE, seed=7, threads=cores)
    cpu = time.perf_counter() - t0
    w.close()
    # The fused kernel has no HBM traffic to speak of (one MOER value and the arriving session records per environment-period):
    # its bound is VALU issue.  Instructions per env-step come from the SQ counters of tools/profile_rollout.sh
    # (profiles/r3_rollout_*.json); peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md).
    valu = valu_src = None
    for rnd in ('r6', 'r5', 'r3'):                        # the newest committed SQ-counter record of this workload
        try:
            valu_src = f'profiles/{rnd}_rollout_{site}_{"synthetic" if episodes == "synthetic" else "gmm"}_{policy}.json'
            valu = json.load(open(os.path.join(ROOT, valu_src)))['SQ_INSTS_VALU']
            break
        except Exception:
            valu_src = None
    peak = 256 * 4 * 2.4e9 / 2
    roof = None if valu is None else {'bound': 'valu', 'valu_instructions_per_env_step': valu,
                                      'achieved': round(N * EPISODE / (kernel_ms * 1e-3) * valu / 1e9, 1), 'peak': round(peak / 1e9, 1),
                                      'unit': 'G wave-instructions/s', 'frac': round(N * EPISODE / (kernel_ms * 1e-3) * valu / peak, 4),
                                      'instructions_from': valu_src,
                                      'note': 'float64 instructions issue at half this rate; SQ_ACTIVE_INST_VALU says the vector ALUs are busy ~68 % of the launch (DESIGN.md 4.6)'}
    return {'workload': f'{N} x {w.n}-station ({site}), {"synthetic days" if episodes == "synthetic" else "device-generated GMM days"}, '
                        f'projection on, {policy} policy on the device, whole episodes (288 periods), autoreset',
            'env_steps_per_s': round(N * EPISODE / fused, 1), 'episode_ms': round(fused * 1e3, 4),
            'us_per_period': round(fused / EPISODE * 1e6, 3), 'launches_per_episode': 1,
            'kernel': 'evc::rollout_kernel', 'kernel_ms': round(kernel_ms, 4), 'waves_per_simd_chosen': waves_chosen, 'roofline': roof,
            'loop_of_steps': {'env_steps_per_s': round(N * EPISODE / loop, 1), 'episode_ms': round(loop * 1e3, 3),
                              'launches_per_episode': EPISODE * (2 if policy == 'random' else 1) + 0,
                              'note': 'EVC_ROLLOUT_FUSED=0: evc_step per period (random: + the action kernel), observations written every period'},
            'cpu_port': {'value': round(cn * EPISODE / cpu, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
                         'sample': f'{cn} episodes x 288 periods, oracle/ orc_batch_rollout, {cpu:.1f} s'}}


def secondary_vector_env_api(dev_index, battery) -> dict:
    """north_star's API surface, boundaries included: steps through EVChargingVectorEnv.step (Gymnasium VectorEnv semantics)
    with episodes from the on-device GMM generator — (a) torch in / torch out, nothing leaves the GPU; (b) the numpy path
    SB3 / RLlib use (actions and observations cross PCIe every step; never the headline `value`)."""
    import torch
    from sustaingym_amd.envs import EVChargingVectorEnv
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    out = {}
    N = 65536
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch',
                               device=dev_index, charge_calculation=battery)
    venv.reset(seed=0)
    dev = torch.device('cuda', dev_index)
    acts = torch.rand((N, venv.num_stations), device=dev)
    for _ in range(EPISODE):
        venv.step(acts)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2 * EPISODE):                       # two whole episodes: two boundaries (bank refill, max_profit download)
        venv.step(acts)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / (2 * EPISODE)
    venv.close()
    out['torch'] = {'workload': f'{N} x 54-station (caltech) EVChargingVectorEnv.step, torch tensors, DeviceGMMTraceGenerator, two episodes incl. boundaries',
                    'ms_per_step': round(dt * 1e3, 5), 'env_steps_per_s': round(N / dt, 1),
                    'note': 'every environment plays its own episode (2 x 65 536 bank slots, 268 MB of session tables): the arrival loads miss '
                            'the caches that the 8 192-episode bank of secondary.gmm_* lives in'}
    # the same through the opt-in pipelined form (pipeline=2: two half-batch launches per step, step() does not join; round 4)
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch',
                               device=dev_index, charge_calculation=battery, pipeline=2)
    venv.reset(seed=0)
    for _ in range(EPISODE):
        venv.step(acts)
    venv.join()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2 * EPISODE):
        venv.step(acts)
    venv.join()
    torch.cuda.synchronize(dev)
    dt2 = (time.perf_counter() - t0) / (2 * EPISODE)
    venv.close()
    out['torch_pipeline2'] = {'workload': 'the same, EVChargingVectorEnv(..., pipeline=2)', 'ms_per_step': round(dt2 * 1e3, 5),
                              'env_steps_per_s': round(N / dt2, 1)}
    # closed loop under greedy (GreedyAlgorithm, baselines.py:22-35), two episodes: (a) the caller computes sign(obs['demands'])
    # with one torch kernel per step and hands the tensor in; (b) step(policy='greedy'): the streaming kernels apply the rule
    # themselves — no action tensor, no policy kernel, the step stays two pipelined halves (round 6)
    for tag, pipe in (('', 1), ('_pipeline2', 2)):
        for form in ('caller_greedy', 'policy_greedy'):
            venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch',
                                       device=dev_index, charge_calculation=battery, pipeline=pipe)
            obs, _ = venv.reset(seed=0)
            a = torch.zeros((N, venv.num_stations), device=dev)

            def one(obs):
                if form == 'policy_greedy':
                    return venv.step(policy='greedy')[0]
                if pipe == 2:
                    for sl, st in halves:
                        with torch.cuda.stream(st):
                            torch.sign(obs['demands'][sl], out=a[sl])
                else:
                    torch.sign(obs['demands'], out=a)
                return venv.step(a)[0]
            halves = venv.pipeline_halves() if pipe == 2 else None
            for _ in range(EPISODE):
                obs = one(obs)
            venv.join()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2 * EPISODE):
                obs = one(obs)
            venv.join()
            torch.cuda.synchronize(dev)
            dtg = (time.perf_counter() - t0) / (2 * EPISODE)
            venv.close()
            out[f'torch{tag}_{form}'] = {'workload': f'the same environments under greedy, {form.replace("_", " ")}'
                                                     + (', pipeline=2' if pipe == 2 else ''),
                                         'ms_per_step': round(dtg * 1e3, 5), 'env_steps_per_s': round(N / dtg, 1)}
    Nn = 16384
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=Nn, output='numpy',
                               zero_copy=True, device=dev_index, charge_calculation=battery)
    venv.reset(seed=0)
    a = np.random.default_rng(0).random((Nn, venv.num_stations), dtype=np.float32)
    for _ in range(10):
        venv.step(a)
    t0 = time.perf_counter()
    for _ in range(100):
        venv.step(a)
    dt = (time.perf_counter() - t0) / 100
    venv.close()
    out['sb3_numpy_path'] = {'workload': f'{Nn} x 54-station (caltech) EVChargingVectorEnv.step, numpy in / numpy out (page-locked, zero_copy): '
                                         f'{a.nbytes / 1e6:.1f} MB of actions in and {Nn * 146 * 4 / 1e6:.1f} MB of observations out per step over PCIe',
                             'ms_per_step': round(dt * 1e3, 5), 'env_steps_per_s': round(Nn / dt, 1), 'pcie_inclusive': True}
    return out


def secondary_closed_loop(dev_index, battery) -> dict:
    """Closed loop (train_stable_baselines.py:271-275): a device policy that READS the observation of step k
    to produce the action of step k + 1 — the caller's own greedy, ``sign(demands)`` as a torch op — on the headline workload
    (65 536 Caltech environments, synthetic days, projection on, staggered phases).  Forms: (a) one launch per step, the
    policy on the engine's stream; (b) pipelined halves with a join after every step (what a caller that treats the step as
    synchronous gets); (c) pipelined halves with each half's policy on that half's stream (evc_pipeline_half): the policy of
    one half runs under the other half's step, no join; (d) EVChargingVectorEnv.step(output='torch') on device-generated GMM
    days, synchronous and with pipeline=2 + per-half policies.  Beside them the open-loop figure of the same engine."""
    import torch
    N = 65536
    dev = torch.device('cuda', dev_index)
    out = {}
    w = EvWorkload('caltech', N, dev_index, 0, project=True, battery=battery, pipeline=1)
    n = w.n
    eng = w.eng
    acts = torch.zeros((N, n), dtype=torch.float32, device=dev)
    obs = w.out['obs']
    demands = obs[:, :n]
    ptr = acts.data_ptr()
    K = 576

    def timed(body, join=False, steps=K):
        for _ in range(48):
            body()
        eng.join()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            body()
        host = (time.perf_counter() - t0) / steps
        eng.join()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        return {'ms_per_step': round(dt * 1e3, 5), 'env_steps_per_s': round(N / dt, 1), 'host_issue_us_per_step': round(host * 1e6, 2)}

    # Two policies, one torch kernel each: 'min_u' = min(demands, u) with u the headline's U[0,1) action ring — a true function
    # of the observation whose values equal the headline's wherever an EV still wants more than 1 kWh, so the WORKLOAD is the
    # headline's and the difference to the open loop is the loop itself; 'greedy' = sign(demands), the reference's
    # GreedyAlgorithm computed by the caller (baselines.py:32-35) — every EV asks for 32 A, pods bind in most periods: a
    # heavier workload, listed for what it is.
    ring = w.ring
    state = {'i': 0}

    ring_views = {None: ring}

    def policy(kind, d, a, sl=None):
        if kind == 'greedy':
            torch.sign(d, out=a)
        else:
            rv = ring_views[sl]
            torch.minimum(d, rv[state['i'] % len(rv)], out=a)

    halves = None
    for kind in ('min_u', 'greedy'):
        rec = {}
        eng.set_pipeline(1)

        def single():
            policy(kind, demands, acts)
            w.step(ptr)
            state['i'] += 1
        rec['single_launch'] = dict(timed(single), form='one launch per step, policy on the engine stream')
        eng.set_pipeline(2)
        if halves is None:
            halves = eng.pipeline_halves()
            views = [(demands[sl], acts[sl], h, st) for h, (sl, st) in enumerate(halves)]
            for h, (sl, st) in enumerate(halves):
                ring_views[h] = [r[sl] for r in ring]          # views built once: the loop below is host-bound otherwise
            main_stream = torch.cuda.current_stream(dev)

        def joined():
            policy(kind, demands, acts)
            w.step(ptr)
            eng.join()
            state['i'] += 1
        rec['pipelined_joined'] = dict(timed(joined, steps=192), form='two half launches per step, evc_join after every step, policy on the engine stream')

        def per_half():
            for d, a, h, st in views:
                torch.cuda.set_stream(st)            # (the `with torch.cuda.stream(...)` form costs ~6 us of host time more per half)
                policy(kind, d, a, h)
            torch.cuda.set_stream(main_stream)
            w.step(ptr)
            state['i'] += 1
        before = eng.pipelined_steps(ordered=True)
        rec['pipelined_per_half_policy'] = dict(timed(per_half), form='two half launches per step, each half\'s policy enqueued on that '
                                                'half\'s stream (evc_pipeline_half): no join, no fork event')
        after = eng.pipelined_steps(ordered=True)
        rec['pipelined_per_half_policy']['pipelined_steps'] = int(after[0] - before[0])
        rec['pipelined_per_half_policy']['steps_ordered_behind_the_engine_stream'] = int(after[1] - before[1])
        out['policy_' + kind] = rec
    # what the policy costs by itself: its kernel over the whole batch, back to back on one stream (14 MB of ring in, the demand
    # columns of the 38 MB observation — every 128-byte line they touch — and 14 MB of actions out, per step)
    for kind in ('min_u', 'greedy'):
        def only_policy():
            policy(kind, demands, acts)
            state['i'] += 1
        out['policy_' + kind]['policy_kernel_alone'] = timed(only_policy)
    eng.set_pipeline(1)
    out['open_loop_single_launch'] = timed(lambda: w.step(w.ptrs[0]))
    eng.set_pipeline(2)
    out['open_loop_pipelined'] = timed(lambda: w.step(w.ptrs[0]))
    w.close()

    # the API north_star names, on the reference's own episode distribution
    from sustaingym_amd.envs import EVChargingVectorEnv
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    for pipeline in (1, 2):
        venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch',
                                   device=dev_index, charge_calculation=battery, pipeline=pipeline)
        o, _ = venv.reset(seed=0)
        d = o['demands']
        a = torch.zeros((N, venv.num_stations), dtype=torch.float32, device=dev)
        if pipeline == 2:
            vv = [(d[sl], a[sl], st) for sl, st in venv.pipeline_halves()]

            def body():
                for dd, aa, st in vv:
                    with torch.cuda.stream(st):
                        torch.sign(dd, out=aa)
                venv.step(a)
        else:
            def body():
                torch.sign(d, out=a)
                venv.step(a)
        for _ in range(EPISODE):
            body()
        venv.join()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(2 * EPISODE):
            body()
        venv.join()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / (2 * EPISODE)
        venv.close()
        out[f'vector_env_api_pipeline{pipeline}'] = {
            'workload': f'{N} x 54-station (caltech) EVChargingVectorEnv.step(output=torch, pipeline={pipeline}), DeviceGMMTraceGenerator, '
                        'greedy computed by the caller from obs[demands], two episodes incl. boundaries',
            'ms_per_step': round(dt * 1e3, 5), 'env_steps_per_s': round(N / dt, 1)}
    out['workload'] = f'{N} x {n}-station (caltech), synthetic days, projection on, staggered phases; policy reads the step\'s observation'
    return out


def dry_rccl() -> int:
    """`bench.py --dry-rccl`: the N > 1 bench's process-group plumbing with ONE rank — init 'nccl' (= RCCL) on cuda:0,
    all-gather a metrics-sized vector, barrier, destroy.  Prints one JSON line."""
    import torch
    import torch.distributed as dist
    from sustaingym_amd.distributed import all_gather_vector, max_over_ranks
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    t0 = time.perf_counter()
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    vec = all_gather_vector(np.arange(9, dtype=np.float64), dev)
    mx = max_over_ranks(1.5, dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    ok = vec.shape == (1, 9) and float(vec[0, 8]) == 8.0 and mx == 1.5
    backend = dist.get_backend()
    dist.destroy_process_group()
    print(json.dumps({'rccl_world1': bool(ok), 'backend': backend, 'seconds': round(time.perf_counter() - t0, 2),
                      'nccl_version': list(torch.cuda.nccl.version()) if hasattr(torch.cuda, 'nccl') else None}))
    return 0 if ok else 1


def secondary_rccl_world1() -> dict:
    """Runs `bench.py --dry-rccl` in a child process (a fault inside RCCL must not cost the headline)."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--dry-rccl'], capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {'rccl_world1': False, 'error': 'timeout after 240 s'}
    for ln in r.stdout.splitlines():
        if ln.startswith('{'):
            return json.loads(ln)
    return {'rccl_world1': False, 'returncode': r.returncode, 'stderr_tail': r.stderr[-400:]}


def secondary_tie_snap(dev_index, battery) -> dict:
    """Reach of the tie snap on the reference's own episode distribution: one whole GMM day of
    16 384 Caltech environments through the kernels WITH per-station outputs (they count every value a
    projection solver moves and how many of those lay within 1e-6 A of a rounding boundary before the snap)."""
    import torch
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.event_generation import gmm_device_tables
    from sustaingym_amd.network import site_str_to_site
    from sustaingym_amd.synthetic import synthetic_moer
    net = site_str_to_site('caltech')
    N = 16384
    eng = StepEngine(net, N, project_action=True, autoreset=True, device=dev_index, bank_slots=N, max_sessions=128,
                     moer_days=32, debug_outputs=True, charge_calculation=battery)
    eng.upload_moer(synthetic_moer(32, seed=7))
    eng.upload_gmm(dict(gmm_device_tables('caltech', 'Summer 2019'), num_days=32))
    eng.generate_episodes(0, N, 4242, 0)
    eng.reset()
    g = torch.Generator(device=torch.device('cuda', dev_index))
    g.manual_seed(77)
    ring = [torch.rand((N, net.num_stations), device=torch.device('cuda', dev_index), generator=g) for _ in range(8)]
    for t in range(EPISODE):
        eng.step(ring[t % 8])
    met = eng.read_metrics()
    eng.close()
    pilots = N * EPISODE * net.num_stations
    return {'workload': f'{N} x 54-station (caltech), one GMM day, projection on, U[0,1) actions',
            'pilot_values': pilots, 'solver_moved_values': int(met['solver_moved_values']),
            'within_1e-6A_of_a_rounding_boundary': int(met['tie_snap_near_boundary']),
            'fraction_of_moved': round(met['tie_snap_near_boundary'] / max(1.0, met['solver_moved_values']), 8),
            'fraction_of_all_pilots': round(met['tie_snap_near_boundary'] / pilots, 10)}


def secondary_action_ring(site, dev_index, battery, project, slices=32) -> dict:
    """The headline workload with an action ring that does NOT fit the 256 MB Infinity Cache: the default ring (8 slices x N x n
    floats = 113 MB at 65 536 x 54) comes round every eight steps and is served from that cache together with the observations and
    the state (~180 MB in all); with 32 slices (450 MB) every action row of every step is read from HBM.  A policy that writes its
    actions on the device each step is the first case; this leg is the second."""
    w = EvWorkload(site, 65536, dev_index, 0, project=project, battery=battery, ring=slices)
    w.run(2 * EPISODE)
    ms = w.wall_ms_per_step(2 * EPISODE)
    w.eng.set_pipeline(1)
    w.run(64)
    single = w.wall_ms_per_step(EPISODE)
    w.close()
    return {'workload': f'65536 x {w.n}-station ({site}), as the headline, action ring of {slices} slices ({slices * 65536 * w.n * 4 / 1e6:.0f} MB: every row from HBM)',
            'ms_per_step': round(ms, 5), 'env_steps_per_s': round(65536 / (ms * 1e-3), 1), 'single_launch_ms_per_step': round(single, 5)}


def secondary_sync_reference(site, dev_index, battery, project) -> dict:
    """The headline workload with SYNCHRONISED episode phases (what a freshly reset vector env plays, and what round 1's
    bench timed): the whole-day average, and the 20 steps after 5 warm-up steps of a fresh day — the driver's window,
    which with synchronised phases covers periods 5..25 of the day (night: no EV plugged in yet, the cheapest steps)."""
    w = EvWorkload(site, 65536, dev_index, 0, project=project, phase='sync', battery=battery)
    w.run(EPISODE)                                   # first day: the engine still runs the slow kernel
    w.run(5)
    night = w.wall_ms_per_step(20)
    w.run(EPISODE - 25)
    day = w.wall_ms_per_step(EPISODE)
    issue = w.host_issue_ms_per_step
    w.close()
    return {'workload': f'65536 x {w.n}-station ({site}), synthetic days, synchronised episodes',
            'whole_day': {'ms_per_step': round(day, 5), 'env_steps_per_s': round(65536 / day * 1e3, 1),
                          'host_issue_ms_per_step': round(issue, 5)},
            'periods_5_to_25': {'ms_per_step': round(night, 5), 'env_steps_per_s': round(65536 / night * 1e3, 1),
                                'note': "round 1's BENCH line (2.31e9) timed this window"}}


def secondary_small_configs(dev_index, battery) -> dict:
    """BASELINE configs[0] and configs[1]: ONE EVChargingEnv behind the Gymnasium API with DiscreteActionWrapper (host round trip
    per step: evc_step_host's direct mode — the kernels write the page-locked numpy buffers themselves), and 4 096 batched
    environments with continuous actions on the device path (launch-bound: one quad per wavefront)."""
    import torch
    from sustaingym_amd import DiscreteActionWrapper, EVChargingEnv, GMMsTraceGenerator
    out = {}
    env = DiscreteActionWrapper(EVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2021'), device=dev_index, charge_calculation=battery))
    env.reset(seed=0)
    acts = np.random.default_rng(0).integers(0, 5, (288, 54))
    for t in range(32):
        env.step(acts[t])
    ts = []
    for t in range(32, 288):
        t0 = time.perf_counter()
        env.step(acts[t])
        ts.append(time.perf_counter() - t0)
    env.close()
    out['config0_single_env_gymnasium'] = {'workload': 'one EVChargingEnv (caltech, GMM day), DiscreteActionWrapper(bins=5), projection on, numpy in / dict of numpy out',
                                           'us_per_step_median': round(float(np.median(ts)) * 1e6, 1), 'us_per_step_mean': round(float(np.mean(ts)) * 1e6, 1),
                                           'env_steps_per_s': round(1.0 / float(np.median(ts)), 1), 'pcie_inclusive': True}
    # the stable_baselines3 VecEnv adapter itself (train_stable_baselines.py:271-275's SubprocVecEnv replaced): 4 096 environments,
    # numpy in / numpy out + the per-environment info objects SB3 asks for
    from sustaingym_amd.envs import EVChargingVectorEnv, SB3VecEnv
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    for mode, copy_obs in (('lazy', True), ('lazy', False), ('dicts', True)):
        sb3 = SB3VecEnv(EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=0), num_envs=4096, device=dev_index,
                                            charge_calculation=battery), infos=mode, copy_obs=copy_obs)
        sb3.reset()
        a = np.random.default_rng(0).random((4096, 54), dtype=np.float32)
        reps = 40 if mode == 'lazy' else 6
        for _ in range(4):
            sb3.step(a)
        t0 = time.perf_counter()
        for _ in range(reps):
            sb3.step(a)
        dt = (time.perf_counter() - t0) / reps
        sb3.close()
        out[f'sb3_vecenv_4096_infos_{mode}' + ('' if copy_obs else '_zero_copy_obs')] = {'ms_per_step': round(dt * 1e3, 4), 'env_steps_per_s': round(4096 / dt, 1), 'pcie_inclusive': True}
    N = 4096
    for project in (True, False):
        w = EvWorkload('caltech', N, dev_index, 0, project=project, bank=1024, phase='stagger', battery=battery, pipeline=1)
        w.run(64)
        wall = w.wall_ms_per_step(2 * EPISODE)
        w.close()
        out[f'config1_4096_device_project{int(project)}'] = {'workload': f'{N} x 54-station (caltech), U[0,1) actions resident in HBM, project_action_in_env={project}, one launch per step',
                                                              'ms_per_step': round(wall, 5), 'env_steps_per_s': round(N / wall * 1e3, 1)}
    return out


def secondary_multiagent(dev_index, battery) -> dict:
    """BASELINE configs[4]: 8 192 environments x 54 agents.  'view' = zero-copy [N, n, F] broadcast of the flat
    observation (what the reference's multiagent_env.py:114-117 hands out: the same array for every agent);
    'materialised' = the gather kernel writes all 258 MB of per-agent rows, 'delay3' additionally takes the
    other agents' entries from the observation 3 periods ago (documented semantics)."""
    import torch
    N = 8192
    w = EvWorkload('caltech', N, dev_index, 0, project=True, phase='stagger', battery=battery)
    n, F = w.n, 2 * w.n + w.k + 2
    eng = w.eng
    w.run(16)
    wall_view = w.wall_ms_per_step(EPISODE)
    obs = w.out['obs']
    buf = torch.empty((N, n, F), dtype=torch.float32, device=w.dev)
    old = obs.clone()
    out = {}
    for name, delayed in (('materialised', None), ('delay3', old)):
        for _ in range(8):
            w.run(1)
            eng.gather_agent_obs(obs, delayed, buf)
        torch.cuda.synchronize(w.dev)
        t0 = time.perf_counter()
        for _ in range(96):
            w.run(1)
            eng.gather_agent_obs(obs, delayed, buf)
        torch.cuda.synchronize(w.dev)
        wall = (time.perf_counter() - t0) / 96 * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.gather_agent_obs(obs, delayed, buf)
        e1.record()
        torch.cuda.synchronize(w.dev)
        gms = e0.elapsed_time(e1) / 20
        bytes_gather = N * n * F * 4 + N * F * 4 * (2 if delayed is not None else 1)
        out[name] = {'ms_per_step': round(wall, 5), 'agent_steps_per_s': round(N * n / wall * 1e3, 1),
                     'gather_kernel_us': round(gms * 1e3, 2),
                     'roofline': {'bound': 'hbm', 'kernel': 'evc::gather_agent_obs_kernel',
                                  'algorithmic_bytes_per_env_step': n * F * 4,
                                  'achieved': round(bytes_gather / (gms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                  'unit': 'GB/s', 'frac': round(bytes_gather / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    rec = {'workload': f'{N} x {n} agents (caltech), synthetic days, projection on',
           'view': {'ms_per_step': round(wall_view, 5), 'agent_steps_per_s': round(N * n / wall_view * 1e3, 1)}, **out}
    w.close()
    return rec


def secondary_battery(dev_index, N=16384) -> dict:
    """BASELINE configs[3] (one GPU's share, 16 384 of 131 072 environments): the synthetic battery-dispatch
    step of include/battery_dispatch.h (no reference implementation exists)."""
    import torch
    from sustaingym_amd.battery import BatteryDispatchVectorEnv, synthetic_market_traces
    k = 36
    env = BatteryDispatchVectorEnv(N, k, bank_slots=1024, device=dev_index, output='torch')
    env.upload_traces(synthetic_market_traces(1024, k, seed=3))
    env.reset(np.arange(N) % 1024)
    dev = torch.device('cuda', dev_index)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    bids = [torch.rand((N, 2 * k), device=dev, generator=g) * 90.0 for _ in range(4)]
    ptrs = [b.data_ptr() for b in bids]
    step, _ = env.make_stepper()           # env.step() itself is 12.7 us of Python per call: more than the kernel runs
    for i in range(32):
        step(ptrs[i % 4])
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(400):
        step(ptrs[i % 4])
    e1.record()
    issue = (time.perf_counter() - t0) / 400 * 1e3
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / 400 * 1e3
    gpu = e0.elapsed_time(e1) / 400
    # the floor of ANY back-to-back kernel launch from this process, measured the same way: a one-element torch kernel
    one = torch.zeros(1, device=dev)
    for _ in range(32):
        one.add_(1.0)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(200):
        one.add_(1.0)
    e1.record()
    torch.cuda.synchronize(dev)
    empty = e0.elapsed_time(e1) / 200
    alg = (4 * k + 6) * 4 + 2 * k * 4 + 8 + 1 + 2 * 16          # obs row + bids + reward + done + state r/w
    env.close()
    return {'workload': f'{N} battery-dispatch envs (synthetic price-taker step), k={k}',
            'parity': 'synthetic specification (no ElectricityMarketEnv code in the reference): oracle parity at N = 37 / 1 000 / 4 096 '
                      '(tests/test_gpu_battery.py), property checks only at this size',
            'ms_per_step': round(wall, 5), 'env_steps_per_s': round(N / wall * 1e3, 1), 'gpu_ms_per_step': round(gpu, 5),
            'roofline': {'bound': 'hbm', 'kernel': 'bat::step_kernel', 'algorithmic_bytes_per_env_step': alg,
                         'achieved': round(alg * N / (gpu * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(alg * N / (gpu * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         'note': 'launch-latency bound at this size (940 B x 16 384 = 15 MB per launch): compare empty_launch_ms'},
            'host_issue_ms_per_step': round(issue, 5),
            'empty_launch_ms': round(empty, 5), 'over_empty_launch': round(gpu / empty, 2)}


def secondary_battery_rollout(dev_index, N=16384, T=EPISODE) -> dict:
    """the battery step as T periods per launch (bat_rollout): bids from a device-resident ring, every step's
    observation and reward written to a trajectory buffer (what a learner's rollout buffer holds) — the traffic of 288 calls of
    bat_step without 288 launch boundaries — and the form without the trajectory (last outputs only)."""
    import torch
    from sustaingym_amd.battery import BatteryDispatchVectorEnv, synthetic_market_traces
    k, R = 36, 8
    env = BatteryDispatchVectorEnv(N, k, bank_slots=1024, device=dev_index, output='torch')
    env.upload_traces(synthetic_market_traces(1024, k, seed=3))
    dev = torch.device('cuda', dev_index)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ring = (torch.rand((R, N, 2 * k), device=dev, generator=g) * 90.0).contiguous()
    F = 4 * k + 6
    rew_traj = torch.empty((T, N), dtype=torch.float64, device=dev)
    # the trajectory as BatteryDispatchVectorEnv.rollout allocates it by default — rows 160 floats (640 B) apart, the
    # [T, N, 150] view handed out (bat_rollout_pitched, round 5) — and with packed 600-byte rows (round 4's form)
    bufs = {'with_trajectory': env.new_trajectory(T),          # rows 640 B apart, padding owned by the kernel: whole-line stores
            'with_trajectory_packed_rows': (torch.empty((T, N, F), dtype=torch.float32, device=dev), rew_traj),
            'last_outputs_only': None}
    slots = np.arange(N) % 1024
    out = {}
    for name, traj in bufs.items():
        trajectory = traj is not None
        ms = []
        for rep in range(6):
            env.reset(slots)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.rollout(ring, T, trajectory=trajectory, out=traj if trajectory else None)
            e1.record()
            torch.cuda.synchronize(dev)
            ms.append(e0.elapsed_time(e1))
        best = float(np.median(ms[1:]))
        alg = 2 * k * 4 + 12 + ((4 * k + 6) * 4 + 8 if trajectory else 0)       # bids + traces in; observation + reward out
        out[name] = {'episode_ms': round(best, 4), 'us_per_step': round(best / T * 1e3, 3), 'env_steps_per_s': round(N * T / best * 1e3, 1),
                     'roofline': {'bound': 'hbm', 'kernel': 'bat_rollout_kernel', 'algorithmic_bytes_per_env_step': alg,
                                  'achieved': round(alg * N * T / (best * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': round(alg * N * T / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    env.close()
    out['workload'] = f'{N} battery-dispatch envs x {T} steps in one launch, bids from a ring of {R} batches, k={k}; trajectory = [{T}, {N}, {4 * k + 6}] float32 + rewards'
    return out


class LegTimeout(Exception):
    pass


def run_leg(fn, budget_s: float, deadline: float):
    """One secondary record, time-boxed: SIGALRM after `budget_s` seconds (or at the overall deadline of the
    secondary section, whichever is sooner) raises inside the leg's Python loop; a leg that fails, overruns or is skipped
    leaves an `error` record and never costs the headline line.  (A kernel that never returns would still block inside the
    HIP call: the alarm bounds host loops and waits the interpreter returns from, which is every wait bench.py has.)"""
    import signal
    left = deadline - time.monotonic()
    if left <= 1.0:
        return {'error': 'skipped: the secondary section used up --secondary-budget-s'}

    def on_alarm(signum, frame):
        raise LegTimeout()
    old = signal.signal(signal.SIGALRM, on_alarm)
    t0 = time.monotonic()
    signal.setitimer(signal.ITIMER_REAL, min(budget_s, left))
    try:
        rec = fn()
        if isinstance(rec, dict):
            rec['leg_seconds'] = round(time.monotonic() - t0, 2)
        return rec
    except LegTimeout:
        err = f'timed out after {time.monotonic() - t0:.1f} s (--leg-budget-s {budget_s})'
    except Exception as exc:          # a secondary record must never cost the headline
        err = f'{type(exc).__name__}: {exc}'
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, old)
    # the leg was cut off mid-loop: whatever engines and environments its frames held go away NOW (their finalisers close the
    # HIP engines and free their buffers) instead of riding along under the legs that follow
    import gc
    gc.collect()
    try:
        import torch
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    except Exception:
        pass
    return {'error': err}


def main():
    args = parse_args()
    if args.dry_rccl:
        sys.exit(dry_rccl())
    from sustaingym_amd.distributed import WorldMismatch, resolve_world
    try:
        rank, local_rank, world, must_spawn = resolve_world(args.gpus, os.environ)
    except WorldMismatch as exc:
        sys.stderr.write(f'bench.py: {exc}\n')
        sys.exit(2)
    if must_spawn:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    host_wait = 'auto'
    if args.host_wait == 'spin' and os.environ.get('BENCH_SPIN_WAIT', '1') != '0':
        # Measurement switch: the rank's host thread SPINS at its synchronisations (hipDeviceScheduleSpin on the rank's own device; set
        # before torch touches the device, through the very runtime library torch has loaded — its bundled copy where there is one: a
        # second copy of the HIP runtime in the process must not happen).  A first A/B seemed to gain 1.1 us per step of the driver's
        # window (profiles/r6_spin_ab.txt) — but that arm had dlopen'ed the SYSTEM libamdhip64 (ROCm 7.2) before torch, so torch ran on it
        # instead of its bundled 7.0 copy: the runtime, not the wait mode.  On one runtime the two modes are level (r6_spin_ab2.txt).
        import ctypes
        try:
            bundled = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
            hip = ctypes.CDLL(bundled if os.path.exists(bundled) else 'libamdhip64.so')
            dev_index = 0 if args.single_device else local_rank
            rc = hip.hipSetDevice(ctypes.c_int(dev_index)) or hip.hipSetDeviceFlags(ctypes.c_uint(0x1))
            host_wait = 'spin' if rc == 0 else f'auto (hipSetDeviceFlags -> {rc})'
        except OSError as exc:
            host_wait = f'auto ({exc})'
    if args.single_device:
        assert args.backend == 'gloo', '--single-device needs --backend gloo (RCCL wants one GPU per rank)'
        local_rank = 0
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        sys.stderr.write(f'bench.py: rank {rank} wants cuda:{local_rank} but only {ndev} device(s) are visible\n')
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
        assert dist.get_world_size() == world == args.gpus
    dev = torch.device('cuda', local_rank)
    coll_dev = dev if (world == 1 or args.backend == 'nccl') else torch.device('cpu')   # where collectives run

    from sustaingym_amd.distributed import all_gather_vector, max_over_ranks, metrics_vector

    N = args.envs_per_gpu if args.scaling == 'weak' else max(4, args.global_envs // world)
    project = not args.no_project
    w = EvWorkload(args.site, N, local_rank, rank, project=project, episodes=args.episodes, bank=args.bank,
                   ring=args.ring, busy=args.busy, phase=args.phase, battery=args.battery, pipeline=args.pipeline)
    n, k = w.n, w.k

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # counters are read BEFORE the warm-up steps: nothing but the barrier stands between the warm-up and the timed steps
    # (a metrics kernel + copy there leaves the GPU idle for a few hundred microseconds more)
    # (same-box A/B, 10 interleaved runs of the 20-step window: 26.0 against 27.2 us per step pipelined, no difference with one launch per step)
    w.run(args.settle_steps)                     # set-up: one untimed episode (steady state, clocks up); not part of --warmup / --steps
    if args.settle_steps > 0:
        # ... and the host runtime's housekeeping for those hundreds of launches, which it does in the first calls BEHIND the next
        # synchronisation (tools/probes/host_issue_profile.py: ~20 us per step() call for the first ten instead of 8): absorbed here
        # by a synchronisation and sixteen untimed steps, not by the first steps of a short timed window
        barrier()
        w.run(16)
        barrier()
    steps0 = w.eng.read_metrics()['env_steps'] + float(N) * args.warmup
    # (the interpreter's cyclic garbage collector stays out of the timed steps, as in timeit: a generation-2 pass over this
    # process's heap takes milliseconds, and the device runs only ~200 us behind the host's launches.  Collected and switched
    # off BEFORE the warm-up steps: milliseconds of idle GPU between warm-up and timed steps cost the window 1.2 us per step)
    import gc
    gc.collect()
    gc.disable()
    w.run(args.warmup)
    barrier()
    pipelined0 = w.eng.pipelined_steps()           # a host-side counter: no engine work
    t0 = time.perf_counter()
    w.run(args.steps)
    host_issue_s = time.perf_counter() - t0         # how long the host took to enqueue the timed steps (diagnostic)
    barrier()                    # torch.cuda.synchronize drains the whole device, the side streams of the pipelined mode included
    local_elapsed = time.perf_counter() - t0
    gc.enable()
    elapsed = max_over_ranks(local_elapsed, coll_dev)
    timed_env_steps = w.eng.read_metrics()['env_steps'] - steps0
    pipelined_timed = w.eng.pipelined_steps() - pipelined0

    # ---- metrics all-gather (the only collective of the path; off the step critical path) ----
    # With synchronised phases the accumulators are per running episode (env.py:329-338 zeroes them at reset)
    # and the default warmup + steps ends exactly on a boundary, so play to mid-day before reading them.
    if args.phase == 'sync':
        w.run((144 - (args.warmup + args.steps) % EPISODE) % EPISODE)
    local_vec = np.concatenate([metrics_vector(w.eng.read_metrics()),
                                [local_elapsed, float(local_rank), float(timed_env_steps)]])
    per_rank = all_gather_vector(local_vec, coll_dev)
    total = per_rank[:, :6].sum(axis=0)

    # ---- strong-scaling form beside a weak N > 1 run: --global-envs environments split over the ranks ----
    strong = None
    if world > 1 and args.scaling == 'weak':
        Ns = max(4, args.global_envs // world)
        ws = EvWorkload(args.site, Ns, local_rank, rank, project=project, episodes=args.episodes, bank=args.bank,
                        ring=args.ring, busy=args.busy, phase=args.phase, battery=args.battery, pipeline=args.pipeline)
        ws.run(args.warmup)
        barrier()
        t1 = time.perf_counter()
        ws.run(args.steps)
        barrier()
        el_s = max_over_ranks(time.perf_counter() - t1, coll_dev)
        ws.close()
        strong = {'scaling': 'strong', 'global_envs': Ns * world, 'envs_per_gpu': Ns, 'steps': args.steps,
                  'ms_per_step': round(el_s / args.steps * 1e3, 5), 'value': round(Ns * world * args.steps / el_s, 1),
                  'unit': 'env-steps/s'}

    roofline = cpu_baseline = episode_generation = secondary = None
    if rank == 0:
        abar = mean_entries_per_env(w)
        window_ms = elapsed / args.steps * 1e3
        timed = w.time_kernels(args.kernel_timing_steps)
        alg_b = algorithmic_bytes_per_env_step(n, k)
        # `frac` / `achieved`: on the window `ms_per_step` is timed in (cold start of the two launch trains and the final drain
        # included); `frac_steady`, `frac_hbm`, `frac_floor`: on the steady step period measured after it with HIP events
        roofline = roofline_record(w, timed, alg_b, window_ms, abar)
        roofline['launch_overhead_ms'] = round(window_ms - roofline.get('step_period_ms', roofline['avg_kernel_ms']), 5)
        roofline['host_issue_ms_per_step'] = round(host_issue_s / args.steps * 1e3, 5)      # of the timed steps, rank 0
        if w.pipeline == 2 and not args.no_single_launch:
            # the same workload as ONE launch per step (evc_set_pipeline(1)), for continuity with rounds 1-2
            w.eng.set_pipeline(1)
            w.run(32)
            single_ms = w.wall_ms_per_step(max(args.steps, 288))
            t1 = w.time_kernels(args.kernel_timing_steps)
            k_ms = float(t1['main_ms'].mean())
            floor_b = roofline['algorithmic_bytes_per_env_step']
            roofline['single_launch'] = {'ms_per_step': round(single_ms, 5), 'env_steps_per_s': round(N / single_ms * 1e3, 1),
                                         'avg_kernel_ms': round(k_ms, 5),
                                         'frac': round(floor_b * N / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                         'frac_survey': round(alg_b * N / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            w.eng.set_pipeline(2)
    if rank == 0:
        # Reset-path row: refill the whole episode bank with the on-device GMM generator
        # (after the timed region; the bank is not used again).
        from sustaingym_amd.event_generation import gmm_device_tables
        w.eng.upload_gmm(dict(gmm_device_tables(args.site, 'Summer 2019'), num_days=w.moer_days))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w.eng.generate_episodes(0, w.P, 1, 0)
        e0.record()
        for rep in range(10):
            w.eng.generate_episodes(0, w.P, 1, rep * w.P)
        e1.record()
        torch.cuda.synchronize()
        gen_ms = e0.elapsed_time(e1) / 10
        episode_generation = {'kernel': 'evc::generate_kernel', 'episodes': w.P, 'ms': round(gen_ms, 4),
                              'episodes_per_s': round(w.P / gen_ms * 1e3, 1)}
    tie = w.eng.read_metrics()
    cpu_inputs = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from sustaingym_amd.hostio import to_host
        cpu_inputs = [to_host(r[:min(args.cpu_envs, N)]) for r in w.ring]        # before the engine (and its ring) goes away
    w.close()
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = {}
        legs_deadline = time.monotonic() + args.secondary_budget_s
        for name, fn in (('sync_reference', lambda: secondary_sync_reference(args.site, local_rank, args.battery, project)),
                         ('action_ring_32', lambda: secondary_action_ring(args.site, local_rank, args.battery, project)),
                         ('gmm_caltech', lambda: secondary_days('caltech', 'gmm', local_rank, args.battery)),
                         ('gmm_jpl', lambda: secondary_days('jpl', 'gmm', local_rank, args.battery)),
                         ('real_caltech', lambda: secondary_days('caltech', 'real', local_rank, args.battery)),
                         ('rollout_greedy_65536', lambda: secondary_rollout('greedy', local_rank, args.battery)),
                         ('rollout_random_65536', lambda: secondary_rollout('random', local_rank, args.battery)),
                         ('rollout_random_65536_gmm', lambda: secondary_rollout('random', local_rank, args.battery, 'gmm')),
                         ('rollout_greedy_65536_gmm', lambda: secondary_rollout('greedy', local_rank, args.battery, 'gmm')),
                         ('rollout_greedy_65536_jpl_gmm', lambda: secondary_rollout('greedy', local_rank, args.battery, 'gmm', 'jpl')),
                         ('closed_loop_65536', lambda: secondary_closed_loop(local_rank, args.battery)),
                         ('vector_env_api', lambda: secondary_vector_env_api(local_rank, args.battery)),
                         ('small_configs', lambda: secondary_small_configs(local_rank, args.battery)),
                         ('rccl_world1', secondary_rccl_world1),
                         ('multiagent_8192x54', lambda: secondary_multiagent(local_rank, args.battery)),
                         ('battery_16384', lambda: secondary_battery(local_rank)),
                         ('battery_rollout_16384', lambda: secondary_battery_rollout(local_rank)),
                         ('tie_snap_reach', lambda: secondary_tie_snap(local_rank, args.battery))):
            secondary[name] = run_leg(fn, args.leg_budget_s, legs_deadline)

    # The CPU baseline runs LAST: its OpenMP threads saturate the container's CPU quota and the cgroup throttles the whole
    # process for a while afterwards — measured GPU legs that follow it become host-bound (secondary records 20 % off).
    if cpu_inputs is not None:
        cpu_baseline = cpu_baseline_record(args, w, cpu_inputs)
    if rank == 0:
        value = N * world * args.steps / elapsed
        tags = (' [congested variant]' if args.busy else '') + (' [GMM episodes]' if args.episodes == 'gmm' else '')
        line = {
            'metric': 'env-steps/sec at 65k batched 54-station EVChargingEnv; 1/2/4/8 MI355X',
            'value': round(value, 1), 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 5),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic' if args.episodes == 'synthetic' else
                    'synthetic actions / MOER, episodes sampled on the device from the packaged GMM',
            'config': {'workload': f'{N} batched {n}-station EVChargingEnv ({args.site}) per GPU, continuous actions, '
                                   f'project_action_in_env={project}, autoreset over a {w.P}-episode bank, '
                                   f'episode phases {"staggered uniformly over the day" if args.phase == "stagger" else "synchronised"}'
                                   + tags,
                       'envs_per_gpu': N, 'global_envs': N * world, 'parallelism': f'env-shard x{world}',
                       'actions': f'U[0,1) float32 resident in HBM, a ring of {args.ring} slices ({args.ring * N * n * 4 / 1e6:.0f} MB) that comes round every {args.ring} steps — '
                                  'with 8 slices it stays in the 256 MB Infinity Cache between visits (secondary.action_ring_32: every row from HBM)',
                       'battery_model': args.battery,
                       'phase': args.phase,
                       'pipeline': ('2 half-batch launches per step on 2 streams (evc_set_pipeline): all outputs of every step written, '
                                    'the halves\' launches overlap across steps' if w.pipeline == 2 else '1 launch per step'),
                       'launches_per_step': 2 if w.pipeline == 2 else 1, 'settle_steps': args.settle_steps, 'host_wait': host_wait,
                       'pipelined_steps_timed': int(pipelined_timed)},
            # proof that `world` ranks stepped: gathered over the process group
            'ranks_seen': int(per_rank.shape[0]),
            'per_rank': {'value': [round(N * args.steps / e, 1) for e in per_rank[:, 6]],
                         'device': [int(d) for d in per_rank[:, 7]],
                         'env_steps_timed': [int(s) for s in per_rank[:, 8]]},
            'env_steps_timed': int(per_rank[:, 8].sum()),
            # the same run in the other scaling form (N > 1 weak runs only): north_star's "65 536 batched ... on 8 GPUs"
            'strong_scaling': strong,
            # the reference's own episode distribution beside the headline's quiet synthetic days (secondary.gmm_caltech)
            'value_reference_distribution': (secondary or {}).get('gmm_caltech', {}).get('env_steps_per_s') if secondary else None,
            # ... its step time, its congested 4-hour blocks (one launch per step) and the JPL day, up here where a truncated tail still shows them
            'gmm_days': None if not secondary else {
                key: {'us_per_step': round(rec['ms_per_step'] * 1e3, 2), 'single_launch_us': round(rec['single_launch']['ms_per_step'] * 1e3, 2),
                      'kernel_us_by_4h': rec['kernel_us_by_4h'], 'solver_kernel_us_by_4h': rec['solver_kernel_us_by_4h']}
                for key, rec in ((k2, secondary.get(k2) or {}) for k2 in ('gmm_caltech', 'gmm_jpl')) if 'ms_per_step' in rec},
            'rollout_greedy_gmm': None if not secondary else {
                key: {'env_steps_per_s': rec['env_steps_per_s'], 'us_per_period': rec['us_per_period']}
                for key, rec in ((k2, secondary.get(k2) or {}) for k2 in ('rollout_greedy_65536_gmm', 'rollout_greedy_65536_jpl_gmm')) if 'us_per_period' in rec},
            'roofline': roofline, 'cpu_baseline': cpu_baseline, 'episode_generation': episode_generation,
            'episode_metrics': {'profit': float(total[0]), 'carbon_cost': float(total[1]),
                                'excess_charge': float(total[2]), 'episodes_finished': float(total[4]),
                                'envs_with_status': float(total[5]),
                                # reach of the tie snap on rank 0: values a projection solver moved, and how
                                # many of them lay within 1e-6 A of a rounding boundary before the snap
                                # the lean streaming kernels do not count what their in-row water-filling moves (the counting
                                # code cost 1 us per step): no figure here; secondary.tie_snap_reach counts a
                                # whole GMM day through the kernels that do
                                'solver_moved_values': None, 'tie_snap_near_boundary': None,
                                'slow_path_moved_values': float(tie['solver_moved_values'])},
            'secondary': secondary,
        }
        emit(line, args.full_out or None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched 54-station EVChargingEnv step() on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  A "step" is one pass of the hot
path (EVChargingEnv.step: projection -> pilots -> ACN-Sim charge/event pass -> observation ->
reward) over one batch of environments with the actions already resident in HBM.  Environments
are independent, so they shard over ranks with no data-path collective (weak scaling: fixed
envs per GPU); only the final metrics are all-gathered.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env-step, Caltech n=54, k=36
#   read action 4n = 216; read+write station state {remaining f64, dep i16, est i16} 12n*2 = 1296;
#   event cursor + next event ~32; MOER row (k+1)*4 = 148; write obs (2n+k+2)*4 = 584;
#   write reward/done/breakdown 8+1+24 = 33   => 2309 B
def algorithmic_bytes_per_env_step(n: int, k: int) -> int:
    return 4 * n + 12 * n * 2 + 32 + (k + 1) * 4 + (2 * n + k + 2) * 4 + 33


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec peak


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=576)
    p.add_argument('--warmup', type=int, default=288)
    p.add_argument('--envs-per-gpu', type=int, default=65536)
    p.add_argument('--site', default='caltech', choices=['caltech', 'jpl'])
    p.add_argument('--no-project', action='store_true', help='project_action_in_env=False')
    p.add_argument('--bank', type=int, default=8192, help='distinct synthetic episodes resident in HBM')
    p.add_argument('--ring', type=int, default=8, help='distinct action batches resident in HBM')
    p.add_argument('--busy', action='store_true',
                   help='congested variant of the workload (30-60 long sessions per day); not the headline')
    p.add_argument('--episodes', default='synthetic', choices=['synthetic', 'gmm'],
                   help="'gmm': the bank is generated on the device from the reference's GMM (Summer 2019) - busier\n"
                        'days than the synthetic default; not the headline')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help="process-group backend for --gpus > 1 ('nccl' = RCCL; 'gloo' only to exercise the N > 1\n"
                        'logic with several ranks on ONE GPU, see --single-device)')
    p.add_argument('--single-device', action='store_true',
                   help='testing aid: every rank uses cuda:0 (needs --backend gloo)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-envs', type=int, default=8192)
    p.add_argument('--cpu-steps', type=int, default=96, help='minimum timed steps of the cpu_baseline sample')
    p.add_argument('--kernel-timing-steps', type=int, default=288,
                   help='steps timed kernel by kernel for the roofline leg (288 = one whole day: the launch\n'
                        'duration follows the time of day)')
    return p.parse_args()


def main():
    args = parse_args()
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.single_device:
            assert args.backend == 'gloo', '--single-device needs --backend gloo (RCCL wants one GPU per rank)'
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    coll_dev = dev if (world == 1 or args.backend == 'nccl') else torch.device('cpu')   # where collectives run

    from sustaingym_amd.distributed import all_gather_metrics, max_over_ranks, metrics_vector
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.network import site_str_to_site
    from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer

    net = site_str_to_site(args.site)
    n, k = net.num_stations, 36
    N = args.envs_per_gpu
    project = not args.no_project
    P = min(args.bank, max(N, 1))
    moer_days = 32
    busy_kw = dict(min_sessions=30, max_sessions=60, max_arrival=120, min_duration=40, max_duration=160) if args.busy else {}
    ns, sess, req, day = synthetic_episodes(P, n, seed=1000 + rank, stride=64, moer_days=moer_days, **busy_kw)
    moer = synthetic_moer(moer_days, seed=7)
    eng = StepEngine(net, N, moer_forecast_steps=k, project_action=project, autoreset=True,
                     device=local_rank, bank_slots=P, max_sessions=128 if args.episodes == 'gmm' else 64,
                     moer_days=moer_days)
    eng.upload_moer(moer)
    if args.episodes == 'gmm':
        from sustaingym_amd.event_generation import gmm_device_tables
        eng.upload_gmm(dict(gmm_device_tables(args.site, 'Summer 2019'), num_days=moer_days))
        eng.generate_episodes(0, P, 1000 + rank, 0)
        ns, sess, req, day, _ = eng.download_episodes(0, P)          # for the cpu_baseline leg
    else:
        eng.upload_episodes(ns, sess, req, day)
    eng.set_autoreset_stride(1)
    eng.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    ring = [torch.rand((N, n), dtype=torch.float32, device=dev, generator=gen) for _ in range(args.ring)]
    ptrs = [t.data_ptr() for t in ring]
    step, out = eng.make_stepper()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(ptrs[i % len(ptrs)])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ptrs[i % len(ptrs)])
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, coll_dev)

    # ---- metrics all-gather (the only collective of the path; off the step critical path) ----
    # The accumulators are per running episode (env.py:329-338 zeroes them at reset) and the default
    # warmup + steps ends exactly on an episode boundary, so play to mid-day before reading them.
    tail = (144 - (args.warmup + args.steps) % 288) % 288
    for i in range(tail):
        step(ptrs[i % len(ptrs)])
    _, total = all_gather_metrics(metrics_vector(eng.read_metrics()), coll_dev)

    # ---- per-kernel duration with HIP events on the engine's stream (rank 0): start / stop events
    # attached to each launch (evc_enable_timing), averaged over a whole day of steps ----
    roofline = None
    if rank == 0:
        eng.enable_timing(True)
        main_ms, slow_ms, slow_cnt = [], [], []
        for i in range(args.kernel_timing_steps):
            step(ptrs[i % len(ptrs)])
            a, b = eng.last_step_ms()
            main_ms.append(a)
            slow_ms.append(b)
            if project:
                slow_cnt.append(eng.last_slow_count())
        eng.enable_timing(False)
        layout = os.environ.get('EVC_LAYOUT', 'compact')          # engine default
        avg_main = float(np.mean(main_ms))
        avg_slow = float(np.mean(slow_ms))
        bytes_per_launch = algorithmic_bytes_per_env_step(n, k) * N
        achieved = bytes_per_launch / (avg_main * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f'{args.site}_N{N}_project{int(project)}' + ('' if layout == 'dense' else f'_{layout}')
                traffic = tj.get(key, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': 'evc::step_kernel_cquad' if layout == 'compact' else 'evc::step_kernel_quad',
                    'achieved': round(achieved, 2),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5),
                    'traffic': traffic, 'state_layout': layout, 'avg_kernel_ms': round(avg_main, 5),
                    'kernel_ms_min_max': [round(float(np.min(main_ms)), 5), round(float(np.max(main_ms)), 5)],
                    'kernel_launches_timed': len(main_ms),
                    'traffic_gbs': (round(traffic / (avg_main * 1e-3) / 1e9, 2) if traffic else None),
                    'solver_kernel_ms': round(avg_slow, 5),
                    'slow_queue_envs_per_step': (round(float(np.mean(slow_cnt)), 1) if slow_cnt else 0.0),
                    'algorithmic_bytes_per_launch': bytes_per_launch}

    # ---- CPU baseline: the oracle (scalar C restatement) on the host cores, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        cn, cs = min(args.cpu_envs, N), args.cpu_steps
        bat = ob.OracleBatch(ob.OracleNetwork(net), cn, k, project)
        bat.set_bank(ns, sess, req, day, moer, autoreset_stride=1)
        bat.reset(np.arange(cn, dtype=np.int32) % P)
        # one thread per CPU this process may actually use (more threads than the cgroup quota only get
        # throttled): oracle.binding.default_threads
        try:
            quota = open('/sys/fs/cgroup/cpu.max').read().split()
            cgroup_cpus = None if quota[0] == 'max' else round(int(quota[0]) / int(quota[1]), 2)
        except Exception:
            cgroup_cpus = None
        host = {'cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0)), 'cgroup_cpus': cgroup_cpus,
                'omp_max_threads': ob.max_threads()}
        cores = ob.default_threads()
        acts = [r[:cn].cpu().numpy() for r in ring]
        # skip the empty early-morning periods so that the sample has plugged-in EVs; their rate sizes the
        # timed sample to ~3 s of wall time on whatever host this is (bounded: --cpu-steps .. 2304 steps = 8 days)
        t1 = time.perf_counter()
        for i in range(96):
            bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=cores)
        rate0 = cn * 96 / (time.perf_counter() - t1)
        cs = int(min(2304, max(cs, 3.0 * rate0 / cn)))
        t1 = time.perf_counter()
        for i in range(cs):
            bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=cores)
        dt = time.perf_counter() - t1
        # the same sample continued on one thread (SURVEY §8d asks for both), ~2 s
        t1 = time.perf_counter()
        for i in range(4):
            bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=1)
        c1 = int(min(96, max(4, 2.0 / ((time.perf_counter() - t1) / 4))))
        t1 = time.perf_counter()
        for i in range(c1):
            bat.step(acts[i % len(acts)], autoreset=True, debug=False, threads=1)
        dt1 = time.perf_counter() - t1
        first = 97
        cpu_baseline = {'value': round(cn * cs / dt, 1), 'unit': 'env-steps/s', 'cores': cores,
                        'kind': 'port',
                        'sample': f'{cn} envs x {cs} steps (from period {first} on, across autoresets) of the same '
                                  f'workload, {dt:.1f} s, oracle/ C restatement, OpenMP over envs with {cores} threads',
                        'single_thread_value': round(cn * c1 / dt1, 1),
                        'single_thread_sample': f'{cn} envs x {c1} steps, {dt1:.1f} s, 1 thread', 'host': host}

    # Reset-path row, reported beside the headline: refill the whole episode bank with
    # the on-device GMM generator (after the timed region; the bank is not used again).
    episode_generation = None
    if rank == 0:
        from sustaingym_amd.event_generation import gmm_device_tables
        eng.upload_gmm(dict(gmm_device_tables(args.site, 'Summer 2019'), num_days=moer_days))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.generate_episodes(0, P, 1, 0)
        e0.record()
        for rep in range(10):
            eng.generate_episodes(0, P, 1, rep * P)
        e1.record()
        torch.cuda.synchronize()
        gen_ms = e0.elapsed_time(e1) / 10
        episode_generation = {'kernel': 'evc::generate_kernel', 'episodes': P, 'ms': round(gen_ms, 4),
                              'episodes_per_s': round(P / gen_ms * 1e3, 1)}

    if rank == 0:
        value = N * world * args.steps / elapsed
        line = {
            'metric': 'env-steps/sec at 65k batched 54-station EVChargingEnv; 1/2/4/8 MI355X',
            'value': round(value, 1), 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 5),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic' if args.episodes == 'synthetic' else 'synthetic actions / MOER, episodes sampled on the device from the packaged GMM',
            'config': {'workload': f'{N} batched {n}-station EVChargingEnv ({args.site}) per GPU, continuous '
                                   f'actions, project_action_in_env={project}, autoreset over a {P}-episode bank' + (' [congested variant]' if args.busy else '') + (' [GMM episodes]' if args.episodes == 'gmm' else ''),
                       'envs_per_gpu': N, 'global_envs': N * world, 'parallelism': f'env-shard x{world}',
                       'actions': 'U[0,1) float32 resident in HBM'},
            'roofline': roofline, 'cpu_baseline': cpu_baseline, 'episode_generation': episode_generation,
            'episode_metrics': {'at_period': 144, 'profit': float(total[0]), 'carbon_cost': float(total[1]),
                                'excess_charge': float(total[2]), 'episodes_finished': float(total[4]),
                                'envs_with_status': float(total[5])},
        }
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

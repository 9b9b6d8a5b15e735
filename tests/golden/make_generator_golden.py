"""Golden vectors of the counter-based episode generator: episodes are a pure function of
(model tables, seed, episode number), so a small committed sample pins the random stream, the arithmetic
and the packaged GMM tables against silent changes — in the C specification (oracle/evc_oracle_gen.c,
CPU test) and in the HIP kernel (GPU test).  Produced by the specification itself:

    python tests/golden/make_generator_golden.py      # rewrites tests/golden/generator_episodes.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [('caltech', 'Summer 2019', 0x5EED, 1 << 40, 48), ('jpl', 'Summer 2021', 12345, 7, 48)]


def main():
    from oracle.binding import OracleGenerator
    from sustaingym_amd.event_generation import gmm_device_tables
    out = {}
    for i, (site, period, seed, first, count) in enumerate(CASES):
        tabs = gmm_device_tables(site, period)
        ns, sess, req, day, mp = OracleGenerator(tabs, len(tabs['station_usage'])).episodes(seed, first, count, 128)
        out[f'ns_{i}'], out[f'sess_{i}'], out[f'req_{i}'], out[f'day_{i}'], out[f'mp_{i}'] = \
            ns, sess.view(np.int16).reshape(count, 128, 4), req, day, mp
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generator_episodes.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()

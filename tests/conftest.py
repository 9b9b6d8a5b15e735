import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the built-in JPL constraint set announces itself once per run, not once per engine
    config.addinivalue_line('filterwarnings', 'once::sustaingym_amd.network.ProvisionalNetworkWarning')


@pytest.fixture(scope='session')
def caltech():
    from sustaingym_amd.network import caltech_acn
    return caltech_acn()


@pytest.fixture(scope='session')
def jpl():
    from sustaingym_amd.network import jpl_acn
    return jpl_acn()

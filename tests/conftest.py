import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd', 'lib')
for p in (REPO, LIB):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle's threads: the box's CPU share (a one-GPU box gets 16 cores but reports 128 - 128 OpenMP threads
    # on 16 cores spend their time spinning: the B=40 oracle step took 22 s there)
    try:
        import torch
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    except Exception:
        pass
    # the C-ABI library is built in-tree (hipcc cross-compiles without a GPU); build it when a fresh checkout
    # runs the tests before `python __graft_entry__.py`
    lib = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd', 'csrc', 'libhrnet_hip.so')
    if not os.path.exists(lib) and os.path.exists('/opt/rocm/bin/hipcc'):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _conv_ring_switch_restored(request):
    """the library reads HRNET_CONV_RING once per process; tests that compare the LDS-ring convolutions with the
    tile-walking body flip hrnet_conv_ring_enable() instead - put it back to the default (on) after every GPU test"""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from hipnet import _capi as C
        C.call('hrnet_conv_ring_enable', 1)

import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore', message='Using padding=')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


DP2_TEST = 'test_two_rank_data_parallel_through_the_engine'


def pytest_collection_finish(session):
    """The two-rank data-parallel run (scripts/dp_two_ranks.py) needs fresh rank processes started by a parent that has not
    touched the GPU.  When its test is selected, the ranks therefore run HERE, right after collection and before any test has
    initialised the device in this process; the test itself only looks at the recorded result."""
    import subprocess
    if not any(item.name == DP2_TEST for item in session.items):
        return
    log = os.path.join(ROOT, 'gpurun_out', 'dp_two_ranks_pytest.log')
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'dp_two_ranks.py'), '--log', log, '--timeout', '420'],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=480)
        q = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'dp_two_ranks.py'), '--train', '--log', log, '--timeout', '300'],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=360)
        rc, out = p.returncode or q.returncode, p.stdout + q.stdout
        import torch
        if torch.cuda.device_count() >= 2:          # (counting devices does not initialise the GPU) two devices: the RCCL transport too
            r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'dp_two_ranks.py'), '--rccl', '--timeout', '300'],
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=360)
            rc, out = rc or r2.returncode, out + r2.stdout.replace('PASS', 'PASS(rccl)').replace('RESULT: OK', 'RESULT(rccl): OK').replace(
                'bucket order [0, 1, 2, 3]', 'bucket order(rccl) [0, 1, 2, 3]')
        session.config._dp_two_ranks = (rc, out)
    except Exception as e:       # noqa: BLE001 -- the test reports it
        session.config._dp_two_ranks = (-1, 'could not run scripts/dp_two_ranks.py: %r' % (e,))

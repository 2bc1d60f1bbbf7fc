import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """numpy's BLAS / OpenMP pools sized to the CPUs this container may use (oracle/oracle_c.py::usable_cpus), not to the CPUs the
    host shows: oversubscribing a 16-CPU quota with 256 threads slowed the oracle-bound tests several times over."""
    if "OMP_NUM_THREADS" in os.environ:
        return
    try:
        from threadpoolctl import threadpool_limits
        from oracle import oracle_c
        session.config._sx_pool_limit = threadpool_limits(limits=oracle_c.usable_cpus())
    except Exception:           # threadpoolctl is a convenience here, not a requirement
        pass

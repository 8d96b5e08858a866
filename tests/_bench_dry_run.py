"""CPU dry run of bench.py's control flow (TEST INFRASTRUCTURE, started by tests/test_bench_dry_run_cpu.py): the SAME bench.main()
and bench.measure() -- rank set-up through launcher.spawn_ranks or a launcher's environment, the per-step all-gather into a
persistent array, the rank check (`rccl_ranks`), rank 0's JSON line -- with the device side replaced by a stand-in: gloo instead
of RCCL, wall-clock "events", and the test-only OracleEngine (Python integers) instead of the HIP engine.  What it proves is the
host logic of the N > 1 path; the numbers in its line mean nothing."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Event:
    def __init__(self):
        self.t = None

    def record(self, *_):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class CpuRuntime:
    backend = "gloo"

    def __init__(self):
        from _oracle_engine import OracleEngine

        class DryEngine(OracleEngine):
            """OracleEngine plus the measurement hooks bench.py reads (constants: there is nothing to measure here)."""

            def set_latency_mode(self, mode): pass
            def set_onelane_mode(self, mode): pass
            def set_fork_mode(self, mode): pass
            def set_chip_share(self, n): pass
            def mac_counter(self, reset=False): return 1.0
            def peak_probe(self): return 1.0e12
            def policy(self): return {"dry_run": True}
            def clock_probe(self, key, rho): return 1.0, 1.0
            def close(self): pass

        self._cls = DryEngine
        self._default = None

    def check_device(self, rank, local_rank): pass
    def dist_device(self, local_rank): return None
    def synchronize(self): pass
    def event(self): return _Event()
    def stream(self): return None
    def current_device(self): return int(os.environ.get("LOCAL_RANK", "0"))
    def empty_cache(self): pass
    def cu_count(self, eng): return 256

    def free_memory(self, eng):
        v = os.environ.get("SC_DRY_FREE_BYTES")       # the test of the free-memory check names the GPU's free bytes
        return None if v is None else int(v)

    def default_engine(self):
        if self._default is None:
            self._default = self._cls()
        return self._default

    def new_engine(self):
        return self._cls()


if __name__ == "__main__":
    import bench

    bench.main(sys.argv[1:], runtime=CpuRuntime(), script=os.path.abspath(__file__))

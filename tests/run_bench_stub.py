"""Test driver: bench.py's launcher, rendezvous, barrier / max-over-ranks timing, extra legs and JSON contract on CPU ranks.

    python tests/run_bench_stub.py --gpus 2 --steps 4 ...     (the same command line as bench.py)

It imports bench, hands it tests/bench_stub.py's StubContext -- a context that computes nothing -- and calls bench.main();
with --gpus N it relaunches ITSELF under torch.distributed.run, so every rank goes through the same injection.  bench.py
itself has no switch for this."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import bench  # noqa: E402
from bench_stub import StubContext  # noqa: E402

bench.TEST_CONTEXT_FACTORY = StubContext
sys.exit(bench.main())

"""Script form of tests/test_bench_launch.py's driver (torch.distributed.run needs a path): bench.py's launcher with the
detector stubbed, gloo on CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from test_bench_launch import stub_target  # noqa: E402

if __name__ == "__main__":
    bench.launch(bench.parse_args(sys.argv[1:]), target=stub_target)

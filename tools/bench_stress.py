#!/usr/bin/env python
"""configs[4] stress of the label-graph GCN step on ONE GPU, cache-cold (mgnns_amd/stress.py::measure):

    python tools/bench_stress.py [N] [batch]      -> one JSON line
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import stress  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else stress.N_NODES
    b = int(sys.argv[2]) if len(sys.argv) > 2 else stress.BATCH
    print(json.dumps(stress.measure("cuda:0", n, b, quick=False)))

#!/usr/bin/env python3
"""Development aid: throughput of the View-file loader (file -> pinned staging -> HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersonic_amd as ss  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
schema = ss.TupleSchema([ss.Attribute("a", ss.INT64), ss.Attribute("b", ss.INT64, ss.NULLABLE), ss.Attribute("d", ss.DOUBLE)])
rng = np.random.default_rng(1)
view = ss.View(schema, [rng.integers(0, 1000, rows), ss.Column(rng.integers(0, 1000, rows), rng.random(rows) < 0.1), rng.standard_normal(rows)])
path = "/tmp/ssgpu_file_bench.ssv"
out = ss.FileOutput(path); out.Write(view); out.Finalize()
size = os.path.getsize(path)
ctx = ss.Context(0)
for rep in range(3):
    t0 = time.perf_counter()
    dev = ss.FileInput(schema, path, ctx)
    dt = time.perf_counter() - t0
    print("load %d rows, %.2f GB in %.3f s -> %.2f GB/s (page-cache resident file)" % (dev.row_count(), size / 1e9, dt, size / dt / 1e9))
    del dev
os.remove(path)

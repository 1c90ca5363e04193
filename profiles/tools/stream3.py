import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: F401
import fortran_davidson_amd as fd
with fd.CEngine(n=1024, max_cols=16) as e:
    for d in (0, 1 << 26, 1 << 29):
        for _ in range(2):
            print(d, [round(x, 1) for x in e.bench_stream3(d, 5)], flush=True)

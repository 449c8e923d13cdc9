#!/usr/bin/env python3
"""Writes tests/golden/traj64.npz: the six-iteration Demo_RSSS trajectory of steps2.npz (same seeds, LR schedule in the loop)
run on the CPU ORACLE in double precision -- the truth tests/test_gpu_interchange.py::test_rsss_trajectory_... measures both the
reference's fp32 fixture and the HIP path against.  Pure oracle arithmetic (no reference import); kept as a fixture because the
fp64 run costs ~4 minutes of CPU per test session.  The test recomputes it when the file is missing or its meta differs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for q in (ROOT, HERE, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, q)


def main():
    import test_gpu_interchange as T
    z = np.load(os.path.join(HERE, 'steps2.npz'))
    T._TRAJ64.clear()
    out = T._oracle_trajectory_fp64(z, use_fixture=False)
    np.savez_compressed(os.path.join(HERE, 'traj64.npz'), meta=z['traj/meta'], **{'it%d' % i: t.numpy() for i, t in enumerate(out)})
    print('wrote traj64.npz', [tuple(t.shape) for t in out])


if __name__ == '__main__':
    main()

"""tools/training_equivalence.py (the tool behind profiles/r06_training_equivalence.md): its data, batch schedule and scoring are
deterministic and shared by every leg, and the report reads what the legs write.  (The legs themselves -- 200 adversarial iterations
each -- are run by hand on the GPU box, not by the suites.)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_scenes_schedule_and_scores_are_deterministic():
    import training_equivalence as te
    a, b = te.scenes(), te.scenes()
    for s, t in zip(a, b):
        assert torch.equal(s, t)
    x, y, region, truth = a
    assert x.shape == (te.SCENES, te.C, te.HW, te.HW) and set(truth.unique().tolist()) <= {0.0, 1.0}
    assert float((truth * (1 - region)).sum()) == 0.0 and 0.02 < float(truth.mean()) < float(region.mean())     # truth inside the weak label
    o1, o2 = te.batch_order(10), te.batch_order(10)
    assert all(torch.equal(p, q) for p, q in zip(o1, o2)) and all(len(p) == te.BATCH for p in o1)
    assert sorted(torch.cat(o1[:te.SCENES // te.BATCH]).tolist()) == list(range(te.SCENES))              # an epoch covers every scene once
    sd = {'w': torch.ones(4, 3, 3, 3), 'bn.running_mean': torch.ones(4), 'n': torch.zeros((), dtype=torch.int64)}
    p = te.perturb(sd, 1)
    assert 0 < float((p['w'] - 1).abs().max()) <= 1.2e-6 and torch.equal(p['bn.running_mean'], sd['bn.running_mean'])
    s = te.scores(torch.tensor([[1, 1, 0, 0]]), torch.tensor([[1, 0, 1, 0]]))
    assert (s['tp'], s['fp'], s['fn'], s['tn']) == (1, 1, 1, 1) and abs(s['f1'] - 0.5) < 1e-12 and abs(s['miou'] - 1 / 3) < 1e-12


def test_report_reads_the_legs(tmp_path):
    import argparse
    import training_equivalence as te
    rng = np.random.default_rng(0)
    x, y, region, truth = te.scenes()
    base = rng.uniform(0.1, 0.9, (te.SCENES, 1, te.HW, te.HW)).astype(np.float32)
    for leg, eps in (('oracle', 0.0), ('oracle_pert', 1e-3), ('hip_winograd', 5e-4)):
        curves = np.abs(rng.standard_normal((20, len(te.KEYS)))) + 1.0 if leg == 'oracle' else None
        if curves is None:
            curves = np.load(str(tmp_path / 'te_oracle.npz'))['curves'] * (1 + eps)
        maps = {'map%04d' % it: base[:te.BATCH] + eps * (it + 1) for it in (0, 10, 19)}
        np.savez_compressed(str(tmp_path / ('te_%s.npz' % leg)), curves=curves, final_cmap=base + eps, truth=truth.numpy(), region=region.numpy(),
                            meta=json.dumps(dict(leg=leg, iters=20, seconds=1.0, torch='x', threads=1)), **maps)
    md = str(tmp_path / 'r.md')
    te.report(argparse.Namespace(report=str(tmp_path), md=md))
    txt = open(md).read()
    assert '| hip_winograd | s_loss |' in txt and '`hip_winograd`: median 0.50, max 0.50' in txt and '| oracle_pert |' in txt

"""GPU: batches from the worker-process loader (recipes from a process that never touched the GPU, pixels rendered here from the
HBM-side source cache) are bit-identical to what the in-process pipeline renders under the same random streams."""
import pickle
import random

import numpy as np
import pytest
import torch

from tests.test_loader_workers_cpu import make_dataset

pytestmark = pytest.mark.gpu


def test_worker_loader_batches_equal_the_in_process_pipeline(tmp_path):
    from sy11.data.dataset import WorkerLoader, YOLODataset
    ds = make_dataset(tmp_path, n=16, size=96, device="cuda")
    twin = pickle.loads(pickle.dumps(ds))                               # in-process reference: real pixels, the worker's seed
    dl = WorkerLoader(ds, 4, procs=1, shuffle=False, seed=5, dtype=torch.float32)
    try:
        got = [next(dl._it) for _ in range(5)]                          # 4 batches of one epoch + the first of the next
    finally:
        dl.close()
    seed = 1000003 * (5 + 1)
    random.seed(seed); np.random.seed(seed % 2**32); torch.manual_seed(seed)
    order = [list(range(k, k + 4)) for k in (0, 4, 8, 12)] + [list(range(0, 4))]
    for batch, idx in zip(got, order):
        want = YOLODataset.collate_fn([twin[i] for i in idx], dtype=torch.float32)
        assert torch.equal(batch["img"], want["img"]) and batch["img"].dtype == torch.float32 and batch["img"].is_cuda
        for k in ("bboxes", "cls", "batch_idx"):
            assert torch.equal(batch[k], want[k]), k
    assert float(got[0]["img"].std()) > 0.05

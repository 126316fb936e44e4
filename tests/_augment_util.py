"""Shared by the augmentation tests: the product-side twin of the generator's in-memory dataset."""
import random
from types import SimpleNamespace

import numpy as np
import torch

from oracle.gen_golden_augment import BASE, CONFIGS, IMGSZ, N_SAMPLES  # noqa: F401  (re-exported)


class FakeDataset:
    data, use_keypoints = {}, False

    def __init__(self, gold, device):
        from sy11.data.augment import DeviceImage
        from sy11.utils.instance import Instances
        self._di, self._inst = DeviceImage, Instances
        self.n = int(gold["n_images"])
        self.imgs = [torch.from_numpy(gold[f"in.{i}.img"]).to(device) for i in range(self.n)]
        self.boxes = [gold[f"in.{i}.boxes"] for i in range(self.n)]
        self.cls = [gold[f"in.{i}.cls"] for i in range(self.n)]
        self.buffer = list(range(self.n))

    def __len__(self):
        return self.n

    def get_image_and_label(self, i):
        h, w = self.imgs[i].shape[:2]
        return {"im_file": f"im{i}", "ori_shape": (h, w), "resized_shape": (h, w), "img": self._di.wrap(self.imgs[i]),
                "cls": self.cls[i].copy(), "ratio_pad": (1.0, 1.0),
                "instances": self._inst(self.boxes[i].copy(), bbox_format="xywh", normalized=True)}


def run_pipeline(gold, name, device):
    """Yield (k, labels) for the N_SAMPLES samples of config `name`, RNG seeded like the generator; finally (None, rng_after)."""
    from sy11.data.augment import Format, v8_transforms
    hyp = SimpleNamespace(**{**BASE, **CONFIGS[name]})
    ds = FakeDataset(gold, device)
    tf = v8_transforms(ds, IMGSZ, hyp)
    tf.append(Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=hyp.bgr, defer=True))
    random.seed(1234)
    np.random.seed(1234)
    for k in range(N_SAMPLES):
        yield k, tf(ds.get_image_and_label(k % len(ds)))
    yield None, np.asarray([random.random(), np.random.uniform()])

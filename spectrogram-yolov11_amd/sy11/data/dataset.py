"""Detection dataset + loader of the data path with the reference's names (ultralytics/data/base.py BaseDataset :22-346,
data/dataset.py YOLODataset :46-248, data/build.py :30-157, data/utils.py img2label_paths :44-47 and the label rules
of verify_image_label :100-165), cut for the MI355X:

  * the host only DECODES (np.load for .npy spectrograms / caches, PIL for image files — the reference's cv2.imread is
    not available here) and keeps the label arithmetic;
  * every pixel operation of load_image / LetterBox / Mosaic / RandomPerspective / RandomHSV / RandomFlip / Format
    happens on the GPU: the raw image is uploaded once, ``load_image``'s long-side resize is one launch, and the whole
    augmentation chain of a sample is ONE launch that writes into that sample's slot of the batch tensor
    (``collate_fn`` allocates the batch, or fills a caller-provided static input, and renders the deferred images).

Two loaders share this dataset: ``InfiniteDataLoader`` runs everything in the training process (file decoding for the next batch
overlaps through a small thread pool); ``WorkerLoader`` moves the per-sample Python — random draws, label geometry, recipe
building — into worker PROCESSES that never touch the GPU and ship ~1.5 KB recipes (data/recipe.py), the reference's
DataLoader(num_workers=...) re-cut for pixels that live in HBM."""
from __future__ import annotations

import glob
import math
import os
import queue
import random
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from .. import ops as K
from ..utils.instance import Instances
from .augment import Compose, DeviceImage, Format, LetterBox, v8_transforms
from .recipe import Materializer, file_image, is_lazy, letterbox_image

IMG_FORMATS = {"bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm", "heic", "npy"}   # data/utils.py:38 + raw .npy
# cfg/default.yaml augmentation block
DEFAULT_HYP = dict(hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0,
                   flipud=0.0, fliplr=0.5, bgr=0.0, mosaic=1.0, mixup=0.0, copy_paste=0.0, mask_ratio=4, overlap_mask=True)


def img2label_paths(img_paths):
    """data/utils.py:44-47: .../images/x.ext -> .../labels/x.txt (last occurrence of /images/)."""
    sa, sb = f"{os.sep}images{os.sep}", f"{os.sep}labels{os.sep}"
    return [sb.join(x.rsplit(sa, 1)).rsplit(".", 1)[0] + ".txt" for x in img_paths]


_NPY_META = {}          # path -> (shape, data offset) of plain uint8 C-order .npy files: later reads are one np.fromfile call


def _load_npy(path):
    """np.load for the common case (uint8, C order, no pickle) without re-parsing the header on every epoch: the header costs
    more interpreter time than the 1.2 MB read itself, and the read (np.fromfile) releases the GIL."""
    key = str(path)
    meta = _NPY_META.get(key)
    if meta is None:
        with open(path, "rb") as f:
            version = np.lib.format.read_magic(f)
            shape, fortran, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
            meta = (shape, f.tell()) if (dtype == np.uint8 and not fortran) else False
        _NPY_META[key] = meta
    if meta is False:
        return np.load(path)
    shape, offset = meta
    return np.fromfile(path, dtype=np.uint8, count=int(np.prod(shape)), offset=offset).reshape(shape)


def read_image(path):
    """Decode to (h, w, 3) uint8 BGR like cv2.imread (base.py:153-165): a sibling / direct .npy wins, else PIL."""
    p = Path(path)
    npy = p if p.suffix == ".npy" else p.with_suffix(".npy")
    if npy.exists():
        im = _load_npy(npy)
    else:
        from PIL import Image
        with Image.open(p) as f:
            im = np.asarray(f.convert("RGB"))[..., ::-1]
    if im.ndim == 2:
        im = np.repeat(im[..., None], 3, 2)
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
        raise ValueError(f"{path}: expected an (h, w, 3) uint8 image, got {im.dtype} {im.shape}")
    return np.ascontiguousarray(im)


def image_hw(path):
    p = Path(path)
    npy = p if p.suffix == ".npy" else p.with_suffix(".npy")
    if npy.exists():
        return tuple(np.load(npy, mmap_mode="r").shape[:2])
    from PIL import Image
    with Image.open(p) as f:
        return (f.size[1], f.size[0])


def read_label(path, nc=None):
    """The checks of verify_image_label (data/utils.py:118-152) for detection: 5 columns, no negatives, normalised
    coordinates, class < nc, duplicate rows dropped (np.unique order).  Returns (n, 5) float32 [cls, x, y, w, h]."""
    if not os.path.isfile(path):
        return np.zeros((0, 5), np.float32)
    with open(path) as f:
        rows = [x.split() for x in f.read().strip().splitlines() if len(x)]
    if any(len(x) > 6 for x in rows):
        raise NotImplementedError(f"{path}: segment labels are out of scope for the detection data path")
    lb = np.array(rows, dtype=np.float32)
    if not len(lb):
        return np.zeros((0, 5), np.float32)
    assert lb.shape[1] == 5, f"labels require 5 columns, {lb.shape[1]} columns detected"
    assert lb[:, 1:].max() <= 1, f"non-normalized or out of bounds coordinates {lb[:, 1:][lb[:, 1:] > 1]}"
    assert lb.min() >= 0, f"negative label values {lb[lb < 0]}"
    if nc is not None:
        assert lb[:, 0].max() < nc, f"Label class {int(lb[:, 0].max())} exceeds dataset class count {nc}."
    _, i = np.unique(lb, axis=0, return_index=True)
    if len(i) < len(lb):
        lb = lb[i]
    return lb


class YOLODataset:
    """base.py:22-346 + dataset.py:46-248 for task="detect"."""

    def __init__(self, img_path, imgsz=640, cache=False, augment=True, hyp=None, prefix="", rect=False, batch_size=16,
                 stride=32, pad=0.5, single_cls=False, classes=None, fraction=1.0, data=None, task="detect", device="cuda"):
        if task != "detect":
            raise NotImplementedError("sy11 YOLODataset carries the detection task only")
        self.img_path, self.imgsz, self.augment, self.single_cls = img_path, imgsz, augment, single_cls
        self.prefix, self.fraction, self.data, self.device = prefix, fraction, data or {}, torch.device(device)
        self.rect, self.batch_size, self.stride, self.pad = rect, batch_size, stride, pad
        self.hyp = hyp if hyp is not None else SimpleNamespace(**DEFAULT_HYP)
        self.im_files = self.get_img_files(img_path)
        self.labels = self.get_labels()
        self.update_labels(include_class=classes)
        self.ni = len(self.labels)
        if self.rect:
            assert self.batch_size is not None
            self.set_rectangle()
        self.buffer = []                                                    # base.py:77-79
        self.max_buffer_length = min((self.ni, self.batch_size * 8, 1000)) if self.augment else 0
        self.ims, self.im_hw0, self.im_hw = [None] * self.ni, [None] * self.ni, [None] * self.ni
        self._decoding = {}                                                 # index -> Future of a background read_image
        self.recipe_mode = False                                            # True inside a loader worker process: images are LazyImage nodes
        self.transforms = self.build_transforms(hyp=self.hyp)

    def __getstate__(self):
        """What a worker process receives: files, labels, hyper-parameters, transforms — no pixels, no futures."""
        st = dict(self.__dict__)
        st["ims"], st["im_hw0"], st["im_hw"] = [None] * self.ni, [None] * self.ni, [None] * self.ni
        st["_decoding"], st["buffer"] = {}, []
        st.pop("_close_listeners", None)
        return st

    # -- files and labels
    def get_img_files(self, img_path):
        f = []
        for p in img_path if isinstance(img_path, list) else [img_path]:
            p = Path(p)
            if p.is_dir():
                f += glob.glob(str(p / "**" / "*.*"), recursive=True)
            elif p.is_file():
                with open(p) as t:
                    parent = str(p.parent) + os.sep
                    f += [x.replace("./", parent) if x.startswith("./") else x for x in t.read().strip().splitlines()]
            else:
                raise FileNotFoundError(f"{self.prefix}{p} does not exist")
        files = sorted(x.replace("/", os.sep) for x in f if x.split(".")[-1].lower() in IMG_FORMATS)
        # an image and its .npy cache are one sample
        seen, im_files = set(), []
        for x in files:
            stem = x.rsplit(".", 1)[0]
            if stem not in seen:
                seen.add(stem)
                im_files.append(x)
        if not im_files:
            raise FileNotFoundError(f"{self.prefix}No images found in {img_path}")
        if self.fraction < 1:
            im_files = im_files[: round(len(im_files) * self.fraction)]
        return im_files

    def get_labels(self):
        nc = len(self.data["names"]) if "names" in self.data else self.data.get("nc")
        labels = []
        for im_file, lb_file in zip(self.im_files, img2label_paths(self.im_files)):
            lb = read_label(lb_file, nc)
            labels.append({"im_file": im_file, "shape": image_hw(im_file), "cls": lb[:, 0:1], "bboxes": lb[:, 1:],
                           "normalized": True, "bbox_format": "xywh"})
        return labels

    def update_labels(self, include_class):
        """base.py:132-149."""
        include = np.array(include_class).reshape(1, -1) if include_class is not None else None
        for lb in self.labels:
            if include is not None:
                j = (lb["cls"] == include).any(1)
                lb["cls"], lb["bboxes"] = lb["cls"][j], lb["bboxes"][j]
            if self.single_cls:
                lb["cls"][:, 0] = 0

    def set_rectangle(self):
        """base.py:270-288: sort by aspect ratio, one stride-rounded shape per batch."""
        bi = np.floor(np.arange(self.ni) / self.batch_size).astype(int)
        nb = bi[-1] + 1
        s = np.array([x.pop("shape") for x in self.labels])
        ar = s[:, 0] / s[:, 1]
        irect = ar.argsort()
        self.im_files = [self.im_files[i] for i in irect]
        self.labels = [self.labels[i] for i in irect]
        ar = ar[irect]
        shapes = [[1, 1]] * nb
        for i in range(nb):
            ari = ar[bi == i]
            mini, maxi = ari.min(), ari.max()
            if maxi < 1:
                shapes[i] = [maxi, 1]
            elif mini > 1:
                shapes[i] = [1, 1 / mini]
        self.batch_shapes = np.ceil(np.array(shapes) * self.imgsz / self.stride + self.pad).astype(int) * self.stride
        self.batch = bi

    # -- pixels
    def load_image(self, i, rect_mode=True):
        """base.py:151-187: decode, long side -> imgsz (cv2.resize INTER_LINEAR semantics, on the GPU), mosaic buffer."""
        im = self.ims[i]
        if im is None:
            if self.recipe_mode:                                            # worker process: the shape is all the transforms need
                src = file_image(i, self.labels[i]["shape"])
            else:
                fut = self._decoding.pop(i, None)
                src = fut.result() if fut is not None else self._decode_upload(i)
            h0, w0 = src.shape[:2]
            if rect_mode:
                r = self.imgsz / max(h0, w0)
                size = (min(math.ceil(h0 * r), self.imgsz), min(math.ceil(w0 * r), self.imgsz)) if r != 1 else (h0, w0)
            else:
                size = (self.imgsz, self.imgsz)
            if size == (h0, w0):
                im = src
            elif is_lazy(src):
                im = letterbox_image(src, size, size, 0, 0, 114)
            else:
                im = torch.empty((*size, 3), dtype=torch.uint8, device=src.device)
                K.image_letterbox(src, im, size, 0, 0, 114, reverse_c=False, chw=False)
            if self.augment:
                self.ims[i], self.im_hw0[i], self.im_hw[i] = im, (h0, w0), tuple(im.shape[:2])
                self.buffer.append(i)
                if 1 < len(self.buffer) >= self.max_buffer_length:
                    j = self.buffer.pop(0)
                    self.ims[j], self.im_hw0[j], self.im_hw[j] = None, None, None
            return im, (h0, w0), tuple(im.shape[:2])
        return self.ims[i], self.im_hw0[i], self.im_hw[i]

    def _decode_upload(self, i):
        """File -> uint8 HWC tensor on the device.  Also what the loader's background threads run for the coming batch: file
        reads and the pageable host-to-device copy both release the GIL, so they overlap the main thread's recipe code."""
        raw = torch.from_numpy(read_image(self.im_files[i]))
        return raw.to(self.device) if self.device.type == "cuda" else raw

    def get_image_and_label(self, index):
        """base.py:290-301 + dataset.py:204-229 (update_labels_info)."""
        src = self.labels[index]                                            # a flat dict of two small arrays: copy those, share the rest
        label = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in src.items() if k != "shape"}
        im, label["ori_shape"], label["resized_shape"] = self.load_image(index)
        label["img"] = DeviceImage.wrap(im)
        label["ratio_pad"] = (label["resized_shape"][0] / label["ori_shape"][0], label["resized_shape"][1] / label["ori_shape"][1])
        if self.rect:
            label["rect_shape"] = self.batch_shapes[self.batch[index]]
        bboxes = label.pop("bboxes")
        label["instances"] = Instances(bboxes, bbox_format=label.pop("bbox_format"), normalized=label.pop("normalized"))
        return label

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        return self.transforms(self.get_image_and_label(index))

    # -- transforms
    def build_transforms(self, hyp=None):
        """dataset.py:174-195."""
        if self.augment:
            hyp.mosaic = hyp.mosaic if self.augment and not self.rect else 0.0
            hyp.mixup = hyp.mixup if self.augment and not self.rect else 0.0
            transforms = v8_transforms(self, self.imgsz, hyp)
        else:
            transforms = Compose([LetterBox(new_shape=(self.imgsz, self.imgsz), scaleup=False, device=self.device)])
        transforms.append(Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=hyp.bgr if self.augment else 0.0, defer=True))
        return transforms

    def close_mosaic(self, hyp):
        """dataset.py:197-202."""
        hyp.mosaic = 0.0
        hyp.copy_paste = 0.0
        hyp.mixup = 0.0
        self.transforms = self.build_transforms(hyp)
        for notify in self.__dict__.get("_close_listeners", ()):           # a WorkerLoader: the same switch in every worker process
            notify(hyp)

    # -- batches
    @staticmethod
    def collate_fn(batch, out=None, dtype=torch.uint8, materialize=None):
        """dataset.py:231-248, plus the rendering of deferred images: every sample's recipe is executed by one launch
        straight into slot b of the batch tensor (``out`` — e.g. a captured graph's static input — or a new one; a
        float dtype also folds preprocess_batch's /255 into the same pass)."""
        new_batch = {}
        keys = batch[0].keys()
        values = list(zip(*[list(b.values()) for b in batch]))
        for i, k in enumerate(keys):
            value = values[i]
            if k == "img":
                if all(isinstance(v, DeviceImage) for v in value):
                    if materialize is not None:                         # recipes from worker processes: source nodes -> HBM tensors
                        value = [v.resolved(materialize) for v in value]
                    H, W = value[0].shape[:2]
                    if any(v.shape[:2] != (H, W) for v in value):
                        raise ValueError("collate_fn: images of one batch must share a shape")
                    if out is not None:
                        if tuple(out.shape[1:]) != (3, H, W) or out.shape[0] < len(value):
                            raise ValueError(f"collate_fn: `out` is {tuple(out.shape)}, the batch is {(len(value), 3, H, W)}")
                        imgs = out[:len(value)]                          # a short last batch fills a prefix of the buffer
                    else:
                        imgs = torch.empty((len(value), 3, H, W), dtype=dtype, device=value[0].device)
                    for b, v in enumerate(value):
                        v.render(dst=imgs[b], chw=True, reverse_c=v.final_reverse_c)
                    value = imgs
                else:
                    value = torch.stack(value, 0)
            if k in {"bboxes", "cls"}:
                value = torch.cat(value, 0)
            new_batch[k] = value
        new_batch["batch_idx"] = list(new_batch["batch_idx"])
        for i in range(len(new_batch["batch_idx"])):
            new_batch["batch_idx"][i] += i
        new_batch["batch_idx"] = torch.cat(new_batch["batch_idx"], 0)
        return new_batch


def build_yolo_dataset(cfg, img_path, batch, data, mode="train", rect=False, stride=32, device="cuda"):
    """data/build.py:104-126."""
    return YOLODataset(img_path=img_path, imgsz=cfg.imgsz, batch_size=batch, augment=mode == "train", hyp=cfg,
                       rect=getattr(cfg, "rect", False) or rect, cache=getattr(cfg, "cache", None) or None,
                       single_cls=getattr(cfg, "single_cls", False) or False, stride=int(stride),
                       pad=0.0 if mode == "train" else 0.5, prefix=f"{mode}: ", classes=getattr(cfg, "classes", None),
                       data=data, fraction=getattr(cfg, "fraction", 1.0) if mode == "train" else 1.0, device=device)


class InfiniteDataLoader:
    """data/build.py:30-86 + build_dataloader :129-157, in-process: an endless stream of batches whose order follows the
    reference's sampler semantics (seeded shuffle per epoch; a strided shard per rank like DistributedSampler, padded by
    wrap-around so every rank sees the same number of batches).  ``len()`` is the number of batches per epoch.
    ``out`` is the batch image buffer to render into: a tensor, None (allocate), or a callable ``n -> tensor | None``
    evaluated per batch (``DetectionTrainer.batch_buffer``: the captured graph's static input once it exists)."""

    def __init__(self, dataset, batch_size, shuffle=True, rank=-1, world_size=1, seed=0, prefetch=4, out=None, dtype=torch.uint8):
        self.dataset, self.batch_size, self.shuffle = dataset, min(batch_size, len(dataset)), shuffle
        self.rank, self.world_size = rank, max(world_size, 1)
        self.generator = torch.Generator()
        self.generator.manual_seed(6148914691236517205 + max(rank, 0))        # data/build.py:145-146
        self.seed, self.epoch = seed, 0
        self.out, self.dtype = out, dtype
        self.pool = ThreadPoolExecutor(max_workers=prefetch) if prefetch else None
        self._it = self._forever()

    def _epoch_indices(self):
        n = len(self.dataset)
        if self.rank == -1:
            return torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        # DistributedSampler: permutation seeded by (seed + epoch), padded to a multiple of world_size, strided shard
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        total = math.ceil(n / self.world_size) * self.world_size
        idx += idx[: total - n]
        return idx[self.rank:total:self.world_size]

    def __len__(self):
        n = len(self.dataset) if self.rank == -1 else math.ceil(len(self.dataset) / self.world_size)
        return math.ceil(n / self.batch_size)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _warm(self, indices):
        """Decode the files of the coming batch in the background (np.load / PIL release the GIL while reading);
        load_image picks the decoded array up instead of reading the file itself."""
        if self.pool is not None:
            ds = self.dataset
            for i in indices:
                if ds.ims[i] is None and i not in ds._decoding:
                    ds._decoding[i] = self.pool.submit(ds._decode_upload, i)

    def _forever(self):
        while True:
            idx = self._epoch_indices()
            batches = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
            for k, b in enumerate(batches):
                if k + 1 < len(batches):
                    self._warm(batches[k + 1])
                out = self.out(len(b)) if callable(self.out) else self.out    # e.g. the captured graph's static input, once it exists
                yield self.dataset.collate_fn([self.dataset[i] for i in b], out=out, dtype=self.dtype)
            self.epoch += 1

    def __iter__(self):
        for _ in range(len(self)):
            yield next(self._it)

    def reset(self):
        self._it = self._forever()


def seed_worker(worker_id=0):
    """data/build.py:89-93."""
    worker_seed = torch.initial_seed() % 2**32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def _worker_main(payload, wid, seed, tasks, results):
    """Body of a loader worker process: never touches the GPU.  Receives (generation, batch id, indices) tasks, answers with
    (generation, batch id, [sample dicts with DeviceImage recipes]); ("close_mosaic", hyp) and None (exit) are control messages."""
    import pickle
    ds = pickle.loads(payload)
    ds.recipe_mode = True
    random.seed(seed + wid)                                   # data/build.py:89-93 seed_worker: a stream of its own per worker
    np.random.seed((seed + wid) % 2**32)
    torch.manual_seed(seed + wid)
    while True:
        msg = tasks.get()
        if msg is None:
            return
        if msg[0] == "close_mosaic":
            ds.close_mosaic(msg[1])
            continue
        gen, bid, indices = msg
        try:
            # label tensors travel as numpy arrays: a torch tensor in a multiprocessing queue goes through a shared-memory file
            # descriptor each (3 per sample: 28 ms of unpickling per 64-sample batch in the training process, measured r03)
            samples = [{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in ds[i].items()} for i in indices]
            results.put((gen, bid, samples, None))
        except Exception as e:                                # noqa: BLE001 — reported to the training process, which raises
            import traceback
            results.put((gen, bid, None, f"{e!r}\n{traceback.format_exc()}"))


class _main_not_reimported:
    """While worker processes are started: hide ``__main__``'s file / spec from multiprocessing's spawn bootstrap.  A spawned child
    otherwise re-runs the parent's main script as ``__mp_main__`` — a training script without an ``if __name__ == '__main__'`` guard
    (fine with the reference's fork workers) would call ``train()`` again inside every loader worker (a second process group on the
    inherited RANK / MASTER_PORT, GPU work in a worker) and the worker would die in its bootstrap.  The worker entry point and the
    pickled dataset live in importable sy11 modules; nothing of ``__main__`` is needed there."""

    def __enter__(self):
        import sys
        self.main = sys.modules.get("__main__")
        self.saved = {}
        if self.main is not None:
            for k in ("__file__", "__spec__"):
                if k in self.main.__dict__:
                    self.saved[k] = self.main.__dict__[k]
            self.main.__dict__.pop("__file__", None)
            self.main.__dict__["__spec__"] = None
        return self

    def __exit__(self, *exc):
        if self.main is not None:
            self.main.__dict__.pop("__spec__", None)
            self.main.__dict__.update(self.saved)
        return False


class WorkerLoader(InfiniteDataLoader):
    """InfiniteDataLoader whose per-sample Python runs in ``procs`` worker processes (data/build.py:129-157 `workers`).

    Batch k goes to worker k % procs (the reference's DataLoader order); up to 2 x procs batches are in flight.  A worker returns
    recipes over LazyImage nodes (data/recipe.py); as soon as a batch's recipes arrive the training process starts decoding /
    uploading the source files it names (thread pool, GIL released), so by the time the batch is due its pixels are in HBM and
    collate_fn renders every sample with one launch.  Each worker owns a copy of the dataset (its own mosaic buffer and random
    streams, seeded seed + worker id), so — as in the reference — the samples depend on the number of workers."""

    def __init__(self, dataset, batch_size, procs, shuffle=True, rank=-1, world_size=1, seed=0, prefetch=4, out=None, dtype=torch.uint8):
        super().__init__(dataset, batch_size, shuffle=shuffle, rank=rank, world_size=world_size, seed=seed, prefetch=max(prefetch, 2), out=out, dtype=dtype)
        import multiprocessing as mp
        import pickle
        ctx = mp.get_context("spawn")                         # fresh interpreters: nothing of this process's GPU state is inherited
        payload = pickle.dumps(dataset)
        self.procs = max(int(procs), 1)
        self.results = ctx.Queue()
        self.tasks = [ctx.Queue() for _ in range(self.procs)]
        self.workers = [ctx.Process(target=_worker_main, args=(payload, w, 1000003 * (seed + 1) + 7919 * max(rank, 0), self.tasks[w], self.results), daemon=True)
                        for w in range(self.procs)]
        with _main_not_reimported():
            for p in self.workers:
                p.start()
        self.generation = 0
        self.materialize = Materializer(self._fetch, dataset.device)
        self._closed = False
        dataset.__dict__.setdefault("_close_listeners", []).append(self._close_mosaic_in_workers)

    def _close_mosaic_in_workers(self, hyp):
        """dataset.close_mosaic (dataset.py:197-202) reaches every worker; batches prepared before the switch are dropped."""
        for q in self.tasks:
            q.put(("close_mosaic", hyp))
        self.reset()

    # -- source pixels
    def _fetch(self, index):
        fut = self.dataset._decoding.pop(index, None)
        return fut.result() if fut is not None else self.dataset._decode_upload(index)

    def _sources(self, node, acc):
        if is_lazy(node):
            if node.op[0] == "file":
                acc.add(node.op[1])
            elif node.op[0] == "letterbox":
                if node.key() not in self.materialize.cache:
                    self._sources(node.op[1], acc)
            elif node.op[0] == "render":
                for t in node.op[1].tiles:
                    self._sources(t[0], acc)

    def _start_decodes(self, samples):
        need = set()
        for smp in samples:
            img = smp.get("img")
            if isinstance(img, DeviceImage):
                for t in img.tiles:
                    self._sources(t[0], need)
        ds = self.dataset
        for i in need:
            if ("file", i) not in self.materialize.cache and i not in ds._decoding and self.pool is not None:
                ds._decoding[i] = self.pool.submit(ds._decode_upload, i)

    # -- batches
    def _recipes(self):
        """Endless stream of per-batch sample lists (recipes over LazyImage nodes), in batch order."""
        depth = 2 * self.procs
        gen = self.generation
        pending, ready = [], {}
        next_id = 0

        def batches():
            while True:
                idx = self._epoch_indices()
                for i in range(0, len(idx), self.batch_size):
                    yield idx[i:i + self.batch_size]
                self.epoch += 1
        stream = batches()
        while True:
            while len(pending) < depth:                       # keep the workers `depth` batches ahead
                b = next(stream)
                self.tasks[next_id % self.procs].put((gen, next_id, b))
                pending.append(next_id)
                next_id += 1
            want = pending[0]
            while True:
                # take everything the workers have finished — blocking only while the batch that is due is still missing — and start
                # decoding / uploading the source files of EVERY arrived recipe at once: the files of batches k+1, k+2, ... are read
                # while batch k trains
                try:
                    if want in ready:
                        g, bid, samples, err = self.results.get(block=False)
                    else:
                        g, bid, samples, err = self._get_or_raise()
                except queue.Empty:
                    break
                if g != gen:
                    continue                                  # a batch prepared before reset(): discarded
                if err is not None:
                    raise RuntimeError(f"loader worker failed on batch {bid}: {err}")
                samples = [{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in smp.items()} for smp in samples]
                if self.dataset.device.type == "cuda":
                    self._start_decodes(samples)
                ready[bid] = samples
            yield ready.pop(pending.pop(0))

    def _get_or_raise(self, poll=2.0):
        """Blocking read of the result queue that notices a dead worker (a crash in its bootstrap, an unpicklable dataset, the OOM
        killer): a plain ``get()`` would wait forever for the batch that worker owed."""
        while True:
            try:
                return self.results.get(timeout=poll)
            except queue.Empty:
                dead = [(w, p.exitcode) for w, p in enumerate(self.workers) if not p.is_alive()]
                if dead:
                    raise RuntimeError("loader worker process(es) died: " + ", ".join(f"worker {w} exit code {c}" for w, c in dead) +
                                       " (run with workers=0 / SY11_LOADER_PROCS=0 for the in-process pipeline)") from None

    def _forever(self):
        for samples in self._recipes():
            out = self.out(len(samples)) if callable(self.out) else self.out
            yield self.dataset.collate_fn(samples, out=out, dtype=self.dtype, materialize=self.materialize)

    def reset(self):
        self.generation += 1
        self._it = self._forever()

    def close(self):
        if self._closed:
            return
        self._closed = True
        for q in self.tasks:
            try:
                q.put(None)
            except Exception:                                  # noqa: BLE001
                pass
        for p in self.workers:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()                                       # exact PIDs of processes this object started

    def __del__(self):
        try:
            self.close()
        except Exception:                                      # noqa: BLE001
            pass


def build_dataloader(dataset, batch, workers=8, shuffle=True, rank=-1, world_size=1, out=None, dtype=torch.uint8, procs=None):
    """data/build.py:129-157.  ``workers``: as in the reference, the loader's parallelism.  Training datasets (augment, no rect) get
    min(workers, CPUs - 1, 8) recipe WORKER PROCESSES (``WorkerLoader``) plus as many decode threads; validation / rect datasets and
    workers <= 1 stay in-process (``InfiniteDataLoader``: decode threads only).  ``procs`` (or SY11_LOADER_PROCS) overrides the count,
    0 = in-process."""
    ncpu = os.cpu_count() or 1
    threads = max(min(workers, ncpu), 0)
    if procs is None:
        env = os.environ.get("SY11_LOADER_PROCS")
        procs = int(env) if env is not None else (min(workers, max(ncpu - 1, 1), 8) if workers > 1 else 0)
    usable = dataset.augment and not dataset.rect and all("shape" in lb for lb in dataset.labels[:1])
    if procs > 0 and usable:
        return WorkerLoader(dataset, batch, procs, shuffle=shuffle, rank=rank, world_size=world_size, prefetch=max(threads, 2), out=out, dtype=dtype)
    return InfiniteDataLoader(dataset, batch, shuffle=shuffle, rank=rank, world_size=world_size, prefetch=threads, out=out, dtype=dtype)

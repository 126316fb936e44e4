"""Where the time of non_max_suppression's wrapper goes at the val shape (64 x (4 + 80) x 8400, conf 0.001, multi-label):
python tools/nms_wrapper_probe.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops as K
from sy11.utils import ops as U
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, nc, A = 64, 80, 8400
pred = torch.zeros(B, 4 + nc, A, device=dev)
pred[:, 0:2] = torch.rand(B, 2, A, generator=g, device=dev) * 640
pred[:, 2:4] = torch.rand(B, 2, A, generator=g, device=dev) * 100 + 10
sc = torch.rand(B, nc, A, generator=g, device=dev)
pred[:, 4:] = torch.where(sc > 0.99, sc, torch.zeros_like(sc) )          # ~1 % of the (anchor, class) pairs above conf
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("whole non_max_suppression: %.3f ms" % t(lambda: U.non_max_suppression(pred.clone(), 0.001, 0.7, multi_label=True, max_det=300)))
print("  pred.clone(): %.3f ms" % t(lambda: pred.clone()))
key, anchor, cidx, counts, by_class = K.nms_candidates(pred, nc, 0.001, True, True)
print("  candidates (2 kernels + cumsum + host read): %.3f ms, %d candidates, per-class segments: %s" % (t(lambda: K.nms_candidates(pred, nc, 0.001, True, True)), key.numel(), by_class))
print("  xywh2xyxy in (B,4,A): %.3f ms" % t(lambda: torch.cat((pred[:, 0:2] - pred[:, 2:4] / 2, pred[:, 0:2] + pred[:, 2:4] / 2), 1)))
print("  sort 64-bit keys (stable): %.3f ms" % t(lambda: torch.sort(key, stable=True)))
order = torch.sort(key, stable=True).indices
ks = key[order]; seg = ks >> 32; img = seg // nc if by_class else seg; a_idx = anchor[order].long(); c = cidx[order].long()
print("  gathers key/anchor/class + conf: %.3f ms" % t(lambda: (key[order], anchor[order].long(), cidx[order].long(), ((~ks) & 0xFFFFFFFF).to(torch.int32).view(torch.float32))))
xy = torch.cat((pred[:, 0:2] - pred[:, 2:4] / 2, pred[:, 0:2] + pred[:, 2:4] / 2), 1)
print("  box gather + class shift: %.3f ms" % t(lambda: (xy[img, :, a_idx] + (c.float() * 7680).unsqueeze(1)).contiguous()))
box = (xy[img, :, a_idx] + (c.float() * 7680).unsqueeze(1)).contiguous()
print("  suppression kernels (one bit matrix per (image, class)): %.3f ms" % t(lambda: K.nms_sorted_segments(box, seg, B * nc, 0.7, 300)))
keep = K.nms_sorted_segments(box, seg, B * nc, 0.7, 300)
def tail():
    kept = torch.nonzero(keep, as_tuple=True)[0]
    kept = kept[torch.sort(order[kept]).indices]
    kept = kept[torch.sort((img[kept] << 32) | (ks[kept] & 0xFFFFFFFF), stable=True).indices]
    kimg = img[kept]
    kcount = torch.bincount(kimg, minlength=B)
    krank = torch.arange(kept.numel(), device=dev) - (torch.cumsum(kcount, 0) - kcount)[kimg]
    kept = kept[krank < 300]
    sizes = torch.bincount(img[kept], minlength=B).tolist()
    return sizes
print("  tail (nonzero, two re-sorts of the survivors, ranks, sizes host read): %.3f ms" % t(tail))

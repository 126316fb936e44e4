"""Debug helper: where does the fused render differ from the oracle?  (warp only / hsv only)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import numpy as np, torch
from oracle import image_ref as IR
from sy11 import ops as K

g = np.random.default_rng(0)
img = g.integers(0, 256, (256, 256, 3), dtype=np.uint8)
src = torch.from_numpy(img).cuda()
for gains in ((1.0, 1.0, 1.0), (1.013, 0.41, 1.37)):
    lut = np.stack(IR.hsv_luts(np.array(gains)))
    dst = torch.empty((256, 256, 3), dtype=torch.uint8, device="cuda")
    K.image_mosaic_warp([(src, 0, 0, 256, 256, 0, 0)], (256, 256), dst, hsv_lut=lut, chw=False)
    got = dst.cpu().numpy(); want = IR.random_hsv(img, np.array(gains))
    bad = np.argwhere((got != want).any(-1))
    print("hsv gains", gains, "mismatch px", len(bad))
    for y, x in bad[:6]:
        hsv = IR.cv2_bgr2hsv_u8(img[y:y+1, x:x+1])[0, 0]
        print("  bgr", img[y, x], "hsv", hsv, "lut->", lut[0][hsv[0]], lut[1][hsv[1]], lut[2][hsv[2]], "got", got[y, x], "want", want[y, x])
for t in range(6):
    M = np.array([[g.uniform(0.5, 1.5), g.uniform(-0.3, 0.3), g.uniform(-9, 9)], [g.uniform(-0.3, 0.3), g.uniform(0.5, 1.5), g.uniform(-9, 9)]], np.float32)
    dst = torch.empty((200, 220, 3), dtype=torch.uint8, device="cuda")
    from sy11.data.augment import invert_affine
    K.image_mosaic_warp([(src, 0, 0, 256, 256, 0, 0)], (256, 256), dst, minv=invert_affine(M), chw=False)
    want = IR.cv2_warp_affine_u8(img, M, (220, 200))
    bad = np.argwhere((dst.cpu().numpy() != want).any(-1))
    print("warp", t, "mismatch px", len(bad), bad[:4].tolist())

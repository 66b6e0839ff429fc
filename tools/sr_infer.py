#!/usr/bin/env python
"""Inference harness reproducing the reference's ``test_image/test.py`` (lines 9-40) on the HIP path
with PIL instead of cv2:  python tools/sr_infer.py <model.pth|synthetic> <in_dir> <out_dir> [fp16|fp32]

Per image: RGB /255 -> NCHW float32 -> RRDB_Net(3,3,64,23,...) -> clamp(0,1) -> *255 round -> PNG."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from esrganplus_amd import architecture as arch, synth


def main():
    model_path, in_dir, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    prec = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
    dev = torch.device('cuda')
    model = arch.RRDB_Net(3, 3, 64, 23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu',
                          mode='CNA', res_scale=1, upsample_mode='upconv')
    sd = synth.rrdbnet_state_dict(23, 0) if model_path == 'synthetic' else torch.load(model_path, map_location='cpu')
    model.load_state_dict(sd, strict=False)           # test_image/test.py:17
    model.eval()
    for _, v in model.named_parameters():
        v.requires_grad = False
    model = model.to(dev).set_precision(prec)
    os.makedirs(out_dir, exist_ok=True)
    for idx, path in enumerate(sorted(glob.glob(os.path.join(in_dir, '*'))), 1):
        base = os.path.splitext(os.path.basename(path))[0]
        img = np.array(Image.open(path).convert('RGB')).astype(np.float64) / 255
        x = torch.from_numpy(np.transpose(img, (2, 0, 1))).float().unsqueeze(0).to(dev)
        with torch.no_grad():
            out = model(x).data.squeeze().float().cpu().clamp_(0, 1).numpy()
        out = (np.transpose(out, (1, 2, 0)) * 255.0).round().astype(np.uint8)
        Image.fromarray(out).save(os.path.join(out_dir, '%s_rlt.png' % base))
        print(idx, base, img.shape[:2], '->', out.shape[:2])


if __name__ == '__main__':
    main()

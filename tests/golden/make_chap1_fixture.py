"""Fixture from the reference's own output file path_tracing_1chap/sample_4096.png (its only converged render of the
c1:355-400 scene, 512x512, 8-bit, gamma 1/1.9 with truncation, c1:340-342): linearised (v/255)^1.9 and box-averaged to
64x64.  Data, not source.  Run where /root/reference is mounted: python tests/golden/make_chap1_fixture.py"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
img = np.asarray(Image.open("/root/reference/path_tracing_1chap/sample_4096.png").convert("RGB")).astype(np.float64)
lin = (img / 255.0) ** 1.9
h, w, _ = lin.shape
small = lin.reshape(64, h // 64, 64, w // 64, 3).mean(axis=(1, 3)).astype(np.float32)
np.save(os.path.join(HERE, "chap1_sample4096_linear_64x64.npy"), small)
print(img.shape, small.shape, small.mean())

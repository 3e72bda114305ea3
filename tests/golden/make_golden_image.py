"""Fixture for the caller-API parity test: a 160 x 256 crop of the reference's sample image tests/data/color.jpg, decoded
once in the build container (PIL) and stored as raw BGR uint8 pixels, because /root/reference does not exist on the GPU
box.  python tests/golden/make_golden_image.py -> tests/golden/color_jpg_crop_bgr.npz"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
with Image.open('/root/reference/tests/data/color.jpg') as im:
    rgb = np.asarray(im.convert('RGB'))
crop = np.ascontiguousarray(rgb[64:224, 128:384, ::-1])          # BGR like cv2.imread
np.savez_compressed(os.path.join(HERE, 'color_jpg_crop_bgr.npz'), bgr=crop, source='tests/data/color.jpg[64:224,128:384]')
print(crop.shape, crop.dtype)

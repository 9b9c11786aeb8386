"""Host-side mirror of the reference's anti_instagram k-means entry points (k-means part).

Reference: /root/reference/src/anti_instagram/include/anti_instagram/kmeans.py
  CENTERS, CENTERS2 (:9-10), getimgdatapts (:14-19), runKMeans (:22-47).
Same names, same arguments, same return values; the clustering itself runs on the GPU (lf_kmeans, k_kmeans.hip).
"""
from collections import Counter

import numpy as np

from .config import default_config
from .frontend import FrontEnd

# kmeans.py:9-10 (B, G, R): dark grey, red, yellow, white / dark grey, yellow, white
CENTERS2 = np.array([[60, 60, 60], [60, 60, 240], [50, 240, 240], [240, 240, 240]])
CENTERS = np.array([[60, 60, 60], [50, 240, 240], [240, 240, 240]])

_fe = None


def _frontend():
    """One small handle for the clustering calls (created on first use: fails loudly without the HIP library / a GPU)."""
    global _fe
    if _fe is None:
        _fe = FrontEnd(default_config("parity"), max_frames=1, max_lines_per_color=64)
    return _fe


def getimgdatapts(cv2img):
    """kmeans.py:14-19: the pixels as an [x*y, 3] array, COLUMN major (the reference transposes the image first)."""
    x, y, p = cv2img.shape
    return np.transpose(np.reshape(cv2img.transpose(), [p, x * y]))


def runKMeans(cv_img, num_colors, init, frontend=None):
    """kmeans.py:22-47.  Returns (trained_centers [num_colors, 3] f64, labelcount Counter{cluster: members}, score)."""
    imgdata = getimgdatapts(cv_img[-100:, :, :])          # the reference's "arbitrary cut off"
    init = np.asarray(init, np.float64)
    if init.shape != (num_colors, 3):
        raise ValueError("init must be a [num_colors, 3] array of B, G, R centres")
    fe = frontend if frontend is not None else _frontend()
    centers, counts, inertia, _ = fe.kmeans(imgdata, init, max_iter=25, tol=1e-4)
    labelcount = Counter()
    for i in range(num_colors):
        labelcount[i] = int(counts[i])
    return centers, labelcount, -inertia

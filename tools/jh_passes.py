"""Correction passes and phase times of the device Huffman decoder per frame (LF_JH_DEBUG=1): six synthetic and four camera streams."""
import io, os, sys
os.environ.setdefault("LF_JH_DEBUG", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
from PIL import Image
from lane_slam_amd import FrontEnd, default_config, synth
streams = []
for i in range(6):
    b = io.BytesIO()
    Image.fromarray(synth.make_frame(i)[..., ::-1].copy()).save(b, "JPEG", quality=80, subsampling=2)
    streams.append(b.getvalue())
zj = np.load("tests/golden/real_jpegs.npz")
streams += [bytes(zj["jpeg%02d" % k]) for k in range(4)]
fe = FrontEnd(default_config("fullres"), max_frames=16)
rf, st = fe.decode_jpeg_batch(streams, n_threads=2, entropy="gpu")
print([len(s) for s in streams], st)
fe.close()

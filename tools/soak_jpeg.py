#!/usr/bin/env python3
"""Soak test of the JPEG ingest path on the GPU box: random images (synthetic lane frames, noise, gradients,
random sizes) encoded by Pillow with random quality / chroma subsampling / restart intervals / optimised
tables, decoded by lf_jpeg_decode_batch and compared bit for bit with Pillow's libjpeg-turbo and the oracle.

    python tools/soak_jpeg.py [--n 600]
"""
import argparse, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from lane_slam_amd import FrontEnd, default_config, synth
from oracle.oracle import jpeg_decode

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=600)
args = ap.parse_args()
rng = np.random.default_rng(777)
fe = FrontEnd(default_config("parity"), max_frames=16, max_lines_per_color=16)
bad = 0
done = 0
while done < args.n:
    rows, cols = (480, 640) if rng.random() < 0.4 else (int(rng.integers(1, 200)), int(rng.integers(1, 300)))
    batch, refs = [], []
    for _ in range(8):
        kind = rng.integers(0, 4)
        if kind == 0 and (rows, cols) == (480, 640):
            img = synth.make_frame(int(rng.integers(0, 1 << 30)))
        elif kind == 1:
            img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        elif kind == 2:
            yy, xx = np.mgrid[0:rows, 0:cols]
            img = np.stack([(xx * 3 + yy) % 256, (yy * 2) % 256, (xx + yy * 5) % 256], -1).astype(np.uint8)
        else:
            img = np.full((rows, cols, 3), rng.integers(0, 256, 3), np.uint8)
            img[rows // 3:, cols // 2:] = rng.integers(0, 256, 3)
        kw = {"quality": int(rng.integers(5, 100)), "subsampling": int(rng.integers(0, 3))}
        r = rng.random()
        if r < 0.2: kw["optimize"] = True
        elif r < 0.4: kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        elif r < 0.5: kw["restart_marker_rows"] = int(rng.integers(1, 4))
        gray = rng.random() < 0.1
        pil = Image.fromarray(img[..., 1].copy()) if gray else Image.fromarray(img[..., ::-1].copy())
        try:
            b = io.BytesIO()
            pil.save(b, "JPEG", **kw)
        except OSError:                      # some restart settings are refused by the encoder for tiny images
            b = io.BytesIO()
            pil.save(b, "JPEG", quality=kw["quality"], subsampling=kw["subsampling"])
        data = b.getvalue()
        batch.append(data)
        refs.append(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1])
    got, status = fe.decode_jpeg_batch(batch, rows=rows, cols=cols, n_threads=4)
    for i in range(len(batch)):
        ok = status[i] == 0 and np.array_equal(got[i], refs[i]) and np.array_equal(got[i], jpeg_decode(batch[i]))
        if not ok:
            bad += 1
            print("MISMATCH", rows, cols, status[i], flush=True)
    done += len(batch)
print("jpeg soak: %d streams, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)

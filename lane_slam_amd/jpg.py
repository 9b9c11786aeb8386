"""Mirror of the reference's duckietown_utils.jpg decode helper on the HIP ingest path
(ref: src/duckietown/include/duckietown_utils/jpg.py:21-31).

    image_cv_from_jpg(data) -> BGR u8 (rows, cols, 3)      # same name, argument and error behaviour

The reference calls cv2.imdecode(np.fromstring(data, np.uint8), cv2.IMREAD_COLOR) and raises ValueError
when it returns None.  Here the stream goes through lf_jpeg_decode_batch (Huffman decoding on the host,
IDCT / upsampling / colour conversion on the GPU, bit identical to libjpeg-turbo); streams this decoder does
not cover (progressive, arithmetic, CMYK, exotic sampling) raise ValueError as well -- say so if you hit one.
Per-frame calls are for drop-in use; a pipeline should batch with FrontEnd.decode_jpeg_batch."""
import ctypes

import numpy as np

from . import _lib
from .config import default_config
from .frontend import FrontEnd

_decoder = None


def _handle():
    global _decoder
    if _decoder is None:
        # any configuration will do: the ingest entry points only use the handle's device and stream
        _decoder = FrontEnd(default_config("parity"), max_frames=1, max_lines_per_color=16)
    return _decoder


def jpg_info(data):
    """(rows, cols, components, hmax, vmax) of a JPEG stream; ValueError if it is not one this decoder reads."""
    lib = _lib.load()
    buf = np.frombuffer(bytes(data), np.uint8)
    v = [ctypes.c_int() for _ in range(5)]
    rc = lib.lf_jpeg_info(buf.ctypes.data_as(ctypes.c_void_p), buf.size, *[ctypes.byref(x) for x in v])
    if rc != 0:
        raise ValueError("Could not decode image (lf_jpeg_info returned %d). This is usual a sign of data corruption." % rc)
    return tuple(x.value for x in v)


def image_cv_from_jpg(data):
    """ Returns an OpenCV-style BGR image from a JPEG string """
    rows, cols = jpg_info(data)[:2]
    frames, status = _handle().decode_jpeg_batch([data], rows=rows, cols=cols, n_threads=1)
    if status[0] != 0:
        msg = 'Could not decode image (lf_jpeg_decode_batch status %d). ' % int(status[0])
        msg += 'This is usual a sign of data corruption.'
        raise ValueError(msg)
    return frames[0]

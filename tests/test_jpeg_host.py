"""CPU check of the product's host-side JPEG entropy decoder (lane_slam_amd/csrc/jpeg_entropy.cpp,
built into a test harness by tests/hostsim/jpeg_host.cpp): its sparse coefficient lists must equal the
oracle's dense coefficient dump for every golden stream, and it must refuse what the oracle refuses.
The device half (IDCT, upsampling, colour) is covered by the -m gpu tests."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host():
    so = os.path.join(HERE, "hostsim", "_build", "libjpeghost.so")
    src = os.path.join(HERE, "hostsim", "jpeg_host.cpp")
    deps = [src] + [os.path.join(HERE, "..", "lane_slam_amd", "csrc", f) for f in ("jpeg_entropy.cpp", "jpeg_entropy.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.hs_jpeg_coefficients.restype = ctypes.c_int
    lib.hs_jpeg_peek.restype = ctypes.c_int
    return lib


@pytest.fixture(scope="module")
def vectors(golden_dir):
    return np.load(os.path.join(golden_dir, "jpeg_vectors.npz"))


def _coefs(lib, data, cap=1 << 16):
    buf = np.frombuffer(bytes(data), np.uint8)
    dense = np.zeros((cap, 64), np.int16)
    qt = np.zeros((3, 64), np.uint16)
    layout = (ctypes.c_int * 6)()
    nb, ne = ctypes.c_int(), ctypes.c_long()
    rc = lib.hs_jpeg_coefficients(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size),
                                  dense.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(nb),
                                  qt.ctypes.data_as(ctypes.c_void_p), layout, ctypes.byref(ne))
    return rc, dense[: nb.value], list(layout), ne.value


def test_host_entropy_decoder_matches_oracle_coefficients(host, vectors):
    from oracle.oracle import jpeg_coefficients, jpeg_info
    for name in [str(n) for n in vectors["names"]]:
        data = bytes(vectors["jpeg_" + name])
        rc, dense, layout, n_entries = _coefs(host, data)
        assert rc == 0, name
        ref = jpeg_coefficients(data)
        assert dense.shape == ref.shape and np.array_equal(dense, ref), name
        assert n_entries == int(np.count_nonzero(ref)), name              # the lists hold the non-zeros and nothing else
        rows, cols, ncomp, hmax, vmax = jpeg_info(data)
        assert layout[:3] == [ncomp, hmax, vmax]
        assert layout[3] == -(-cols // (8 * hmax)) and layout[4] == -(-rows // (8 * vmax))


def test_host_decoder_refuses_bad_streams(host, vectors):
    LF_ERR_UNSUPPORTED, LF_ERR_DECODE = -5, -6
    assert _coefs(host, bytes(vectors["jpeg_progressive"]))[0] == LF_ERR_UNSUPPORTED
    assert _coefs(host, bytes(vectors["jpeg_truncated"]))[0] == LF_ERR_DECODE
    good = bytes(vectors["jpeg_lane_q75_420"])
    for junk in (b"\xff\xd8", b"garbage", good[:100], good[:-200]):
        assert _coefs(host, junk)[0] in (LF_ERR_DECODE, LF_ERR_UNSUPPORTED)
    # a flipped bit inside the entropy data must never crash; it either still parses or is refused
    rng = np.random.default_rng(9)
    for _ in range(200):
        b = bytearray(good)
        b[int(rng.integers(len(good) // 2, len(good) - 2))] ^= 1 << int(rng.integers(0, 8))
        assert _coefs(host, bytes(b))[0] in (0, LF_ERR_DECODE, LF_ERR_UNSUPPORTED)


def test_host_decoder_bounds_allocation_by_the_stream(host, vectors):
    """A tiny stream whose SOF declares 8192x8192 must not make the decoder allocate for 8192x8192: with
    the batch's declared size it is refused right after its headers, without it by the stream-length bound."""
    data = bytearray(bytes(vectors["jpeg_lane_q75_420"]))
    i = data.find(b"\xff\xc0")
    assert i > 0
    rows0, cols0 = (data[i + 5] << 8) | data[i + 6], (data[i + 7] << 8) | data[i + 8]
    data[i + 5:i + 9] = bytes([0x20, 0x00, 0x20, 0x00])          # 8192 x 8192
    buf = np.frombuffer(bytes(data), np.uint8)
    host.hs_jpeg_decode_expect.restype = ctypes.c_int
    alloc = ctypes.c_long()
    rc = host.hs_jpeg_decode_expect(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size), rows0, cols0, ctypes.byref(alloc))
    assert rc == -1 and alloc.value == 0                         # LF_ERR_BAD_ARG, nothing sized from the stream
    rc = host.hs_jpeg_decode_expect(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size), 0, 0, ctypes.byref(alloc))
    assert rc == -6 and alloc.value <= 64 * buf.size + 4096      # LF_ERR_DECODE; at most a small multiple of the stream
    # the genuine stream still decodes when its size is declared
    good = np.frombuffer(bytes(vectors["jpeg_lane_q75_420"]), np.uint8)
    rc = host.hs_jpeg_decode_expect(good.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(good.size), rows0, cols0, ctypes.byref(alloc))
    assert rc == 0


def test_host_decoder_builtin_tables(host, vectors):
    """DHT-less streams: the product's copy of the Annex K tables gives the same coefficients as the stream's own."""
    from test_oracle_jpeg import strip_dht
    for name in ("lane_q75_420", "noise_q30_444", "gray"):
        data = bytes(vectors["jpeg_" + name])
        rc0, a, _, _ = _coefs(host, data)
        rc1, b, _, _ = _coefs(host, strip_dht(data))
        assert rc0 == 0 and rc1 == 0 and np.array_equal(a, b), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/anti_instagram/annotation-tool/images"),
                    reason="the reference checkout (real camera JPEGs) is only present in the build container")
def test_host_decoder_on_reference_camera_images(host):
    """The product's entropy decoder on the reference's 173 real camera frames: coefficients equal the oracle's
    (which equals libjpeg-turbo on these files, tests/test_oracle_jpeg.py)."""
    import glob
    from oracle.oracle import jpeg_coefficients
    files = sorted(glob.glob("/root/reference/src/anti_instagram/annotation-tool/images/*.jpg"))[::4]
    for f in files:
        data = open(f, "rb").read()
        rc, dense, _, _ = _coefs(host, data)
        assert rc == 0 and np.array_equal(dense, jpeg_coefficients(data)), os.path.basename(f)


def test_decoders_survive_hostile_input_under_sanitizers(vectors, tmp_path):
    """AddressSanitizer + UBSan build of the product's host decoder and the oracle's (tests/hostsim/jpeg_fuzz.cpp):
    truncations, bit flips, byte splices and deletions of the golden streams.  No memory error, no undefined
    behaviour, and whenever both accept a mutated stream they agree on every coefficient."""
    exe = os.path.join(HERE, "hostsim", "_build", "jpeg_fuzz")
    srcs = [os.path.join(HERE, "hostsim", "jpeg_fuzz.cpp"), os.path.join(HERE, "..", "oracle", "lf_oracle_jpeg.c"),
            os.path.join(HERE, "..", "lane_slam_amd", "csrc", "jpeg_entropy.cpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        obj = os.path.join(os.path.dirname(exe), "lf_oracle_jpeg_asan.o")
        san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1"]
        subprocess.check_call(["gcc", "-std=c11", "-c", "-I" + os.path.join(HERE, "..", "oracle"), srcs[1], "-o", obj] + san)
        subprocess.check_call(["g++", "-std=c++17", "-Wall", srcs[0], obj, "-o", exe] + san)
    files = []
    for name in ("lane_q75_420", "noise_q30_444", "smooth_q95_422", "gray", "lane_rst_420", "noise_opt_420", "tiny_3x5_420", "lane_rgb_444"):
        p = tmp_path / (name + ".jpg")
        p.write_bytes(bytes(vectors["jpeg_" + name]))
        files.append(str(p))
    out = subprocess.run([exe] + files + ["--iters", "250"], capture_output=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, (out.stdout + out.stderr).decode()[-2000:]
    assert b"identical coefficients" in out.stdout

"""BinaryDescriptor::Params on the host side (no GPU): read / write of the reference's FileStorage keys
(ref: src/line_descriptor/src/binary_descriptor_custom.cpp:108-116, 189-204) and the oracle's width-of-band switch."""
import numpy as np

from lane_slam_amd import BinaryDescriptorParams, default_config, synth
from oracle import oracle as O


def test_params_defaults_read_write():
    p = BinaryDescriptorParams()
    assert (p.numOfOctave_, p.widthOfBand_, p.reductionRatio, p.ksize_) == (1, 7, 2, 5)                 # :110-116
    out = p.write()
    assert out == {"numOfOctave_": 1, "numOfBand_": 9, "widthOfBand_": 7, "reductionRatio": 2}          # :197-204 (no ksize_, numOfBand_ added)
    q = BinaryDescriptorParams().read({"numOfOctave_": 3, "widthOfBand_": 9, "reductionRatio": 2, "ksize_": 11})
    assert (q.numOfOctave_, q.widthOfBand_, q.reductionRatio, q.ksize_) == (3, 9, 2, 5)                 # read() leaves ksize_ alone (:189-194)
    assert BinaryDescriptorParams().read({}).widthOfBand_ == 0                                          # an empty FileNode converts to 0
    assert BinaryDescriptorParams().read(q.write()).write() == q.write()


def test_oracle_width_of_band_switch():
    cfg = default_config("parity")
    o = O.Oracle(cfg)
    f = synth.make_batch(1, seed0=70)[0]
    base = o.process_frame(f)
    try:
        o.set_width_of_band(9)
        wide = o.process_frame(f)
        assert wide["n"] == base["n"] and np.array_equal(wide["lines"], base["lines"])
        assert base["n"] > 0 and not np.array_equal(wide["desc"], base["desc"])
        assert np.all(np.isfinite(wide["desc"])) and np.allclose(np.linalg.norm(wide["desc"], axis=1), 1.0, atol=1e-5)
    finally:
        o.set_width_of_band(7)
    again = o.process_frame(f)
    assert np.array_equal(again["desc"], base["desc"]) and np.array_equal(again["code"], base["code"])

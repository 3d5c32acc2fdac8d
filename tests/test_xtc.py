"""codlad_amd/utils/xtc.py: the .xtc trajectory format restated from the published xdrfile algorithm (parity unpinned: no
third-party reader or sample file here).  Held to: its own round trip at the format's precision, the frame layout, the raw form
of small frames, and a hand-packed stream that exercises the decoder's run branch."""
import struct

import numpy as np
import pytest

from codlad_amd.utils import xtc


def test_round_trip_at_the_formats_precision(tmp_path):
    rng = np.random.default_rng(0)
    xyz = (rng.normal(0, 15.0, (5, 700, 3)) + np.array([30.0, -20.0, 5.0])).astype(np.float32)     # Angstrom
    path = str(tmp_path / "t.xtc")
    xtc.write_xtc(path, xyz, time=np.arange(5) * 2.5, step=np.arange(5) * 10)
    back, time, step, box = xtc.read_xtc(path)
    assert back.shape == xyz.shape and np.allclose(time, np.arange(5) * 2.5) and step.tolist() == [0, 10, 20, 30, 40]
    assert (box == 0).all()
    assert np.abs(back - xyz).max() <= 0.5 * 10.0 / 1000.0 + 1e-4        # half a unit of 1 / precision nm, in Angstrom
    # smaller than the raw floats
    assert (tmp_path / "t.xtc").stat().st_size < 0.5 * xyz.size * 4


def test_frame_layout(tmp_path):
    xyz = np.zeros((1, 12, 3), dtype=np.float32)
    xyz[0, :, 0] = np.arange(12) * 1.5
    xyz[0, :, 1] = -np.arange(12)
    path = str(tmp_path / "f.xtc")
    xtc.write_xtc(path, xyz, time=[7.0], step=[3])
    raw = open(path, "rb").read()
    magic, natoms, step, time = struct.unpack_from(">iiif", raw, 0)
    assert (magic, natoms, step, time) == (1995, 12, 3, 7.0)
    assert struct.unpack_from(">9f", raw, 16) == (0.0,) * 9
    assert struct.unpack_from(">i", raw, 52)[0] == 12
    precision, = struct.unpack_from(">f", raw, 56)
    minint, maxint = struct.unpack_from(">3i", raw, 60), struct.unpack_from(">3i", raw, 72)
    assert precision == 1000.0 and minint == (0, -1100, 0) and maxint == (1650, 0, 0)
    smallidx, nbytes = struct.unpack_from(">ii", raw, 84)
    # ranges 1651 x 1101 x 1: bit length of their product, plus the flag bit, per atom
    per_atom = (1651 * 1101).bit_length() + 1
    assert smallidx == 9 and nbytes == (12 * per_atom + 7) // 8 and len(raw) == 92 + nbytes + (-nbytes % 4)
    # the first atom: integers (0, 0, 0) -> relative (0, 1100, 0) -> the number 1100, least significant byte first
    first = int.from_bytes(raw[92:96], "big") >> (32 - per_atom)
    nb = per_atom - 1
    lo, hi, top = (first >> (per_atom - 8)) & 0xff, (first >> (per_atom - 16)) & 0xff, (first >> 1) & ((1 << (nb - 16)) - 1)
    assert lo + 256 * hi + 65536 * top == 1100 and first & 1 == 0


def test_small_frames_are_raw_floats(tmp_path):
    xyz = np.arange(27, dtype=np.float32).reshape(1, 9, 3)
    path = str(tmp_path / "s.xtc")
    xtc.write_xtc(path, xyz)
    raw = open(path, "rb").read()
    assert len(raw) == 56 + 9 * 12
    assert np.allclose(struct.unpack_from(">27f", raw, 56), xyz.reshape(-1) / 10.0)
    assert np.allclose(xtc.read_xtc(path)[0], xyz)


def test_wide_ranges_use_per_coordinate_fields(tmp_path):
    xyz = np.zeros((1, 10, 3), dtype=np.float32)
    xyz[0, :, 0] = np.linspace(-1e6, 1e6, 10)               # range x 100 > 2^24: three separate bit fields per atom
    xyz[0, :, 2] = np.arange(10)
    path = str(tmp_path / "w.xtc")
    xtc.write_xtc(path, xyz)
    back = xtc.read_xtc(path)[0]
    assert np.abs(back - xyz).max() <= 0.1                    # float32 rounding of 1e5 nm x 1000


class _Writer:
    def __init__(self):
        self.bits = []

    def put(self, n, v):
        self.bits += [(v >> (n - 1 - j)) & 1 for j in range(n)]

    def ints(self, nbits, sizes, nums):                       # sendints: mixed-radix number, least significant byte first
        value = (nums[0] * sizes[1] + nums[1]) * sizes[2] + nums[2]
        full, rest = divmod(nbits, 8)
        for b in range(full):
            self.put(8, (value >> (8 * b)) & 0xff)
        if rest:
            self.put(rest, value >> (8 * full))


def test_reader_decodes_a_run(tmp_path):
    """A hand-packed frame as a third-party writer would lay it out: atom 0 in full, then a run of two small atoms (the first
    of which changes places with the full one), then atom 3 in full with the flag cleared."""
    n, precision, smallidx = 10, 1000.0, 12                  # table entry 16: small numbers are offsets in [0, 16), minus 8
    ints = np.array([[100, 200, 300], [103, 198, 301], [101, 204, 297]] + [[50 * k, 10 * k, 400 - 7 * k] for k in range(3, 10)])
    minint, maxint = ints.min(0), ints.max(0)
    sizeint = [int(maxint[k] - minint[k] + 1) for k in range(3)]
    bitsize = xtc._bit_length_of_product(sizeint)
    small, sn = xtc.MAGICINTS[smallidx], xtc.MAGICINTS[smallidx] // 2
    w = _Writer()
    # the writer sends the SECOND atom of the file in full and the first as the run's first small atom (they are swapped)
    w.ints(bitsize, sizeint, list(ints[1] - minint))
    w.put(1, 1)
    w.put(5, 6 + 0 + 1)                                      # run of 2 atoms (6 numbers), table index unchanged: run + is_smaller + 1
    w.ints(smallidx, [small] * 3, list(ints[0] - ints[1] + sn))
    w.ints(smallidx, [small] * 3, list(ints[2] - ints[0] + sn))
    for k in range(3, 10):
        w.ints(bitsize, sizeint, list(ints[k] - minint))
        if k == 3:
            w.put(1, 1)
            w.put(5, 0 + 0 + 1)                              # run length back to 0
        else:
            w.put(1, 0)
    data = np.packbits(np.array(w.bits, dtype=np.uint8)).tobytes()
    raw = struct.pack(">iiif", 1995, n, 0, 0.0) + struct.pack(">9f", *([0.0] * 9)) + struct.pack(">i", n)
    raw += struct.pack(">f3i3ii", precision, *[int(v) for v in minint], *[int(v) for v in maxint], smallidx)
    raw += struct.pack(">i", len(data)) + data + b"\0" * (-len(data) % 4)
    path = tmp_path / "r.xtc"
    path.write_bytes(raw)
    back = xtc.read_xtc(str(path))[0]
    assert np.allclose(back[0], ints / precision * 10.0, atol=1e-4)

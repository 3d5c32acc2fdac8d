"""GROMACS .xtc trajectories without mdtraj (reference test.py:787-803 writes the generated and the true ensemble through
`md.Trajectory.save_xtc`; utils/dataset_module.py:150-160 reads the Atlas trajectories through `md.load`).

FORMAT, restated from the published xdrfile library (xdrfile.c / xdrfile_xtc.c, the code mdtraj and GROMACS share; it is
neither part of the reference tree nor installed here, and no .xtc file ships with the reference - **parity unpinned**:
what the tests can hold this file to is its own round trip, the frame header's layout and the raw-float form of frames of at
most nine atoms).  All numbers big-endian (XDR).  One frame:
    int 1995 | int natoms | int step | float time | float box[3][3]
    int natoms | then, natoms <= 9:  3 natoms floats
                 else: float precision | int minint[3] | int maxint[3] | int smallidx | int nbytes | nbytes bytes, padded to 4
The compressed block is a bit stream (most significant bit first).  Coordinates are rounded to integers x precision; an
atom is sent either in full - its three integers minus minint as ONE mixed-radix number ((a sx + b) sy + c ... in `bitsize`
= bit length of the product of the three ranges, least significant byte first) - or, inside a "run", as a small
difference to its predecessor (water molecules).  After every full atom comes one flag bit; 1 announces five bits that
(re)define the run length and move the small-number table index.

`write_xtc` emits the subset without runs: every atom in full, flag bit 0 - a valid stream for any reader of the format
(the decoder's run length starts at 0 and only a set flag changes it), about 40 % of the raw size for protein-sized boxes
instead of the ~30 % the run coding reaches.  `read_xtc` implements the whole decoder, runs included (that part can only be
exercised by files written elsewhere).  Coordinates: nanometres in the file, Angstrom at this module's interface, like
the reference's `/ 10` before `md.Trajectory`.
"""
import struct

import numpy as np

MAGIC = 1995
FIRSTIDX = 9
MAGICINTS = [0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512,
             645, 812, 1024, 1290, 1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768,
             41285, 52015, 65536, 82570, 104031, 131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561, 832255,
             1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304, 5284491, 6658042, 8388607, 10568983, 13316085,
             16777216]


def _bit_length_of_product(sizes):
    """sizeofints: the number of bits that hold the product of the ranges."""
    p = 1
    for s in sizes:
        p *= int(s)
    return int(p).bit_length()


def _stream_positions(nbits):
    """Bit index (in the number) of every position of the stream: whole bytes least significant first, each most
    significant bit first, the top nbits % 8 bits last."""
    pos = np.empty(nbits, dtype=np.int64)
    full, rest = divmod(nbits, 8)
    for p in range(nbits):
        b, o = divmod(p, 8)
        pos[p] = 8 * b + (7 - o) if b < full else 8 * b + (rest - 1 - o)
    return pos


def _compress(ints, minint, sizeint):
    """ints [n, 3] -> bytes of the bit stream: every atom in full followed by a 0 flag."""
    n = ints.shape[0]
    rel = (ints - minint).astype(object)
    if max(sizeint) > 0xffffff:
        widths = [int(s).bit_length() for s in sizeint]         # sizeofint per coordinate, sent one after the other
        bits = []
        for k in range(3):
            v = rel[:, k]
            col = np.array([[(int(x) >> (widths[k] - 1 - j)) & 1 for j in range(widths[k])] for x in v], dtype=np.uint8)
            bits.append(col.reshape(n, widths[k]))
        bits.append(np.zeros((n, 1), dtype=np.uint8))
        stream = np.concatenate(bits, axis=1).reshape(-1)
    else:
        nbits = _bit_length_of_product(sizeint)
        value = (rel[:, 0] * int(sizeint[1]) + rel[:, 1]) * int(sizeint[2]) + rel[:, 2]
        pos = _stream_positions(nbits)
        if nbits <= 63:
            v = value.astype(np.uint64)
            bits = ((v[:, None] >> pos[None].astype(np.uint64)) & np.uint64(1)).astype(np.uint8)
        else:
            bits = np.array([[(int(x) >> int(p)) & 1 for p in pos] for x in value], dtype=np.uint8)
        stream = np.concatenate([bits, np.zeros((n, 1), dtype=np.uint8)], axis=1).reshape(-1)
    return np.packbits(stream).tobytes()


def write_xtc(path, xyz, time=None, step=None, box=None, precision=1000.0):
    """xyz [n_frames, n_atoms, 3] in Angstrom -> a .xtc file (nm).  time / step: per frame (default 0, 1, ...); box:
    [n_frames, 3, 3] nm (default zeros, what mdtraj writes for a trajectory without unit cell)."""
    xyz = np.asarray(xyz, dtype=np.float64) / 10.0
    if xyz.ndim == 2:
        xyz = xyz[None]
    n_frames, n_atoms = xyz.shape[:2]
    with open(path, "wb") as f:
        for fr in range(n_frames):
            t = float(fr if time is None else time[fr])
            s = int(fr if step is None else step[fr])
            b = np.zeros((3, 3)) if box is None else np.asarray(box[fr], dtype=np.float64)
            f.write(struct.pack(">iiif", MAGIC, n_atoms, s, t))
            f.write(struct.pack(">9f", *b.reshape(-1)))
            f.write(struct.pack(">i", n_atoms))
            x = xyz[fr].astype(np.float32)
            if n_atoms <= 9:
                f.write(struct.pack(f">{3 * n_atoms}f", *x.reshape(-1)))
                continue
            # xdrfile: lf = x * precision; (int)(lf + 0.5) / (int)(lf - 0.5) - round half away from zero, in float arithmetic
            lf = x * np.float32(precision)
            ints = np.where(lf >= 0, np.floor(lf + np.float32(0.5)), np.ceil(lf - np.float32(0.5))).astype(np.int64)
            if np.abs(ints).max() > 2 ** 31 - 2:
                raise ValueError("coordinates x precision exceed the format's 32-bit integers")
            minint, maxint = ints.min(0), ints.max(0)
            sizeint = [int(maxint[k] - minint[k] + 1) for k in range(3)]
            data = _compress(ints, minint, sizeint)
            f.write(struct.pack(">f3i3ii", float(precision), *[int(v) for v in minint], *[int(v) for v in maxint], FIRSTIDX))
            f.write(struct.pack(">i", len(data)))
            f.write(data + b"\0" * (-len(data) % 4))


class _Bits:
    def __init__(self, data):
        self.bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
        self.p = 0

    def take(self, n):
        v = 0
        for b in self.bits[self.p:self.p + n]:
            v = (v << 1) | int(b)
        self.p += n
        return v

    def ints(self, nbits, sizes):
        """receiveints: `nbits` bits (bytes least significant first) -> the three digits of the mixed-radix number."""
        value, shift = 0, 0
        while nbits > 8:
            value |= self.take(8) << shift
            shift += 8
            nbits -= 8
        if nbits > 0:
            value |= self.take(nbits) << shift
        out = [0, 0, 0]
        for k in (2, 1):
            value, out[k] = divmod(value, int(sizes[k]))
        out[0] = value
        return out


def read_xtc(path):
    """-> (xyz float32 [n_frames, n_atoms, 3] in Angstrom, time [n_frames], step [n_frames], box [n_frames, 3, 3] nm)."""
    frames, times, steps, boxes = [], [], [], []
    with open(path, "rb") as f:
        raw = f.read()
    o = 0
    while o < len(raw):
        magic, n_atoms, step, time = struct.unpack_from(">iiif", raw, o)
        if magic != MAGIC:
            raise ValueError(f"{path}: bad magic number {magic} at byte {o}")
        o += 16
        box = np.array(struct.unpack_from(">9f", raw, o)).reshape(3, 3)
        o += 36
        (size,) = struct.unpack_from(">i", raw, o)
        o += 4
        if size != n_atoms:
            raise ValueError(f"{path}: atom counts of the header ({n_atoms}) and the coordinate block ({size}) differ")
        if size <= 9:
            xyz = np.array(struct.unpack_from(f">{3 * size}f", raw, o), dtype=np.float32).reshape(size, 3)
            o += 12 * size
        else:
            precision, = struct.unpack_from(">f", raw, o)
            minint = struct.unpack_from(">3i", raw, o + 4)
            maxint = struct.unpack_from(">3i", raw, o + 16)
            smallidx, nbytes = struct.unpack_from(">ii", raw, o + 28)
            o += 36
            bits = _Bits(raw[o:o + nbytes])
            o += nbytes + (-nbytes % 4)
            sizeint = [maxint[k] - minint[k] + 1 for k in range(3)]
            big = max(sizeint) > 0xffffff
            widths = [int(s).bit_length() for s in sizeint]
            bitsize = 0 if big else _bit_length_of_product(sizeint)
            smaller = MAGICINTS[max(FIRSTIDX, smallidx - 1)] // 2
            smallnum = MAGICINTS[smallidx] // 2
            sizesmall = [MAGICINTS[smallidx]] * 3
            out = np.empty((size, 3), dtype=np.int64)
            i, run = 0, 0
            while i < size:
                this = [bits.take(w) for w in widths] if big else bits.ints(bitsize, sizeint)
                this = [this[k] + minint[k] for k in range(3)]
                prev = list(this)
                at = i
                i += 1
                is_smaller = 0
                if bits.take(1) == 1:
                    run = bits.take(5)
                    is_smaller = run % 3
                    run -= is_smaller
                    is_smaller -= 1
                if run > 0:
                    for k in range(0, run, 3):
                        d = bits.ints(smallidx, sizesmall)
                        cur = [d[j] + prev[j] - smallnum for j in range(3)]
                        if k == 0:                       # the first small atom and the full one change places (water molecules):
                            out[at] = cur                # it is written first, the full atom second, and the run continues
                            out[at + 1] = prev           # from the small atom
                        else:
                            out[i] = cur
                        prev = cur
                        i += 1
                else:
                    out[at] = this
                smallidx += is_smaller
                if is_smaller < 0:
                    smallnum = smaller
                    smaller = MAGICINTS[smallidx - 1] // 2 if smallidx > FIRSTIDX else 0
                elif is_smaller > 0:
                    smaller = smallnum
                    smallnum = MAGICINTS[smallidx] // 2
                sizesmall = [MAGICINTS[smallidx]] * 3
            xyz = (out / np.float64(precision)).astype(np.float32)
        frames.append(xyz * np.float32(10.0))
        times.append(time)
        steps.append(step)
        boxes.append(box)
    return np.stack(frames), np.array(times, dtype=np.float32), np.array(steps), np.stack(boxes)

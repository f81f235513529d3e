"""numpy writers of the reference's on-disk formats (test fixtures; the C++ readers are host/qadc_io.hpp).
Layouts: vector_io.hpp:153-166 (vecs), convert-quantizer.py:8-40 (.pq.data / .opq.data), and cereal 1.2.2's
binary archive of std::unique_ptr<base_db> as described at the top of host/qadc_io.hpp."""
import struct

import numpy as np

MSB, MSB2 = 0x80000000, 0x40000000


def write_vecs(path, a):
    """a: [count][dim] float32 (.fvecs) / uint8 (.bvecs) / int32 (.ivecs)."""
    a = np.ascontiguousarray(a)
    with open(path, "wb") as f:
        for row in a:
            f.write(struct.pack("<i", a.shape[1]))
            f.write(row.tobytes())


def write_pq_data(path, codebooks, rotation=None):
    m, k, sq_dim = codebooks.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", m * sq_dim, m, int(np.log2(k))))
        f.write(np.ascontiguousarray(codebooks, np.float32).tobytes())
        if rotation is not None:
            f.write(np.ascontiguousarray(rotation, np.float32).tobytes())


class _Ar:
    def __init__(self):
        self.b = bytearray()
        self.ids = {}

    def u32(self, v):
        self.b += struct.pack("<I", v)

    def i32(self, v):
        self.b += struct.pack("<i", v)

    def string(self, s):
        self.b += struct.pack("<Q", len(s)) + s.encode()

    def vec(self, a):
        a = np.ascontiguousarray(a)
        self.b += struct.pack("<Q", a.size) + a.tobytes()

    def poly(self, name):
        if name == "":
            self.u32(MSB2)
        elif name in self.ids:
            self.u32(self.ids[name])
        else:
            self.ids[name] = len(self.ids) + 1
            self.u32(self.ids[name] | MSB)
            self.string(name)
        self.b += b"\x01"

    def pq(self, codebooks, rotation):
        m, k, sq_dim = codebooks.shape
        self.poly("opq" if rotation is not None else "")
        self.i32(m)
        self.i32(int(np.log2(k)))
        self.i32(m * sq_dim)
        self.b += np.ascontiguousarray(codebooks, np.float32).tobytes()
        if rotation is not None:
            self.b += np.ascontiguousarray(rotation, np.float32).tobytes()


def write_flat_db(path, codebooks, codes, rotation=None):
    ar = _Ar()
    ar.poly("flat_db")
    ar.pq(codebooks, rotation)
    ar.u32(codes.shape[0])
    ar.vec(np.ascontiguousarray(codes, np.uint8).reshape(-1))
    open(path, "wb").write(bytes(ar.b))


def write_index_db(path, codebooks, centroids, partitions, labels, rotation=None):
    ar = _Ar()
    ar.poly("index_db")
    ar.i32(len(partitions))
    ar.pq(codebooks, rotation)
    ar.b += np.ascontiguousarray(centroids, np.float32).tobytes()
    for p in partitions:
        ar.vec(np.ascontiguousarray(p, np.uint8).reshape(-1))
    for l in labels:
        ar.vec(np.ascontiguousarray(l, np.uint32))
    open(path, "wb").write(bytes(ar.b))


def pq_encode(codebooks, vecs, rotation=None):
    """Nearest centroid per sub-quantizer (first minimum), even sub-quantizer in the low nibble
    (multiple_set_bits_4, quantizers.hpp:49-68); OPQ rotates first: x @ rotation.T (quantizers.hpp:289-301)."""
    m, k, ds = codebooks.shape
    x = vecs if rotation is None else vecs @ rotation.T
    idx = np.empty((x.shape[0], m), np.uint8)
    for j in range(m):
        d = ((x[:, None, j * ds:(j + 1) * ds] - codebooks[j][None]) ** 2).sum(-1)
        idx[:, j] = d.argmin(1)
    return (idx[:, 0::2] | (idx[:, 1::2] << 4)).astype(np.uint8)

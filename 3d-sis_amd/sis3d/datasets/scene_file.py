"""The `.chunk` / `.scene` binary container (writer: datagen/SceneSampler/main.cpp:348-415; reader:
lib/datasets/dataset.py:45-153 over lib/datasets/BinaryReader.py).  Little-endian, no padding:

    u64 dims[3]                      X, Y, Z
    f32 sdf[X*Y*Z]                   x fastest, then y, then z
    u32 nbox;   nbox  x { f32 min[3]; f32 max[3]; u32 label }
    u32 nmask;  nmask x { u32 label; u64 dims[3]; u16 data[dx*dy*dz] }      (x fastest)
    u32 nstat;  nstat x f32 part_in_volume
    f32 world2chunk[16]              column-major 4x4
    u32 nimg;   nimg x u32 frame id
Sections after the boxes are optional on the read side (the reference reads them depending on cfg).

The reference unpacks every float through `struct.unpack` into a Python tuple (442 k objects per chunk, millions per
scene); here the file is mapped once and each section is a zero-copy numpy view, so the sdf can go to the GPU as is
(`ops.tsdf_encode` does the TSDF encoding and the layout change there).
"""
import numpy as np

U32 = np.dtype("<u4")
U64 = np.dtype("<u8")
F32 = np.dtype("<f4")
U16 = np.dtype("<u2")


class SceneFileError(IOError):
    """the reference raises BinaryReaderEOFException ('Not enough bytes in file to satisfy read request')"""


class _Cursor(object):
    def __init__(self, buf):
        self.buf = buf
        self.pos = 0

    def take(self, dtype, count=1):
        nbytes = dtype.itemsize * int(count)
        if self.pos + nbytes > len(self.buf):
            raise SceneFileError("Not enough bytes in file to satisfy read request")
        a = np.frombuffer(self.buf, dtype=dtype, count=int(count), offset=self.pos)
        self.pos += nbytes
        return a

    def left(self):
        return len(self.buf) - self.pos


class SceneFile(object):
    """Parsed container.  Attributes: dims (X,Y,Z); sdf (flat f32 view, file order); boxes (n,6) f32; box_labels (n,) u32;
    masks [(label, (dx,dy,dz) uint16 F-order view)]; part_in_volume (m,) f32 or None; world2chunk (4,4) f32 row-major
    view of the column-major block (i.e. already the matrix the writer stored) or None; frame_ids (k,) u32 or None."""

    def __init__(self, path, want_masks=True, want_stats=True, want_images=True):
        self.path = path
        buf = np.memmap(path, dtype=np.uint8, mode="r")
        c = _Cursor(buf)
        self.dims = tuple(int(v) for v in c.take(U64, 3))
        n = self.dims[0] * self.dims[1] * self.dims[2]
        self.sdf = c.take(F32, n)
        (nbox,) = c.take(U32)
        rec = np.dtype([("box", F32, 6), ("label", U32)])
        boxes = c.take(rec, nbox)
        self.boxes = boxes["box"].reshape(-1, 6)
        self.box_labels = boxes["label"]
        self.masks = []
        self.part_in_volume = self.world2chunk = self.frame_ids = None
        if want_masks:
            (nmask,) = c.take(U32)
            for _ in range(int(nmask)):
                (label,) = c.take(U32)
                md = tuple(int(v) for v in c.take(U64, 3))
                data = c.take(U16, md[0] * md[1] * md[2]).reshape(md, order="F")
                self.masks.append((int(label), data))
            if want_stats:
                (nstat,) = c.take(U32)
                self.part_in_volume = c.take(F32, nstat)
                if want_images:
                    self.world2chunk = c.take(F32, 16).reshape(4, 4, order="F")
                    (nimg,) = c.take(U32)
                    self.frame_ids = c.take(U32, nimg)
        self.bytes_read = c.pos

    def sdf_grid(self):
        """(X,Y,Z) view of the sdf (Fortran order, no copy)"""
        return self.sdf.reshape(self.dims, order="F")


def write_scene_file(path, sdf, boxes=(), labels=(), masks=(), part_in_volume=None, world2chunk=None, frame_ids=None):
    """Writes the container (tests / synthetic data).  sdf: (X,Y,Z) array; masks: [(label, (dx,dy,dz) array)]."""
    sdf = np.asarray(sdf, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(np.asarray(sdf.shape, dtype=U64).tobytes())
        f.write(sdf.astype(F32).tobytes(order="F"))
        f.write(np.asarray([len(boxes)], dtype=U32).tobytes())
        for b, lab in zip(boxes, labels):
            f.write(np.asarray(b, dtype=F32).tobytes())
            f.write(np.asarray([lab], dtype=U32).tobytes())
        if masks is None:
            return
        f.write(np.asarray([len(masks)], dtype=U32).tobytes())
        for lab, m in masks:
            m = np.asarray(m)
            f.write(np.asarray([lab], dtype=U32).tobytes())
            f.write(np.asarray(m.shape, dtype=U64).tobytes())
            f.write(m.astype(U16).tobytes(order="F"))
        if part_in_volume is None:
            return
        f.write(np.asarray([len(part_in_volume)], dtype=U32).tobytes())
        f.write(np.asarray(part_in_volume, dtype=F32).tobytes())
        if world2chunk is None:
            return
        f.write(np.asarray(world2chunk, dtype=F32).tobytes(order="F"))
        frame_ids = [] if frame_ids is None else frame_ids
        f.write(np.asarray([len(frame_ids)], dtype=U32).tobytes())
        f.write(np.asarray(frame_ids, dtype=U32).tobytes())

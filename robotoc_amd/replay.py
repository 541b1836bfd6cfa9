"""Stage dump / replay files (include/rtoc.h: rtoc_dump_header) in numpy.

The same bytes `rtoc_save_stage_dump` writes and `rtoc_load_stage_dump` reads: stage data recorded at
the evalKKT boundary of a robotoc run ("Recording real stage data") can be inspected,
generated or replayed without the C library.
"""
import ctypes as C
import struct

import numpy as np

from .types import BoxRow, Dims, Grid

MAGIC = b"RTOCDMP1"
NUM_SLOTS = 16
_HDR = struct.Struct("<8sII6i8i16Q")  # magic, version, header_bytes, dims, nstages..reserved, counts
assert _HDR.size == 8 + 8 + 24 + 32 + 128


def write_dump(path, dims, grids, batch, buffers, rows=(), cone_contacts=0, cone_dim=0, cone_rows=0):
    """buffers: {RTOC_BUF_* index: float64 array of [batch][nstages][stride] (or [batch][n])}."""
    counts = [0] * NUM_SLOTS
    for b, arr in buffers.items():
        counts[b] = int(np.asarray(arr).size)
    hdr = _HDR.pack(MAGIC, 1, _HDR.size, dims.nv, dims.nu, dims.np, dims.nf_max, dims.ns_max, dims.nc_max,
                    len(grids), batch, len(rows), cone_contacts, cone_dim, cone_rows, 0, 0, *counts)
    with open(path, "wb") as f:
        f.write(hdr)
        for g in grids:
            f.write(bytes(g))
        for r in rows:
            f.write(bytes(r))
        for b in sorted(buffers):
            f.write(np.ascontiguousarray(buffers[b], dtype=np.float64).tobytes())


def read_dump(path):
    """-> dict(dims, grids, batch, rows, cone_contacts, cone_dim, cone_rows (0|5 friction, 17 wrench), buffers={index: flat float64 array})."""
    with open(path, "rb") as f:
        raw = f.read()
    (magic, version, hbytes, nv, nu, npas, nf, ns, nc, nstages, batch, nrows, cc, cd, crows, _r1, _r2,
     *counts) = _HDR.unpack_from(raw, 0)
    if magic != MAGIC or version != 1 or hbytes != _HDR.size:
        raise ValueError("not a stage dump: %r" % path)
    off = hbytes
    grids = []
    for _ in range(nstages):
        grids.append(Grid.from_buffer_copy(raw, off))
        off += C.sizeof(Grid)
    rows = []
    for _ in range(nrows):
        rows.append(BoxRow.from_buffer_copy(raw, off))
        off += C.sizeof(BoxRow)
    buffers = {}
    for b, n in enumerate(counts):
        if n:
            buffers[b] = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy()
            off += 8 * n
    if off != len(raw):
        raise ValueError("trailing or missing bytes in %r" % path)
    return dict(dims=Dims(nv, nu, npas, nf, ns, nc), grids=grids, batch=batch, rows=rows, cone_contacts=cc,
                cone_dim=cd, cone_rows=crows, buffers=buffers)

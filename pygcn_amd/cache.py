"""Binary cache file of a prepared adjacency: CSR(Â) + its launch schedule (+ CSR(Âᵀ) and its
schedule), so that a graph ingested once (reference recipe pygcn/utils.py:356-368, or the synthetic
generators) is re-opened without re-sorting, re-transposing or re-planning.  SURVEY §8 row f4.

Layout (little-endian, version 1):

    offset 0   8 bytes   magic  b"PYGCNCSR"
           8   u32       format version (1)
          12   u32       length H of the JSON header
          16   H bytes   JSON header (utf-8): {"meta": {...}, "sections": [{"name", "dtype",
                         "count", "offset", "crc32"}, ...]}; offsets are absolute and 64-byte
                         aligned
          ...            the raw arrays

Pure host code (numpy + zlib), no device involved: `CSRGraph.save` / `CSRGraph.load`
(pygcn_amd/graph.py) move the arrays between HBM and this file.  Readers reject a wrong magic, an
unknown version, a truncated file and (with verify=True) any section whose CRC-32 does not match.
"""
import json
import struct
import zlib

import numpy as np

MAGIC = b"PYGCNCSR"
VERSION = 1
_ALIGN = 64
_DTYPES = {"int32": np.int32, "int64": np.int64, "float32": np.float32}


class CacheFormatError(RuntimeError):
    pass


def _crc(a):
    crc, mv, step = 0, memoryview(a).cast("B"), 1 << 26
    for i in range(0, len(mv), step):
        crc = zlib.crc32(mv[i:i + step], crc)
    return crc & 0xFFFFFFFF


def write_file(path, meta, arrays):
    """`arrays`: ordered mapping name -> 1-D numpy array (int32 / int64 / float32)."""
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    for k, v in arrays.items():
        if v.dtype.name not in _DTYPES or v.ndim != 1:
            raise CacheFormatError(f"section {k}: unsupported array {v.dtype} {v.shape}")
    # two passes: the header's own length moves the offsets, so size it with placeholder offsets
    sections = [{"name": k, "dtype": v.dtype.name, "count": int(v.size), "offset": 0,
                 "crc32": _crc(v)} for k, v in arrays.items()]

    def header_bytes():
        return json.dumps({"meta": meta, "sections": sections}, sort_keys=True).encode()
    for _ in range(3):                                   # offsets have a fixed point after <= 2 rounds
        off = 16 + len(header_bytes())
        for s, v in zip(sections, arrays.values()):
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            s["offset"] = off
            off += v.nbytes
    hdr = header_bytes()
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", VERSION, len(hdr)) + hdr)
        for s, v in zip(sections, arrays.values()):
            f.write(b"\0" * (s["offset"] - f.tell()))
            v.astype(v.dtype.newbyteorder("<"), copy=False).tofile(f)
    return off


def read_file(path, verify=True, mmap=True):
    """-> (meta dict, {name: numpy array}).  Arrays are memory-mapped read-only when `mmap`."""
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise CacheFormatError(f"{path}: not a pygcn_amd CSR cache file (bad magic)")
        version, hlen = struct.unpack("<II", head[8:16])
        if version != VERSION:
            raise CacheFormatError(f"{path}: format version {version}, this reader knows {VERSION}")
        raw = f.read(hlen)
        if len(raw) != hlen:
            raise CacheFormatError(f"{path}: truncated header")
        try:
            hdr = json.loads(raw.decode())
            meta, sections = hdr["meta"], hdr["sections"]
        except (ValueError, KeyError) as ex:
            raise CacheFormatError(f"{path}: malformed header: {ex}") from ex
        f.seek(0, 2)
        size = f.tell()
    out = {}
    for s in sections:
        if s["dtype"] not in _DTYPES:
            raise CacheFormatError(f"{path}: section {s['name']}: unknown dtype {s['dtype']}")
        dt = np.dtype(_DTYPES[s["dtype"]]).newbyteorder("<")
        nbytes = s["count"] * dt.itemsize
        if s["offset"] % _ALIGN or s["offset"] + nbytes > size:
            raise CacheFormatError(f"{path}: section {s['name']} lies outside the file (truncated?)")
        if s["count"] == 0:
            a = np.empty(0, dt)
        elif mmap:
            a = np.memmap(path, dtype=dt, mode="r", offset=s["offset"], shape=(s["count"],))
        else:
            a = np.fromfile(path, dtype=dt, count=s["count"], offset=s["offset"])
        if verify and _crc(a) != s["crc32"]:
            raise CacheFormatError(f"{path}: section {s['name']} fails its CRC-32 (corrupt file)")
        out[s["name"]] = a
    return meta, out

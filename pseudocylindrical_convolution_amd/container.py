"""Bitstream container of the codec files.

The reference writes the raw arithmetic-coder bytes and nothing else
(coder/coder.h:20-24), so a file cannot be decoded without knowing the image
size, the model and its valid_dim out of band (pseudo_codec.py:206,209,229-234
hard-code them).  This container puts a 16-byte header in front of the same
payload -- the payload itself is byte for byte the reference-format stream:

    offset  size  field
    0       4     magic  b"PCVC"
    4       1     version (1)
    5       1     flags: bit 0 = model from the VSSIM list (--ssim), else VMSE
    6       1     model index inside its list
    7       1     ngroup = valid_dim / 4   (14, 28 or 48)
    8       2     ERP height / 16, little endian
    10      2     ERP width / 16, little endian
    12      4     payload length in bytes, little endian

The command line writes the reference's headerless files unless `--container` is given;
decoding recognises a container by `sniff` (magic, version and a payload length that
matches the file), so headerless files produced by the reference decode as before.
"""
import os
import struct

MAGIC = b"PCVC"
VERSION = 1
HEADER_BYTES = 16
_FMT = "<4sBBBBHHI"


class ContainerError(ValueError):
    pass


def pack(payload, height, width, model_idx, ssim, valid_dim):
    """header + payload"""
    if height % 16 or width % 16 or not (0 < height // 16 < 65536 and 0 < width // 16 < 65536):
        raise ContainerError("ERP size %dx%d does not fit the header (multiples of 16, < 2^20)" % (width, height))
    if valid_dim % 4 or not 0 < valid_dim // 4 < 256:
        raise ContainerError("valid_dim %d does not fit the header" % valid_dim)
    if not 0 <= model_idx < 256:
        raise ContainerError("model index %d does not fit the header" % model_idx)
    if len(payload) >= 1 << 32:
        raise ContainerError("payload too long")
    head = struct.pack(_FMT, MAGIC, VERSION, 1 if ssim else 0, model_idx, valid_dim // 4, height // 16, width // 16,
                       len(payload))
    return head + bytes(payload)


def unpack(data):
    """-> (dict(height, width, model_idx, ssim, valid_dim), payload bytes)"""
    if len(data) < HEADER_BYTES:
        raise ContainerError("file shorter than the %d-byte header" % HEADER_BYTES)
    magic, version, flags, model_idx, ngroup, h16, w16, n = struct.unpack(_FMT, data[:HEADER_BYTES])
    if magic != MAGIC:
        raise ContainerError("no container magic: a headerless reference-format stream?")
    if version != VERSION:
        raise ContainerError("container version %d, this build reads %d" % (version, VERSION))
    if len(data) - HEADER_BYTES != n:
        raise ContainerError("payload is %d bytes, header says %d" % (len(data) - HEADER_BYTES, n))
    if not (h16 and w16 and ngroup):
        raise ContainerError("empty field in the header")
    return ({"height": h16 * 16, "width": w16 * 16, "model_idx": model_idx, "ssim": bool(flags & 1),
             "valid_dim": ngroup * 4}, bytes(data[HEADER_BYTES:]))


def sniff(path):
    """header dict when the file is a well-formed container, else None (a headerless stream).
    A raw arithmetic-coded stream passes for a container only if its first 16 bytes happen to
    spell the magic, the version and its own length: 2^-72 for random bytes."""
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if len(head) < HEADER_BYTES or head[:4] != MAGIC:
            return None
        magic, version, flags, model_idx, ngroup, h16, w16, n = struct.unpack(_FMT, head)
        if version != VERSION or n != size - HEADER_BYTES or not (h16 and w16 and ngroup):
            return None
        return {"height": h16 * 16, "width": w16 * 16, "model_idx": model_idx, "ssim": bool(flags & 1),
                "valid_dim": ngroup * 4}
    except OSError:
        return None


def read(path):
    with open(path, "rb") as f:
        return unpack(f.read())


def write(path, payload, **fields):
    with open(path, "wb") as f:
        f.write(pack(payload, **fields))

"""Image files for the two buffers of a frame: PNG for the presented canvas (RGBA8), OpenEXR for the float render
target before tone reproduction. Both written with the standard library only (zlib, struct); the EXR is a
single-part scanline file with three 32-bit float channels and no compression, which every EXR reader accepts."""
import struct
import zlib

import numpy as np


def write_png(path, rgba_bottom_up):
    """rgba_bottom_up: (H, W, 3 or 4) uint8, first row = bottom row (gl.readPixels order, what present() returns)."""
    h, w, c = rgba_bottom_up.shape
    rows = np.ascontiguousarray(rgba_bottom_up[::-1, :, :3])    # PNG rows run top-down
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


EXR_MAGIC = 20000630


def _attr(name, kind, data):
    return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(data)) + data


def write_exr(path, rgb_bottom_up):
    """rgb_bottom_up: (H, W, 3) float32, first row = bottom row. Channels are stored in alphabetical order (B, G, R),
    scan lines top-down (increasing y), pixel type FLOAT, compression NONE."""
    img = np.ascontiguousarray(rgb_bottom_up[::-1], np.float32)
    h, w, c = img.shape
    assert c == 3
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    window = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", b"\0") +
              _attr("dataWindow", "box2i", window) + _attr("displayWindow", "box2i", window) +
              _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    start = 8 + len(header) + 8 * h
    line_bytes = 8 + 12 * w
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", EXR_MAGIC, 2) + header)
        f.write(struct.pack("<%dQ" % h, *(start + y * line_bytes for y in range(h))))
        for y in range(h):
            f.write(struct.pack("<ii", y, 12 * w))
            f.write(img[y, :, 2].tobytes() + img[y, :, 1].tobytes() + img[y, :, 0].tobytes())


def read_exr(path):
    """Reads back what write_exr wrote (uncompressed scan lines, float channels): (H, W, 3) float32 RGB, bottom-up."""
    with open(path, "rb") as f:
        b = f.read()
    magic, version = struct.unpack_from("<ii", b, 0)
    if magic != EXR_MAGIC or (version & 0xff) != 2 or (version & ~0xff):
        raise ValueError("not a single-part scan-line OpenEXR file")
    pos, attrs = 8, {}
    while b[pos] != 0:
        end = b.index(b"\0", pos)
        name = b[pos:end].decode()
        pos = end + 1
        end = b.index(b"\0", pos)
        kind = b[pos:end].decode()
        pos = end + 1
        size, = struct.unpack_from("<i", b, pos)
        attrs[name] = (kind, b[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("compressed EXR")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, p, ch = [], 0, attrs["channels"][1]
    while ch[p] != 0:
        end = ch.index(b"\0", p)
        names.append(ch[p:end].decode())
        if struct.unpack_from("<i", ch, end + 1)[0] != 2:
            raise ValueError("channel %s is not FLOAT" % names[-1])
        p = end + 1 + 16
    offsets = struct.unpack_from("<%dQ" % h, b, pos)
    planes = {n: np.empty((h, w), np.float32) for n in names}
    for off in offsets:
        y, size = struct.unpack_from("<ii", b, off)
        row = np.frombuffer(b, np.float32, w * len(names), off + 8).reshape(len(names), w)
        for k, n in enumerate(names):
            planes[n][y - y0] = row[k]
    return np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)[::-1]

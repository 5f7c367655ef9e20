'use strict';
// RGBA8 -> PNG with node's zlib (colour type 6, 8 bits, filter 0 on every row), and the reverse for files this module
// wrote. The browser front end (examples/serve.js) streams the canvas as PNG frames.
const zlib = require('zlib');

const CRC_TABLE = (() => {
    const t = new Uint32Array(256);
    for (let n = 0; n < 256; n++) {
        let c = n;
        for (let k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320 ^ (c >>> 1)) : (c >>> 1);
        t[n] = c >>> 0;
    }
    return t;
})();

function crc32(buf) {
    let c = 0xffffffff;
    for (let i = 0; i < buf.length; i++) c = CRC_TABLE[(c ^ buf[i]) & 0xff] ^ (c >>> 8);
    return (c ^ 0xffffffff) >>> 0;
}

function chunk(type, data) {
    const head = Buffer.alloc(8), tail = Buffer.alloc(4);
    head.writeUInt32BE(data.length, 0);
    head.write(type, 4, 'latin1');
    tail.writeUInt32BE(crc32(Buffer.concat([head.slice(4), data])), 0);
    return Buffer.concat([head, data, tail]);
}

// pixels: width * height * 4 bytes; bottomUp: the first row of `pixels` is the bottom row of the image (as readPixels
// and RayTracer.render() deliver it). level: zlib level (1 = fastest, what a stream wants).
function encodePNG(pixels, width, height, bottomUp, level) {
    const stride = width * 4, raw = Buffer.alloc((stride + 1) * height);
    const src = Buffer.from(pixels.buffer, pixels.byteOffset, pixels.byteLength);
    for (let y = 0; y < height; y++) {
        const from = (bottomUp ? height - 1 - y : y) * stride;
        raw[y * (stride + 1)] = 0;
        src.copy(raw, y * (stride + 1) + 1, from, from + stride);
    }
    const ihdr = Buffer.alloc(13);
    ihdr.writeUInt32BE(width, 0);
    ihdr.writeUInt32BE(height, 4);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    return Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), chunk('IHDR', ihdr),
        chunk('IDAT', zlib.deflateSync(raw, { level: level === undefined ? 1 : level })), chunk('IEND', Buffer.alloc(0))]);
}

// the reverse of encodePNG (filter 0 rows only): { width, height, pixels } with rows top-down
function decodePNG(buf) {
    if (buf.readUInt32BE(0) !== 0x89504e47) throw new Error('decodePNG: not a PNG');
    let pos = 8, width = 0, height = 0;
    const idat = [];
    while (pos < buf.length) {
        const len = buf.readUInt32BE(pos), type = buf.toString('latin1', pos + 4, pos + 8), data = buf.slice(pos + 8, pos + 8 + len);
        if (buf.readUInt32BE(pos + 8 + len) !== crc32(buf.slice(pos + 4, pos + 8 + len))) throw new Error('decodePNG: bad CRC in ' + type);
        if (type === 'IHDR') {
            width = data.readUInt32BE(0); height = data.readUInt32BE(4);
            if (data[8] !== 8 || data[9] !== 6) throw new Error('decodePNG: only 8-bit RGBA');
        } else if (type === 'IDAT') idat.push(data);
        pos += 12 + len;
    }
    const raw = zlib.inflateSync(Buffer.concat(idat)), stride = width * 4, pixels = new Uint8Array(stride * height);
    for (let y = 0; y < height; y++) {
        if (raw[y * (stride + 1)] !== 0) throw new Error('decodePNG: filtered rows are not supported');
        pixels.set(raw.slice(y * (stride + 1) + 1, (y + 1) * (stride + 1)), y * stride);
    }
    return { width, height, pixels };
}

module.exports = { encodePNG, decodePNG, crc32 };

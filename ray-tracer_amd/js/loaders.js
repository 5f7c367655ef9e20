'use strict';
// loadPLY / loadOBJ with the reference's contract (js/8-kd_tree.js:2851-2921): a Promise that resolves to
// { verts: [THREE.Vector3], faces: [Triangle] }. The numbers are what THREE.PLYLoader / THREE.OBJLoader followed
// by Geometry.fromBufferGeometry produce: positions pass through a Float32 attribute (so every coordinate is a
// double holding a single-precision value), vertex and face order are the file's, PLY quads split 0-1-3 / 1-2-3
// (js/loaders/PLYLoader.js:309-347), OBJ geometry is de-indexed so face k uses vertices 3k, 3k+1, 3k+2
// (js/loaders/OBJLoader.js:653-655).
const fs = require('fs');

const PLY_TYPES = {
    int8: ['getInt8', 1], char: ['getInt8', 1], uint8: ['getUint8', 1], uchar: ['getUint8', 1],
    int16: ['getInt16', 2], short: ['getInt16', 2], uint16: ['getUint16', 2], ushort: ['getUint16', 2],
    int32: ['getInt32', 4], int: ['getInt32', 4], uint32: ['getUint32', 4], uint: ['getUint32', 4],
    float32: ['getFloat32', 4], float: ['getFloat32', 4], float64: ['getFloat64', 8], double: ['getFloat64', 8],
};

// data: a Buffer with the whole file (ASCII, binary_little_endian or binary_big_endian; js/loaders/PLYLoader.js:252-486)
function parsePLY(data) {
    if (typeof data === 'string') data = Buffer.from(data, 'latin1');
    const probe = data.toString('latin1', 0, Math.min(data.length, 1 << 16));
    const hm = /ply([\s\S]*?)end_header\s/.exec(probe);
    if (!hm) throw new Error('loadPLY: no end_header');
    const bodyStart = hm.index + hm[0].length;
    let format = 'ascii';
    const elements = [];
    for (const line of hm[1].split('\n')) {
        const tok = line.trim().split(/\s+/);
        if (tok[0] === 'format') format = tok[1];
        else if (tok[0] === 'element') elements.push({ name: tok[1], count: parseInt(tok[2], 10), props: [] });
        else if (tok[0] === 'property') {
            elements[elements.length - 1].props.push(tok[1] === 'list'
                ? { list: true, countType: tok[2], itemType: tok[3], name: tok[4] } : { list: false, type: tok[1], name: tok[2] });
        }
    }
    const positions = [], indices = [];
    const addFace = (ix) => {
        if (ix.length === 3) indices.push(ix[0], ix[1], ix[2]);
        else if (ix.length === 4) indices.push(ix[0], ix[1], ix[3], ix[1], ix[2], ix[3]);
    };
    if (format === 'ascii') {
        const text = data.toString('latin1', bodyStart);
        let pos = 0;
        const nextLine = () => {
            for (;;) {
                let end = text.indexOf('\n', pos);
                if (end < 0) end = text.length;
                const line = text.slice(pos, end).trim();
                pos = end + 1;
                if (line !== '' || pos > text.length) return line;
            }
        };
        for (const el of elements) {
            const names = el.props.map((p) => p.name);
            if (el.name === 'vertex') {
                const ix = names.indexOf('x'), iy = names.indexOf('y'), iz = names.indexOf('z');
                for (let i = 0; i < el.count; i++) {
                    const t = nextLine().split(/\s+/);
                    positions.push(parseFloat(t[ix]), parseFloat(t[iy]), parseFloat(t[iz]));
                }
            } else if (el.name === 'face') {
                for (let i = 0; i < el.count; i++) {
                    const t = nextLine().split(/\s+/).map((s) => parseInt(s, 10));
                    addFace(t.slice(1, 1 + t[0]));
                }
            } else {
                for (let i = 0; i < el.count; i++) nextLine();
            }
        }
    } else if (format === 'binary_little_endian' || format === 'binary_big_endian') {
        const little = format === 'binary_little_endian';
        const view = new DataView(data.buffer, data.byteOffset + bodyStart, data.length - bodyStart);
        let at = 0;
        const read = (type) => {
            const t = PLY_TYPES[type];
            if (!t) throw new Error('loadPLY: unknown property type ' + type);
            const v = view[t[0]](at, little);
            at += t[1];
            return v;
        };
        for (const el of elements) {
            for (let i = 0; i < el.count; i++) {
                const rec = {};
                for (const p of el.props) {
                    if (p.list) {
                        const n = read(p.countType), list = [];
                        for (let k = 0; k < n; k++) list.push(read(p.itemType));
                        rec[p.name] = list;
                    } else rec[p.name] = read(p.type);
                }
                if (el.name === 'vertex') positions.push(rec.x, rec.y, rec.z);
                else if (el.name === 'face') addFace(rec.vertex_indices || rec.vertex_index);
            }
        }
    } else throw new Error('loadPLY: format ' + format);
    return { positions: new Float32Array(positions), indices };
}

function parseOBJ(text) {
    const pts = [], out = [];
    const corner = (i) => { out.push(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]); };
    for (const raw of text.split('\n')) {
        const t = raw.trim().split(/\s+/);
        if (t[0] === 'v') pts.push(parseFloat(t[1]), parseFloat(t[2]), parseFloat(t[3]));
        else if (t[0] === 'f') {
            const idx = t.slice(1).map((s) => { const i = parseInt(s.split('/')[0], 10); return i > 0 ? i - 1 : pts.length / 3 + i; });
            if (idx.length === 3) { corner(idx[0]); corner(idx[1]); corner(idx[2]); }
            else if (idx.length === 4) { corner(idx[0]); corner(idx[1]); corner(idx[3]); corner(idx[1]); corner(idx[2]); corner(idx[3]); }
            else throw new Error('loadOBJ: faces with more than 4 vertices are not supported');
        }
    }
    const indices = [];
    for (let i = 0; i < out.length / 3; i++) indices.push(i);
    return { positions: new Float32Array(out), indices };
}

function makeLoaders(G, THREE, Triangle) {
    const resolvePath = (filename) => (G.resolveAsset ? G.resolveAsset(filename) : filename);
    function toMesh(parsed) {
        const verts = [], faces = [];
        const p = parsed.positions;
        for (let i = 0; i < p.length; i += 3) verts.push(new THREE.Vector3(p[i], p[i + 1], p[i + 2]));
        const ix = parsed.indices;
        for (let i = 0; i < ix.length; i += 3) faces.push(new Triangle(ix[i], ix[i + 1], ix[i + 2]));
        return { verts, faces };
    }
    function loadPLY(filename) {
        return new Promise((resolve, reject) => {
            fs.readFile(resolvePath(filename), (err, data) => {
                if (err) return reject(err);
                try { resolve(toMesh(parsePLY(data))); } catch (e) { reject(e); }
            });
        });
    }
    function loadOBJ(filename) {
        return new Promise((resolve, reject) => {
            fs.readFile(resolvePath(filename), 'utf8', (err, text) => {
                if (err) return reject(err);
                try { resolve(toMesh(parseOBJ(text))); } catch (e) { reject(e); }
            });
        });
    }
    return { loadPLY, loadOBJ };
}

module.exports = { makeLoaders, parsePLY, parseOBJ };

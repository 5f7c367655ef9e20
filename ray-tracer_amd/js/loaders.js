'use strict';
// loadPLY / loadOBJ with the reference's contract (js/8-kd_tree.js:2851-2921): a Promise that resolves to
// { verts: [THREE.Vector3], faces: [Triangle] }. The numbers are what THREE.PLYLoader / THREE.OBJLoader followed
// by Geometry.fromBufferGeometry produce: positions pass through a Float32 attribute (so every coordinate is a
// double holding a single-precision value), vertex and face order are the file's, PLY quads split 0-1-3 / 1-2-3
// (js/loaders/PLYLoader.js:309-347), OBJ geometry is de-indexed so face k uses vertices 3k, 3k+1, 3k+2
// (js/loaders/OBJLoader.js:653-655).
const fs = require('fs');

function parsePLY(text) {
    const cut = text.search(/end_header\s/);
    if (cut < 0) throw new Error('loadPLY: no end_header');
    const head = text.slice(0, cut).split('\n');
    const bodyStart = text.indexOf('\n', cut) + 1;
    const elements = [];
    for (const line of head) {
        const tok = line.trim().split(/\s+/);
        if (tok[0] === 'format' && tok[1] !== 'ascii') throw new Error('loadPLY: only ASCII PLY is supported');
        if (tok[0] === 'element') elements.push({ name: tok[1], count: parseInt(tok[2], 10), props: [] });
        else if (tok[0] === 'property') elements[elements.length - 1].props.push(tok[tok.length - 1]);
    }
    const positions = [], indices = [];
    let pos = bodyStart;
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
        if (el.name === 'vertex') {
            const ix = el.props.indexOf('x'), iy = el.props.indexOf('y'), iz = el.props.indexOf('z');
            for (let i = 0; i < el.count; i++) {
                const t = nextLine().split(/\s+/);
                positions.push(parseFloat(t[ix]), parseFloat(t[iy]), parseFloat(t[iz]));
            }
        } else if (el.name === 'face') {
            for (let i = 0; i < el.count; i++) {
                const t = nextLine().split(/\s+/).map((s) => parseInt(s, 10));
                if (t[0] === 3) indices.push(t[1], t[2], t[3]);
                else if (t[0] === 4) indices.push(t[1], t[2], t[4], t[2], t[3], t[4]);
            }
        } else {
            for (let i = 0; i < el.count; i++) nextLine();
        }
    }
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
            fs.readFile(resolvePath(filename), 'utf8', (err, text) => {
                if (err) return reject(err);
                try { resolve(toMesh(parsePLY(text))); } catch (e) { reject(e); }
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

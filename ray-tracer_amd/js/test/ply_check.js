'use strict';
// Parses PLY files with loaders.js and prints hashes of the positions (float32) and face indices; with a second
// argument - the root of a checkout of the reference - the same files also go through the reference's own
// THREE.PLYLoader (three.min.js + js/loaders/PLYLoader.js evaluated in a vm context) and must give the same arrays.
//   node ply_check.js <file.ply>[,<file.ply>...] [reference root]
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const crypto = require('crypto');
const { parsePLY } = require('../loaders.js');

const files = process.argv[2].split(',');
const refRoot = process.argv[3];
const sha = (ta) => crypto.createHash('sha256').update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest('hex');

let refParse = null;
if (refRoot) {
    const sandbox = { console, ArrayBuffer, Float32Array, Float64Array, Uint8Array, Uint16Array, Uint32Array, Int8Array, Int16Array, Int32Array, DataView };
    sandbox.window = sandbox;
    sandbox.self = sandbox;
    vm.createContext(sandbox);
    const run = (rel, edit) => { let src = fs.readFileSync(path.join(refRoot, rel), 'utf8'); if (edit) src = edit(src); vm.runInContext(src, sandbox, { filename: rel }); };
    run('js/three.min.js', (s) => 'var module=undefined, exports=undefined, define=undefined;\n' + s);
    run('js/loaders/PLYLoader.js');
    refParse = (buf) => {
        sandbox.__data = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
        return vm.runInContext(`(function () {
            var g = new THREE.PLYLoader().parse(__data);
            return { positions: g.attributes.position.array, indices: g.index ? Array.from(g.index.array) : [] };
        })()`, sandbox);
    };
}

const out = [];
for (const f of files) {
    const buf = fs.readFileSync(f);
    const mine = parsePLY(buf);
    const rec = { file: path.basename(f), n_verts: mine.positions.length / 3, n_faces: mine.indices.length / 3,
        positions_sha256: sha(mine.positions), indices_sha256: sha(new Uint32Array(mine.indices)) };
    if (refParse) {
        const ref = refParse(buf);
        rec.reference_equal = sha(new Float32Array(ref.positions)) === rec.positions_sha256 &&
            sha(new Uint32Array(ref.indices)) === rec.indices_sha256;
    }
    out.push(rec);
}
console.log(JSON.stringify(out));

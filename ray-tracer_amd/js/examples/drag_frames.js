'use strict';
// The mesh moved a little on every frame (Mesh.translate, what the reference's mouse drag calls, js/8-kd_tree.js:2055-2064,
// :2771) while RayTracer.update follows the reference's rule for the tree (js/8-kd_tree.js:2494-2522): vertices on every
// frame, tree and root box only when the rebuild interval has passed. Prints one JSON line: per frame its time, the hash
// of the RGBA8 render target and of the vertex array, and the k-d root box in use.
//   node drag_frames.js --times 1,1.0166,1.0333,1.05,1.31 [--step dx,dy,dz] [--width W --height H] [--depth D]
//                       [--model models/bun_zipper_res4.ply] [--mesh-material ...] [--kd-builder host|device] [--interval 0.30]
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const rt = require('..');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const opts = {
    WIDTH: parseInt(arg('width', '320'), 10), HEIGHT: parseInt(arg('height', '240'), 10),
    MAX_RECURSION_DEPTH: parseInt(arg('depth', '3'), 10), MODEL_FILE_NAME: arg('model', 'models/bun_zipper_res4.ply'),
    KD_BUILDER: arg('kd-builder', 'host'), KD_REBUILD_INTERVAL: parseFloat(arg('interval', '0.30')),
};
const mm = arg('mesh-material', null);
if (mm) opts.MESH_MATERIAL = mm.split(',').map(parseFloat);
const G = rt.createContext(opts);
const sceneFile = path.join(__dirname, 'default_scene.js');
rt.run(G, fs.readFileSync(sceneFile, 'utf8'), sceneFile);
const sha = (a) => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');

(async () => {
    const data = await G.loadPLY(G.MODEL_FILE_NAME);
    const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
    const mesh = scene.objects[scene.objects.length - 1];
    const tracer = new G.RayTracer();
    tracer.setScene(scene);
    const times = arg('times', '1,1.0166666666666666,1.0333333333333334,1.05,1.31').split(',').map(Number);
    const step = arg('step', '1.5,0.75,-1').split(',').map(Number);
    const frames = [];
    times.forEach((t, k) => {
        if (k > 0) mesh.translate(new G.THREE.Vector3(step[0], step[1], step[2]));
        G.animate(scene, t);
        tracer.update(t);
        tracer.render();
        const v = new Float64Array(G.g_verts.length * 3);
        G.g_verts.forEach((p, i) => { v[3 * i] = p.x; v[3 * i + 1] = p.y; v[3 * i + 2] = p.z; });
        frames.push({ time: t, rgba8_sha256: sha(tracer.pixels), verts_sha256: sha(v), kd_box: Array.from(tracer.exportKdBox()), start_time: tracer.startTime });
    });
    console.log(JSON.stringify({ width: G.WIDTH, height: G.HEIGHT, depth: G.MAX_RECURSION_DEPTH, frames }));
    tracer.dispose();
    process.exit(0);
})().catch((e) => { console.error(e.stack || e); process.exit(1); });

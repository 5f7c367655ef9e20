'use strict';
// Animated frames from JavaScript (the job of benchmark configuration 5): light 0 on the reference's orbit, the mesh dragged
// with Mesh.translate (what the reference's mouse drag calls, js/8-kd_tree.js:2055-2064), the k-d tree rebuilt on the
// GPU for every frame, and two RayTracer objects alternating frames so that JavaScript prepares frame k+1 (moves the
// mesh, packs the arrays) while the GPU traces frame k. Prints a JSON summary.
//   node animate.js [--frames 60] [--width 3840 --height 2160] [--depth 4] [--model models/bun_zipper.ply]
//                   [--kd-builder device|host] [--tracers 2]
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const rt = require('..');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const G = rt.createContext({
    WIDTH: parseInt(arg('width', '3840'), 10), HEIGHT: parseInt(arg('height', '2160'), 10),
    MAX_RECURSION_DEPTH: parseInt(arg('depth', '4'), 10), SUPERSAMPLING: parseInt(arg('ss', '1'), 10),
    MODEL_FILE_NAME: arg('model', 'models/bun_zipper.ply'), KD_BUILDER: arg('kd-builder', 'device'),
    KD_REBUILD_INTERVAL: 0,     // geometry and tree change together, every frame
});
const sceneFile = path.join(__dirname, 'default_scene.js');
rt.run(G, fs.readFileSync(sceneFile, 'utf8'), sceneFile);
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');

(async () => {
    const data = await G.loadPLY(G.MODEL_FILE_NAME);
    const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
    const mesh = scene.objects[scene.objects.length - 1];
    const frames = parseInt(arg('frames', '60'), 10);
    const tracers = [];
    for (let i = 0; i < parseInt(arg('tracers', '2'), 10); i++) { const t = new G.RayTracer(); t.setScene(scene); tracers.push(t); }
    const busy = tracers.map(() => false);
    let rays = 0, last = null, lastCanvas = null;
    const finish = (i) => {
        if (!busy[i]) return;
        tracers[i].updateEnd();
        lastCanvas = tracers[i].render();
        rays += tracers[i].stats().raysTotal;
        last = tracers[i];
        busy[i] = false;
    };
    const pos = (k) => { const a = 2 * Math.PI * k / 240; return [12 * Math.sin(a), 0, 12 * (Math.cos(a) - 1)]; };
    const frame = (k) => {
        const i = k % tracers.length;
        finish(i);
        const p = pos(k), q = pos(k - 1);
        G.animate(scene, k / 60);
        if (k > 0) mesh.translate(new G.THREE.Vector3(p[0] - q[0], p[1] - q[1], p[2] - q[2]));
        tracers[i].updateBegin(k / 60);
        busy[i] = true;
    };
    const warm = tracers.length;
    for (let k = 0; k < warm; k++) frame(k);
    tracers.forEach((t, i) => finish(i));
    rays = 0;
    const t0 = process.hrtime.bigint();
    for (let k = warm; k < warm + frames; k++) frame(k);
    for (let j = 0; j < tracers.length; j++) finish((warm + frames + j) % tracers.length);
    const ms = Number(process.hrtime.bigint() - t0) / 1e6;
    const st = last.stats();
    console.log(JSON.stringify({ width: G.WIDTH, height: G.HEIGHT, depth: G.MAX_RECURSION_DEPTH, model: G.MODEL_FILE_NAME, frames,
        tracers: tracers.length, kd_builder: G.KD_BUILDER, ms_per_frame: ms / frames, frames_per_s: frames / (ms / 1e3),
        mrays_per_s: rays / ms / 1e3, kd_build_ms: st.kdBuildMs, trace_ms: st.traceMs, last_canvas_sha256: sha(lastCanvas) }));
    tracers.forEach((t) => t.dispose());
    process.exit(0);
})().catch((e) => { console.error(e.stack || e); process.exit(1); });

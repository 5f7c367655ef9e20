'use strict';
// The animated job of benchmark configuration 5 with its FRAMES sharded over the GPUs of a node, from JavaScript: one Node
// process per GPU, frame k rendered by process k mod N. Frames are independent given the time (the scene is replicated and
// every process moves the mesh through every frame's Mesh.translate, so the geometry of frame k is the same whoever
// renders it), so there is no exchange between the processes: each one needs only what examples/animate.js needs. The
// parent starts the workers, lets them start together, and adds up what they report.
//   node animate_sharded.js --gpus N [--devices 0,1,...] [--frames 60] [--width 3840 --height 2160] [--depth 4]
//                           [--model models/bun_zipper.ply] [--kd-builder device|host] [--hash all|last]
// --hash all: the SHA-256 of EVERY frame's canvas (13 ms per 4K frame on the host - for checks, not for timing); last
// (default): only of each worker's final frame.
// --devices names the HIP device of every worker (default 0 .. N-1); several workers may share one GPU (rehearsal).
// Prints one JSON line: frames per second over all workers and the canvas hashes in frame order (null = not hashed).
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { fork } = require('child_process');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const N = parseInt(arg('gpus', '1'), 10);
const FRAMES = parseInt(arg('frames', '60'), 10);
const devices = arg('devices', Array.from({ length: N }, (_, i) => i).join(',')).split(',').map((s) => parseInt(s, 10));
if (devices.length !== N) throw new Error('--devices must name one device per worker');

if (process.env.RT_SHARD_RANK === undefined) {
    // ---- parent: never touches the GPU
    const workers = [];
    const ready = new Set(), done = new Map();
    let failed = false, exited = 0;
    for (let r = 0; r < N; r++) {
        const w = fork(__filename, process.argv.slice(2), { env: Object.assign({}, process.env, { RT_SHARD_RANK: String(r), RT_SHARD_DEVICE: String(devices[r]) }) });
        w.on('message', (m) => {
            if (m.ready) { ready.add(r); if (ready.size === N) workers.forEach((x) => x.send({ go: true })); }   // all start together
            if (m.done) done.set(r, m.done);
        });
        w.on('exit', (code) => {
            exited++;
            if (code !== 0 && !failed) { failed = true; workers.forEach((x) => x.kill()); console.error('worker ' + r + ' failed with status ' + code); process.exit(1); }
            if (!failed && exited === N) report();
        });
        workers.push(w);
    }
    const report = () => {
        if (done.size !== N) { console.error('a worker ended without its report'); process.exit(1); }
        const shas = new Array(FRAMES).fill(undefined);
        let ms = 0, rays = 0;
        for (const d of done.values()) {
            ms = Math.max(ms, d.ms);
            rays += d.rays;
            for (const [k, h] of d.frames) shas[k] = h;
        }
        if (shas.some((h) => h === undefined)) { console.error('a frame is missing'); process.exit(1); }
        const first = done.get(0);
        console.log(JSON.stringify({ gpus: N, devices, frames: FRAMES, width: first.width, height: first.height, depth: first.depth, model: first.model,
            kd_builder: first.kd_builder, sharding: 'frames (frame k on worker k mod N, no exchange)', ms_per_frame: ms / FRAMES, frames_per_s: FRAMES / (ms / 1e3),
            mrays_per_s: rays / ms / 1e3, frame_sha256: shas }));
        process.exit(0);
    };
} else {
    // ---- worker: one GPU, the frames k = rank, rank + N, ...
    const rank = parseInt(process.env.RT_SHARD_RANK, 10);
    const rt = require('..');
    const G = rt.createContext({
        WIDTH: parseInt(arg('width', '3840'), 10), HEIGHT: parseInt(arg('height', '2160'), 10),
        MAX_RECURSION_DEPTH: parseInt(arg('depth', '4'), 10), SUPERSAMPLING: parseInt(arg('ss', '1'), 10),
        MODEL_FILE_NAME: arg('model', 'models/bun_zipper.ply'), KD_BUILDER: arg('kd-builder', 'device'),
        KD_REBUILD_INTERVAL: 0, DEVICE: parseInt(process.env.RT_SHARD_DEVICE, 10),
    });
    const sceneFile = path.join(__dirname, 'default_scene.js');
    rt.run(G, fs.readFileSync(sceneFile, 'utf8'), sceneFile);
    const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');
    (async () => {
        const data = await G.loadPLY(G.MODEL_FILE_NAME);
        const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
        const mesh = scene.objects[scene.objects.length - 1];
        const tracers = [new G.RayTracer(), new G.RayTracer()];   // the host prepares its next frame while the GPU traces this one
        tracers.forEach((t) => t.setScene(scene));
        // one frame per tracer before the clock starts (buffers sized, tree built once); the scene is put back afterwards
        tracers.forEach((t) => { t.update(0); t.render(); });
        const pos = (k) => { const a = 2 * Math.PI * k / 240; return [12 * Math.sin(a), 0, 12 * (Math.cos(a) - 1)]; };
        const busy = [null, null];
        const frames = [];
        let rays = 0;
        const hashAll = arg('hash', 'last') === 'all';
        const lastOfMine = FRAMES - 1 - ((FRAMES - 1 - rank) % N + N) % N;   // the last frame index k with k mod N == rank
        const finish = (i) => {
            if (busy[i] === null) return;
            tracers[i].updateEnd();
            const canvas = tracers[i].render();
            frames.push([busy[i], (hashAll || busy[i] === lastOfMine) ? sha(canvas) : null]);
            rays += tracers[i].stats().raysTotal;
            busy[i] = null;
        };
        process.send({ ready: true });
        await new Promise((resolve) => process.once('message', resolve));
        const t0 = process.hrtime.bigint();
        let mine = 0;
        for (let k = 0; k < FRAMES; k++) {
            // every worker moves the mesh through every frame: the vertices of frame k then carry the same rounding history
            // whoever renders it (Mesh.translate is a chain of float32-matrix round trips, js/8-kd_tree.js:2055-2064)
            if (k > 0) { const p = pos(k), q = pos(k - 1); mesh.translate(new G.THREE.Vector3(p[0] - q[0], p[1] - q[1], p[2] - q[2])); }
            if (k % N !== rank) continue;
            const i = mine++ % 2;
            finish(i);
            G.animate(scene, k / 60);
            tracers[i].updateBegin(k / 60);
            busy[i] = k;
        }
        finish(mine % 2);
        finish((mine + 1) % 2);
        const ms = Number(process.hrtime.bigint() - t0) / 1e6;
        process.send({ done: { ms, rays, frames, width: G.WIDTH, height: G.HEIGHT, depth: G.MAX_RECURSION_DEPTH, model: G.MODEL_FILE_NAME, kd_builder: G.KD_BUILDER } }, () => {
            tracers.forEach((t) => t.dispose());
            process.exit(0);
        });
    })().catch((e) => { console.error(e.stack || e); process.exit(1); });
}

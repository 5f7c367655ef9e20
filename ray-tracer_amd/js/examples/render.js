'use strict';
// Renders one frame of a scene description on the GPU and prints a JSON summary (hashes of the RGBA8 render
// target and of the presented canvas, Lbar, sf, ray counts, timings). Optional: --ppm out.ppm writes the canvas.
//   node render.js [--scene file.js] [--model models/bun_zipper_res4.ply] [--width W --height H] [--depth D]
//                  [--ss S] [--time T] [--tr 0|1] [--frames N] [--mesh-material ka,kd,ks,ke,kReflect,kRefract,ior]
//                  [--kd-builder host|device] [--split 1: update() as updateBegin() + updateEnd()]
//                  [--pick x,y] [--drag x0,y0,x1,y1[,x2,y2...]: after the frame(s), click at (x0,y0), press again to grab,
//                   move through the other positions, release, and render one more frame 0.5 s later]
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const rt = require('..');

const arg = (name, dflt) => { const i = process.argv.indexOf('--' + name); return i > 0 ? process.argv[i + 1] : dflt; };
const opts = {
    WIDTH: parseInt(arg('width', '640'), 10), HEIGHT: parseInt(arg('height', '480'), 10),
    MAX_RECURSION_DEPTH: parseInt(arg('depth', '1'), 10), SUPERSAMPLING: parseInt(arg('ss', '1'), 10),
    TONE_REPRODUCTION_METHOD: parseInt(arg('tr', '0'), 10), MODEL_FILE_NAME: arg('model', 'models/bun_zipper_res4.ply'),
    DEVICE: parseInt(arg('device', '0'), 10), KD_BUILDER: arg('kd-builder', 'host'),
};
const mm = arg('mesh-material', null);
if (mm) opts.MESH_MATERIAL = mm.split(',').map(parseFloat);
const G = rt.createContext(opts);
const sceneFile = arg('scene', path.join(__dirname, 'default_scene.js'));
rt.run(G, fs.readFileSync(sceneFile, 'utf8'), sceneFile);
const sha = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');

(async () => {
    const data = await G.loadPLY(G.MODEL_FILE_NAME);
    const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
    const tracer = new G.RayTracer();
    tracer.setScene(scene);
    const time = parseFloat(arg('time', '0'));
    const frames = parseInt(arg('frames', '1'), 10);
    let canvas = null;
    const t0 = process.hrtime.bigint();
    for (let f = 0; f < frames; f++) {
        G.animate(scene, time + f / 60);
        if (arg('split', '0') === '1') { tracer.updateBegin(time + f / 60); tracer.updateEnd(); }
        else tracer.update(time + f / 60);
        canvas = tracer.render();
    }
    const ms = Number(process.hrtime.bigint() - t0) / 1e6 / frames;
    const st = tracer.stats();
    const out = { width: G.WIDTH, height: G.HEIGHT, depth: G.MAX_RECURSION_DEPTH, model: G.MODEL_FILE_NAME, frames,
        Lbar: tracer.Lbar, sf: tracer.sf, rgba8_sha256: sha(tracer.pixels), canvas_sha256: sha(canvas),
        rays: st.raysTotal, shadow_rays: st.shadowRays, trace_ms: st.traceMs, kd_build_ms: st.kdBuildMs, wall_ms_per_frame: ms };
    const dg = arg('drag', null);
    if (dg) {
        const d = dg.split(',').map(Number), ui = new G.UI(tracer);
        ui.mouseDown(d[0], d[1]);
        ui.mouseUp(d[0], d[1]);
        out.drag_selected = ui.selectedObject ? ui.selectedObject.constructor.name : null;
        out.drag_grabbed = ui.selectedObject ? ui.mouseDown(d[0], d[1]) && ui.moving : false;
        for (let k = 2; k + 1 < d.length; k += 2) ui.mouseMove(d[k], d[k + 1]);
        ui.mouseUp(d[d.length - 2], d[d.length - 1]);
        if (ui.selectedObject) out.drag_model = Array.from(ui.selectedObject.modelTransform.elements);
        const t1 = time + (frames - 1) / 60 + 0.5;      // past the rebuild interval: the tree follows the mesh
        G.animate(scene, t1);
        tracer.update(t1);
        canvas = tracer.render();
        out.drag_rgba8_sha256 = sha(tracer.pixels);
        out.drag_canvas_sha256 = sha(canvas);
    }
    const pk = arg('pick', null);   // --pick x,y (canvas pixel, y from the top): class name of the object under it
    if (pk) { const o = tracer.pick(...pk.split(',').map(Number)); out.picked = o ? o.constructor.name : null; }
    const ppm = arg('ppm', null);
    if (ppm) {   // canvas rows are bottom-up; PPM is top-down
        const W = G.WIDTH, H = G.HEIGHT, body = Buffer.alloc(W * H * 3);
        for (let y = 0; y < H; y++) for (let x = 0; x < W; x++) for (let c = 0; c < 3; c++) body[((H - 1 - y) * W + x) * 3 + c] = canvas[(y * W + x) * 4 + c];
        fs.writeFileSync(ppm, Buffer.concat([Buffer.from('P6\n' + W + ' ' + H + '\n255\n'), body]));
    }
    console.log(JSON.stringify(out));
    tracer.dispose();
    // Node 12 can fault while tearing its environment down when a garbage-collection pass is still pending (large
    // typed arrays make that likely); leaving through process.exit() skips that teardown
    process.exit(0);
})().catch((e) => { console.error(e.stack || e); process.exit(1); });

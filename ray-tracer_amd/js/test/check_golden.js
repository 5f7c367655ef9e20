'use strict';
// Checks the JavaScript host against one golden fixture directory (tests/golden/<name>): every uniform the
// reference handed to WebGL, the mesh arrays and the k-d tree export must match bit for bit.
//   node check_golden.js <golden dir> [path of a page script whose generateScene() should be used | -] [TrackballControls.js]
// Without the second argument (or with -) the scene comes from examples/default_scene.js. The third argument is three.js's
// own camera-controls file: a fixture with recorded mouse events is then replayed through it, otherwise the recorded camera
// pose is taken over as it is. Runs on a host-only context.
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const rt = require('..');

const dir = process.argv[2];
const pageScript = process.argv[3] && process.argv[3] !== '-' ? process.argv[3] : null;
const controlsFile = process.argv[4];
const man = JSON.parse(fs.readFileSync(path.join(dir, 'manifest.json'), 'utf8'));
const sha = (ta) => crypto.createHash('sha256').update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest('hex');

function extractGenerateScene(text) {
    const start = text.indexOf('function generateScene(');
    if (start < 0) throw new Error('no generateScene in ' + pageScript);
    const end = text.indexOf('\n}\n', start);
    return text.slice(start, end + 3);
}

const edits = man.constant_edits || {};
const opts = { WIDTH: man.width, HEIGHT: man.height, DEVICE: -1 };
let extra = null;   // a second mesh: "<file> scale <s> translate <x,y,z> ..." as the fixture generator records it
for (const k of Object.keys(edits)) {
    if (k === 'meshMaterial') opts.MESH_MATERIAL = edits[k].split(',').map(parseFloat);
    else if (k === 'extra_mesh') {
        const m = /^(\S+) scale (\S+) translate (\S+)/.exec(edits[k]);
        extra = { file: m[1], scale: parseFloat(m[2]), translate: m[3].split(',').map(parseFloat) };
        opts.REBASE_MESH_INDICES = true;
    } else opts[k] = JSON.parse(edits[k]);
}
const G = rt.createContext(opts);
const source = pageScript ? extractGenerateScene(fs.readFileSync(pageScript, 'utf8'))
    : fs.readFileSync(path.join(__dirname, '..', 'examples', 'default_scene.js'), 'utf8');
rt.run(G, source, pageScript || 'default_scene.js');

(async () => {
    const failures = [];
    const data = await G.loadPLY(G.MODEL_FILE_NAME);
    const scene = G.generateScene({ [G.MODEL_FILE_NAME]: data });
    if (pageScript && opts.MESH_MATERIAL) {   // the page script has the mesh's constants inline: set them afterwards
        const mat = scene.objects[scene.objects.length - 1].faces[0].getMaterial();
        ['ka', 'kd', 'ks', 'ke', 'kReflect', 'kRefract', 'ior'].forEach((f, i) => { mat[f] = opts.MESH_MATERIAL[i]; });
    }
    if (extra) {
        const d2 = await (/\.obj$/.test(extra.file) ? G.loadOBJ : G.loadPLY)(extra.file);
        const m2 = new G.Mesh(G.nextObjectId++, d2.verts, d2.faces);
        m2.scale(new rt.THREE.Vector3(extra.scale, extra.scale, extra.scale));
        m2.translate(new rt.THREE.Vector3(...extra.translate));
        m2.setMaterial(scene.objects[scene.objects.length - 1].faces[0].getMaterial());
        scene.objects.push(m2);
    }
    G.animate(scene, man.time);
    const tracer = new G.RayTracer();
    tracer.setScene(scene);
    tracer.prepare(man.time);
    let frameTime = man.time, selectionSegments = null, selectionPixels = 0;
    if (man.drag) {
        // the reference's ui.mouseDown / mouseMove / mouseUp were driven with these pixels; here the click's pick is
        // replaced by its recorded result (picking needs a traced frame, this check runs without a GPU)
        const d = man.drag, ui = new G.UI(tracer);
        const near = (a, b) => a.length === b.length && a.every((x, i) => x === b[i]);
        ui.select(scene.objects[d.selected_index]);
        if (ui.mouseDown(d.mouse[0], d.mouse[1]) !== d.grab_returned || ui.moving !== d.moving) failures.push('drag: the press did not grab the object');
        if (ui.zPos !== d.zPos) failures.push('drag: zPos ' + ui.zPos + ' != ' + d.zPos);
        if (!near([ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z], d.old_position)) failures.push('drag: oldPosition differs');
        const got = [];
        for (let k = 2; k + 1 < d.mouse.length; k += 2) {
            ui.mouseMove(d.mouse[k], d.mouse[k + 1]);
            got.push([ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z]);
        }
        ui.mouseUp(d.mouse[d.mouse.length - 2], d.mouse[d.mouse.length - 1]);
        got.push([ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z]);
        got.forEach((p, i) => { if (!near(p, d.positions[i])) failures.push('drag: position ' + i + ' = [' + p + '] want [' + d.positions[i] + ']'); });
        const so = ui.selectedObject;
        if (!near(Array.from(so.modelTransform.elements), d.model_after)) failures.push('drag: modelTransform differs');
        if (!near(Array.from(so.invModelTransform.elements), d.inv_model_after)) failures.push('drag: invModelTransform differs');
        frameTime = man.time + 0.5;     // the generator's next tick, past the rebuild throttle
        G.animate(scene, frameTime);
        ui.prepareCamera();
        tracer.prepare(frameTime);
        // the selection box the reference drew round the dragged object in that tick (uniforms of its line program)
        const box = ui.selectionBox();
        for (const name of ['cubeMin', 'cubeMax', 'modelViewProjection']) {
            const u = man.uniforms.filter((x) => x.name === name).pop();
            if (!u) failures.push('fixture holds no ' + name);
            else if (!near(Array.from(box[name]), Array.from(new Float32Array(u.value)))) failures.push('selection box: ' + name + ' differs');
        }
        selectionSegments = ui.selectionBoxSegments(man.width, man.height);
        for (const sg of selectionSegments)
            if (!(Math.min(sg[0], sg[2]) >= 0 && Math.max(sg[0], sg[2]) <= man.width && Math.min(sg[1], sg[3]) >= 0 && Math.max(sg[1], sg[3]) <= man.height))
                failures.push('selection box: edge [' + sg + '] leaves the canvas');
        const px = new Uint8Array(man.width * man.height * 4);
        ui.drawSelectionBox(px, man.width, man.height);
        for (let i = 0; i < px.length; i += 4) if (px[i] === 255) selectionPixels++;
    }
    let controlsReplayed = false;
    if (man.trackball) {
        // camera controls: the mouse events the reference's THREE.TrackballControls received, tick by tick
        const tb = man.trackball;
        const same = (a, b) => a.length === b.length && a.every((x, i) => x === b[i]);
        const cam = G.camera;
        if (controlsFile) {
            rt.run(G, fs.readFileSync(controlsFile, 'utf8'), controlsFile);
            const ui = new G.UI(tracer);
            tracer.pick = () => null;   // the presses of the fixture land on the background (no GPU frame to pick from here)
            let t = man.time, frame = 0, lx = 0, ly = 0;
            for (const step of tb.steps) {
                const f = step.split(':');
                if (f[0] === 'down') { lx = +f[2]; ly = +f[3]; ui.handleEvent('mousedown', { button: +f[1], x: lx, y: ly }); }
                else if (f[0] === 'move') { lx = +f[1]; ly = +f[2]; ui.handleEvent('mousemove', { x: lx, y: ly }); }
                else if (f[0] === 'up') { lx = +f[1]; ly = +f[2]; ui.handleEvent('mouseup', { x: lx, y: ly }); }
                else if (f[0] === 'wheel') ui.handleEvent('wheel', { deltaY: +f[1], x: lx, y: ly });
                else if (f[0] === 'tick') {
                    t += 1 / 60;
                    G.animate(scene, t);
                    const vp = ui.prepareCamera();
                    const want = tb.frames[frame];
                    const got = { position: [cam.position.x, cam.position.y, cam.position.z], up: [cam.up.x, cam.up.y, cam.up.z],
                        quaternion: [cam.quaternion.x, cam.quaternion.y, cam.quaternion.z, cam.quaternion.w],
                        matrixWorld: Array.from(cam.matrixWorld.elements), viewProjection: Array.from(vp.elements),
                        target: [ui.controls.target.x, ui.controls.target.y, ui.controls.target.z] };
                    for (const k of Object.keys(got)) if (!same(got[k], want[k])) failures.push('controls: frame ' + frame + ' ' + k + ' = [' + got[k] + '] want [' + want[k] + ']');
                    frame++;
                }
            }
            if (frame !== tb.frames.length) failures.push('controls: ' + frame + ' frames replayed, fixture has ' + tb.frames.length);
            controlsReplayed = true;
        } else {
            const last = tb.frames[tb.frames.length - 1];
            cam.position.set(...last.position);
            cam.up.set(...last.up);
            cam.quaternion.set(...last.quaternion);
            cam.updateMatrixWorld(true);
            if (!same(Array.from(cam.matrixWorld.elements), last.matrixWorld)) failures.push('camera: matrixWorld composed from the recorded pose differs');
        }
        frameTime = tb.last_time;
        G.animate(scene, frameTime);
        tracer.prepare(frameTime);
    }

    // a uniform that was set several times (several ticks) is compared with its last value
    const lastValue = new Map();
    for (const u of man.uniforms) lastValue.set(u.name, u);
    let checked = 0;
    for (const u of lastValue.values()) {
        if (!(u.name in tracer.uniforms)) continue;
        const v = tracer.uniforms[u.name];
        let got;
        if (v instanceof rt.THREE.Vector3) got = new Float32Array([v.x, v.y, v.z]);
        else if (v instanceof rt.THREE.Matrix4) got = new Float32Array(v.toArray());
        else got = new Float32Array([v]);
        const want = new Float32Array(u.value);
        checked++;
        if (got.length !== want.length || !got.every((x, i) => Object.is(x, want[i]) || x === want[i])) {
            failures.push('uniform ' + u.name + ': got [' + Array.from(got) + '] want [' + Array.from(want) + ']');
        }
    }
    const expectNames = man.uniforms.map((u) => u.name).filter((n) => /^(model\d|invModel|phong|pointLights|ambientLight|bgColor|cameraPos|ray|kdBox)/.test(n));   // cubeMin / cubeMax / modelViewProjection belong to the selection box and are checked with the drag
    for (const n of expectNames) if (!(n in tracer.uniforms)) failures.push('uniform ' + n + ' was not produced');

    const k = tracer._meshKey;
    const hashes = { verts_f64: sha(k.verts), verts_f32: sha(new Float32Array(k.verts)), faces_u32: sha(k.faces), face_mat_u32: sha(k.mats) };
    for (const name of Object.keys(man.sha256)) {
        if (!(name in hashes)) failures.push('sha256 ' + name + ' is not checked');
        else if (hashes[name] !== man.sha256[name]) failures.push('sha256 ' + name + ' differs');
    }
    const kd = tracer.exportKdTree();
    if (kd.nodes.length / 2 !== man.kd.n_nodes) failures.push('k-d node count ' + kd.nodes.length / 2 + ' != ' + man.kd.n_nodes);
    if (sha(kd.nodes) !== man.kd.sha256.nodes_sparse_u32) failures.push('k-d nodes differ');
    if (sha(kd.leaves) !== man.kd.sha256.leaves_i32) failures.push('k-d leaves differ');
    if (sha(kd.box) !== man.kd.sha256.root_box_f64) failures.push('k-d root box differs');

    console.log(JSON.stringify({ golden: path.basename(dir), scene: pageScript ? 'page script' : 'example', uniforms_checked: checked, controls_replayed: controlsReplayed, selection_segments: selectionSegments, selection_pixels: selectionPixels,
        n_verts: k.verts.length / 3, n_faces: k.faces.length / 3, kd_nodes: kd.nodes.length / 2, failures }));
    tracer.dispose();
    process.exit(failures.length ? 1 : 0);
})().catch((e) => { console.error(e.stack || e); process.exit(2); });

#!/usr/bin/env node
/*
 * Golden-vector generator: runs the reference's own JavaScript (read at run time from
 * /root/reference, never copied) under Node with a stub DOM and a recording fake WebGL
 * context, and dumps the host-side artefacts of one frame:
 *   - every uniform the reference sets (name -> f32 values), in setUniforms order
 *   - g_verts (f64) / vertsData (f32) / g_faces / face materials
 *   - the k-d tree as (heapIndex, metadata) pairs + leaf array + root box (f64)
 *   - Lbar / sf for a synthetic RGBA8 frame handed to readPixels
 *
 * This is test infrastructure only. It exists in this container; nothing that runs on the
 * GPU box imports it. Usage:
 *   node ref_capture.js --script 8-kd_tree.js --model models/bun_zipper_res4.ply \
 *        --width 640 --height 480 --depth 1 --ss 1 --time 0 --out DIR [--full-arrays]
 *        [--set NAME=VALUE ...]   (edits `const NAME = ...;` lines, which is how the
 *                                  reference is configured: js/8-kd_tree.js:25-46)
 *        [--mesh-material ka,kd,ks,ke,kReflect,kRefract,ior]   (edits the constants of the mesh's Phong
 *                                  material inside generateScene, js/8-kd_tree.js:2963)
 *        [--extra-obj models/teapot.obj --extra-scale 60 --extra-translate 50,75,10]
 *              a second mesh: loaded by the reference's own loadOBJ (js/8-kd_tree.js:2851-2887), constructed,
 *              scaled and translated with the reference's own Mesh class and given the first mesh's material.
 *              Mesh.pushToGlobal appends faces without rebasing their vertex indices (js/8-kd_tree.js:2114-2115;
 *              the reference only ever loads one mesh), so THIS HARNESS adds the mesh's first vertex index to
 *              its faces afterwards - the one step that is not the reference's own code.
 *        [--drag x0,y0,x1,y1]  drives ui.mouseDown(x0,y0), ui.mouseMove(x1,y1), ui.mouseUp(x1,y1)
 *        [--trackball steps]   camera controls: DOM events delivered to the listeners the reference's
 *                              THREE.TrackballControls registered on the canvas, and ticks in between; steps is a
 *                              comma list of down:button:x:y | move:x:y | up:x:y | wheel:deltaY | tick
 *              (js/8-kd_tree.js:2718-2792) after the first tick and records what was picked and how the
 *              picked object's transform changed; a further tick then renders the moved scene
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const crypto = require('crypto');

const REF = process.env.RT_REFERENCE || '/root/reference';

function parseArgs(argv) {
    const a = { script: '8-kd_tree.js', width: 640, height: 480, time: 0, sets: {}, fullArrays: false,
                pixelPattern: 'lcg', out: null, ticks: 1,
                meshMaterial: null, extraObj: null, extraScale: 1, extraTranslate: [0, 0, 0], drag: null, dragDt: 0.5, trackball: null };
    for (let i = 2; i < argv.length; i++) {
        const k = argv[i];
        if (k === '--script') a.script = argv[++i];
        else if (k === '--model') a.sets.MODEL_FILE_NAME = JSON.stringify(argv[++i]);
        else if (k === '--width') a.width = +argv[++i];
        else if (k === '--height') a.height = +argv[++i];
        else if (k === '--depth') a.sets.MAX_RECURSION_DEPTH = argv[++i];
        else if (k === '--ss') a.sets.SUPERSAMPLING = argv[++i];
        else if (k === '--time') a.time = +argv[++i];
        else if (k === '--out') a.out = argv[++i];
        else if (k === '--full-arrays') a.fullArrays = true;
        else if (k === '--ticks') a.ticks = +argv[++i];
        else if (k === '--pixels') a.pixelPattern = argv[++i];
        else if (k === '--mesh-material') a.meshMaterial = argv[++i].split(',').map(Number);
        else if (k === '--extra-obj') a.extraObj = argv[++i];
        else if (k === '--extra-scale') a.extraScale = +argv[++i];
        else if (k === '--extra-translate') a.extraTranslate = argv[++i].split(',').map(Number);
        else if (k === '--drag') a.drag = argv[++i].split(',').map(Number);
        else if (k === '--drag-dt') a.dragDt = +argv[++i];   // seconds between the first tick and the tick after the drag (default 0.5)
        else if (k === '--trackball') a.trackball = argv[++i].split(',');
        else if (k === '--set') { const [n, v] = argv[++i].split('='); a.sets[n] = v; }
        else if (k === '--scene-js') a.sceneJs = argv[++i];     // a file with another `function generateScene(meshes)`, written against the reference's classes
        else if (k === '--dump-glsl') a.dumpGlsl = argv[++i];   // directory OUTSIDE the repository: shader sources + texture payloads for oracle/gen_golden/glsl_exec
        else throw new Error('unknown arg ' + k);
    }
    if (!a.out) throw new Error('--out required');
    return a;
}

const args = parseArgs(process.argv);

// ---- fake GL: records every call; constants evaluate to their own name -------------
const glCalls = [];
const shaderSources = [];
let readPixelsFill = null;
const glTarget = {
    getShaderParameter: () => true,
    getProgramParameter: () => true,
    getUniformLocation: (prog, name) => name,
    getAttribLocation: () => 0,
    createShader: () => ({}), createProgram: () => ({}), createBuffer: () => ({}),
    createTexture: () => ({}), createFramebuffer: () => ({}),
    getExtension: () => ({}),
    shaderSource: (sh, src) => { shaderSources.push(src); },
    readPixels: (x, y, w, h, fmt, type, dst) => { if (readPixelsFill) readPixelsFill(dst, w, h); },
};
const gl = new Proxy(glTarget, {
    get(t, prop) {
        if (prop in t) {
            const f = t[prop];
            return function (...a) { glCalls.push([prop, a]); return f.apply(t, a); };
        }
        if (typeof prop === 'string' && /^[A-Z0-9_]+$/.test(prop)) return prop;
        return function (...a) { glCalls.push([prop, a]); };
    }
});

// ---- stub DOM ----------------------------------------------------------------------
// listeners by event type: TrackballControls registers its handlers on the canvas and, while a button is down, on the document
const canvasListeners = {}, documentListeners = {};
const canvas = {
    clientWidth: args.width, clientHeight: args.height, width: args.width, height: args.height,
    getContext: (kind) => (kind === 'webgl2' || kind === 'experimental-webgl' || kind === 'webgl') ? gl : null,
    addEventListener(type, fn) { (canvasListeners[type] = canvasListeners[type] || []).push(fn); }, removeEventListener() {}, style: {},
    getBoundingClientRect: () => ({ left: 0, top: 0, width: args.width, height: args.height }),
    ownerDocument: null,
};
const document = {
    getElementById: () => canvas,
    body: { appendChild() {}, scrollLeft: 0, scrollTop: 0 },
    documentElement: { scrollLeft: 0, scrollTop: 0, clientLeft: 0, clientTop: 0 },
    addEventListener(type, fn) { (documentListeners[type] = documentListeners[type] || []).push(fn); },
    removeEventListener(type, fn) { const l = documentListeners[type] || [], i = l.indexOf(fn); if (i >= 0) l.splice(i, 1); },
    createElement: () => ({ style: {}, appendChild() {}, addEventListener() {}, getContext: () => null }),
    createElementNS: () => ({ style: {} }),
};
canvas.ownerDocument = document;
let tickFn = null;
const sandbox = {
    document, window: null, console, alert: (m) => { throw new Error('alert: ' + m); },
    setInterval: (fn) => { tickFn = fn; return 1; },
    Stats: function () { this.dom = {}; this.begin = () => {}; this.end = () => {}; },
    // main-realm typed arrays: PLYLoader tests `data instanceof ArrayBuffer`
    ArrayBuffer, Float32Array, Float64Array, Uint8Array, Uint16Array, Uint32Array, Int8Array,
    Int16Array, Int32Array, DataView, Promise, Date, Math, Number, String, Array, Object, JSON,
    Error, parseFloat, parseInt, isNaN, Symbol, RegExp,
    performance: { now: () => 0 },
    pageXOffset: 0, pageYOffset: 0, innerWidth: args.width, innerHeight: args.height,
};
sandbox.window = sandbox;
sandbox.self = sandbox;
sandbox.addEventListener = () => {};
sandbox.removeEventListener = () => {};
vm.createContext(sandbox);

function runRef(rel, edit) {
    let src = fs.readFileSync(path.join(REF, rel), 'utf8');
    if (edit) src = edit(src);
    vm.runInContext(src, sandbox, { filename: path.join(REF, rel) });
}

// three.js r84 is a UMD bundle; load it as a plain script so that `THREE` is a global
runRef('js/three.min.js', (s) => 'var module=undefined, exports=undefined, define=undefined;\n' + s);
vm.runInContext(`
THREE.FileLoader.prototype.load = function (url, onLoad) {
    var data = __readFile(url, this.responseType);
    onLoad(data);
};`, sandbox);
sandbox.__readFile = (url, responseType) => {
    const buf = fs.readFileSync(path.join(REF, url));
    if (responseType === 'arraybuffer') {
        return buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
    }
    return buf.toString('utf8');
};
runRef('js/loaders/PLYLoader.js');
runRef('js/loaders/OBJLoader.js');
runRef('js/controls/TrackballControls.js');

// synthetic frame for the log-luminance loop (js/8-kd_tree.js:2558-2569)
readPixelsFill = (dst, w, h) => {
    if (args.pixelPattern === 'zero') { dst.fill(0); return; }
    if (args.pixelPattern.startsWith('file:')) {   // a frame made elsewhere (the executed tracer shader's, quantised): RGBA8, rows bottom-up
        const buf = fs.readFileSync(args.pixelPattern.slice(5));
        if (buf.length !== dst.length) throw new Error('--pixels file: ' + buf.length + ' bytes, readPixels wants ' + dst.length);
        dst.set(buf);
        return;
    }
    if (args.pixelPattern.startsWith('const:')) {
        const [r, g, b] = args.pixelPattern.slice(6).split(',').map(Number);
        for (let i = 0; i < dst.length; i += 4) { dst[i] = r; dst[i + 1] = g; dst[i + 2] = b; dst[i + 3] = 255; }
        return;
    }
    // 32-bit LCG (Numerical Recipes constants); byte = top 8 bits. Tests regenerate the same stream.
    let s = 12345 >>> 0;
    for (let i = 0; i < dst.length; i++) {
        s = (Math.imul(s, 1664525) + 1013904223) >>> 0;
        dst[i] = s >>> 24;
    }
};

// configuration = constant edits at the top of the reference script
const usedSets = {};
runRef('js/' + args.script, (src) => {
    for (const name of Object.keys(args.sets)) {
        const re = new RegExp('^(const|var)\\s+' + name + '\\s*=\\s*[^;]*;', 'm');
        if (!re.test(src)) throw new Error('constant not found: ' + name);
        src = src.replace(re, (m, kw) => { usedSets[name] = args.sets[name]; return kw + ' ' + name + ' = ' + args.sets[name] + ';'; });
    }
    if (args.meshMaterial) {
        const re = /(var meshMaterial = new Phong\(nextPhongId\+\+,)[^\n]*\n/;
        if (!re.test(src)) throw new Error('mesh material line not found');
        src = src.replace(re, (m, head) => head + ' ' + args.meshMaterial.join(', ') + ',\n');
        usedSets.meshMaterial = args.meshMaterial.join(',');
    }
    if (args.extraObj) {
        // the second mesh is loaded next to the first and added between generateScene() and ui.setScene()
        const a = 'loadPLY(MODEL_FILE_NAME).then(function(data) {';
        const b = 'var scene = generateScene(meshes);';
        if (src.indexOf(a) < 0 || src.indexOf(b) < 0) throw new Error('initWebGL hooks not found');
        src = src.replace(a, 'Promise.all([loadPLY(MODEL_FILE_NAME), loadOBJ(' + JSON.stringify(args.extraObj) +
                             ')]).then(function(all) { var data = all[0]; var __extra = all[1];');
        src = src.replace(b, b + `
            (function () {
                var m2 = new Mesh(nextObjectId++, __extra.verts, __extra.faces);
                m2.scale(new THREE.Vector3(${args.extraScale}, ${args.extraScale}, ${args.extraScale}));
                m2.translate(new THREE.Vector3(${args.extraTranslate.join(', ')}));
                m2.setMaterial(scene.objects[scene.objects.length - 1].faces[0].getMaterial());
                for (var t = 0; t < m2.faces.length; t++)      // the harness's rebase (not reference code)
                    for (var j = 0; j < 3; j++) m2.faces[t].indices[j] += m2.startVertIndex;
                scene.objects.push(m2);
            })();`);
        usedSets.extra_mesh = args.extraObj + ' scale ' + args.extraScale + ' translate ' + args.extraTranslate.join(',') +
                              ' (indices rebased by the harness)';
    }
    if (args.sceneJs) {
        // a later declaration of the same function name replaces the earlier one for the whole script (hoisting)
        src = src + '\n' + fs.readFileSync(args.sceneJs, 'utf8') + '\n';
        usedSets.scene_js = path.basename(args.sceneJs);
    }
    return src;
});

// loadPLY resolves through a Promise: give the microtask queue a turn, then drive one tick.
setImmediate(() => {
    if (!tickFn) throw new Error('reference did not start its tick loop');
    const get = (expr) => vm.runInContext(expr, sandbox);
    // one tick = ui.update(time); ui.render()   (js/8-kd_tree.js:2842-2849)
    vm.runInContext('tick(' + args.time + ')', sandbox);
    // checkpoint 6 sets its uniforms AFTER the draw (js/6-transmission.js:1619-1624): a second tick is
    // the first frame that renders with them. Uniform values are identical in both ticks.
    if (args.ticks > 1) for (let k = 1; k < args.ticks; k++) vm.runInContext('tick(' + args.time + ')', sandbox);
    let dragRecord = null;
    if (args.drag) {
        // a click selects, a second press on the selection box grabs, moves follow, release ends (UI.mouseDown /
        // mouseMove / mouseUp, js/8-kd_tree.js:2718-2792, driven like the handlers at :3054-3088 do)
        const d = args.drag;
        sandbox.__drag = d;
        dragRecord = vm.runInContext(`(function () {
            var d = __drag, rec = {};
            rec.select_returned = ui.mouseDown(d[0], d[1]);
            ui.mouseUp(d[0], d[1]);
            var sel = ui.renderer.selectedObject;
            rec.selected_index = ui.scene.objects.indexOf(sel);
            if (sel) {
                rec.model_before = Array.from(sel.modelTransform.elements);
                rec.grab_returned = ui.mouseDown(d[0], d[1]);
                rec.moving = ui.moving;
                rec.zPos = ui.zPos;
                rec.old_position = [ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z];
                rec.positions = [];
                for (var k = 2; k + 1 < d.length; k += 2) {
                    ui.mouseMove(d[k], d[k + 1]);
                    rec.positions.push([ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z]);
                }
                ui.mouseUp(d[d.length - 2], d[d.length - 1]);
                rec.positions.push([ui.oldPosition.x, ui.oldPosition.y, ui.oldPosition.z]);
                rec.model_after = Array.from(sel.modelTransform.elements);
                rec.inv_model_after = Array.from(sel.invModelTransform.elements);
            }
            return rec;
        })()`, sandbox);
        dragRecord.mouse = d;
        // the frame after the drag; by default time has advanced past the rebuild throttle (js/8-kd_tree.js:2495) so the tree
        // follows; with --drag-dt under 0.30 (and a first tick at a time other than 0) the reference renders the moved
        // vertices through the tree of the first tick
        dragRecord.dt = args.dragDt;
        vm.runInContext('tick(' + (args.time + args.dragDt) + ')', sandbox);
    }

    let trackballRecord = null;
    if (args.trackball) {
        // the page's own canvas.onmousedown / onmousemove / onmouseup (js/8-kd_tree.js:3056-3088) get the event first, as
        // in a browser (they call ui.mouseDown etc.; over the background nothing is selected and the controls stay enabled)
        const pose = () => get(`({ position: [camera.position.x, camera.position.y, camera.position.z],
            up: [camera.up.x, camera.up.y, camera.up.z],
            quaternion: [camera.quaternion.x, camera.quaternion.y, camera.quaternion.z, camera.quaternion.w],
            matrixWorld: Array.from(camera.matrixWorld.elements),
            viewProjection: Array.from(ui.renderer.viewProjection.elements),
            target: [ui.controls.target.x, ui.controls.target.y, ui.controls.target.z] })`);
        const fire = (type, ev) => {
            ev.type = type; ev.preventDefault = () => {}; ev.stopPropagation = () => {};
            ev.clientX = ev.pageX; ev.clientY = ev.pageY;
            const on = canvas['on' + type];
            if (on) on(ev);
            for (const fn of (canvasListeners[type] || []).slice()) fn(ev);
            for (const fn of (documentListeners[type] || []).slice()) fn(ev);
        };
        trackballRecord = { steps: args.trackball, frames: [], start: pose() };
        let t = args.time, lastX = 0, lastY = 0;
        for (const step of args.trackball) {
            const f = step.split(':');
            if (f[0] === 'down') { lastX = +f[2]; lastY = +f[3]; fire('mousedown', { button: +f[1], pageX: lastX, pageY: lastY }); }
            else if (f[0] === 'move') { lastX = +f[1]; lastY = +f[2]; fire('mousemove', { pageX: lastX, pageY: lastY }); }
            else if (f[0] === 'up') { lastX = +f[1]; lastY = +f[2]; fire('mouseup', { pageX: lastX, pageY: lastY }); }
            else if (f[0] === 'wheel') fire('wheel', { deltaY: +f[1], deltaMode: 0, pageX: lastX, pageY: lastY });
            else if (f[0] === 'tick') { t += 1 / 60; vm.runInContext('tick(' + t + ')', sandbox); trackballRecord.frames.push(pose()); }
            else throw new Error('unknown trackball step ' + step);
        }
        trackballRecord.last_time = t;
    }

    if (args.dumpGlsl) {
        // The generated shader text and the data it reads, for the build-container-only execution harness
        // (oracle/gen_golden/glsl_exec). The text is the reference's: it must not land inside the repository.
        const repo = path.resolve(__dirname, '..', '..');
        const dir = path.resolve(args.dumpGlsl);
        if (dir === repo || dir.startsWith(repo + path.sep)) throw new Error('--dump-glsl must point outside the repository');
        fs.mkdirSync(dir, { recursive: true });
        shaderSources.forEach((src, i) => fs.writeFileSync(path.join(dir, 'shader' + i + '.glsl'), src));
        const tex = [];
        let unit = null, nTex = 0;
        for (const [fn, a] of glCalls) {
            if (fn === 'activeTexture') unit = a[0];
            if ((fn === 'texSubImage2D' || fn === 'texImage2D') && a[a.length - 1] && a[a.length - 1].buffer) {
                const ta = a[a.length - 1];
                const name = 'tex' + (nTex++) + '.bin';
                fs.writeFileSync(path.join(dir, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength));
                tex.push({ file: name, call: fn, unit, args: a.slice(0, a.length - 1).map((v) => (typeof v === 'object' ? null : v)), type: ta.constructor.name, length: ta.length });
            }
        }
        const un = [];
        for (const [fn, a] of glCalls) {
            if (/^uniform(Matrix)?[1-4][fi]v?$/.test(fn)) {
                const val = fn.startsWith('uniformMatrix') ? a[2] : a[1];
                un.push({ name: a[0], fn, value: (typeof val === 'number') ? [val] : Array.from(val) });
            }
        }
        fs.writeFileSync(path.join(dir, 'dump.json'), JSON.stringify({ script: args.script, width: args.width, height: args.height, traced_width: get('Number(WIDTH) * (typeof SUPERSAMPLING !== "undefined" ? SUPERSAMPLING : 1)'), traced_height: get('Number(HEIGHT) * (typeof SUPERSAMPLING !== "undefined" ? SUPERSAMPLING : 1)'), textures: tex, uniforms: un }));
    }
    fs.mkdirSync(args.out, { recursive: true });
    const sha = (buf) => crypto.createHash('sha256').update(Buffer.from(buf.buffer, buf.byteOffset, buf.byteLength)).digest('hex');
    const writeBin = (name, ta) => fs.writeFileSync(path.join(args.out, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength));

    // ---- uniforms, in call order -----------------------------------------------------
    const uniforms = [];
    for (const [fn, a] of glCalls) {
        if (/^uniform(Matrix)?[1-4][fi]v?$/.test(fn)) {
            const name = a[0];
            const val = fn.startsWith('uniformMatrix') ? a[2] : a[1];
            uniforms.push({ name, fn, value: (typeof val === 'number') ? [val] : Array.from(val) });
        }
    }

    const rt = get('ui.renderer.rayTracer');
    // checkpoint 6 has no global mesh arrays (its `triangles` list stays empty, js/6-transmission.js:1225-1227)
    const has = (n) => get('typeof ' + n + ' !== "undefined"');
    const gVerts = has('g_verts') ? get('g_verts') : [], gFaces = has('g_faces') ? get('g_faces') : [];
    const nV = gVerts.length, nF = gFaces.length;
    const v64 = new Float64Array(nV * 3), v32 = new Float32Array(nV * 3);
    for (let i = 0; i < nV; i++) {
        v64[3 * i] = gVerts[i].x; v64[3 * i + 1] = gVerts[i].y; v64[3 * i + 2] = gVerts[i].z;
        v32[3 * i] = gVerts[i].x; v32[3 * i + 1] = gVerts[i].y; v32[3 * i + 2] = gVerts[i].z;
    }
    const faces = new Uint32Array(nF * 3), faceMat = new Uint32Array(nF * 2);
    for (let i = 0; i < nF; i++) {
        for (let j = 0; j < 3; j++) faces[3 * i + j] = gFaces[i].indices[j];
        const m = gFaces[i].getMaterial ? gFaces[i].getMaterial() : gFaces[i].material;
        if (m) { faceMat[2 * i] = m.getType(); faceMat[2 * i + 1] = m.getId(); }
    }

    const tracerSrc = shaderSources.find((t) => /castRayRecursive/.test(t)) || '';
    const cnt = (re) => { const m = re.exec(tracerSrc); return m ? +m[1] : null; };
    const glslCounts = {
        pointLights: cnt(/uniform PointLight pointLights\[(\d+)\]/),
        light_loop: cnt(/for \(int i = 0; i < (\d+); i\+\+\) \{\s*vec3 S = normalize/),
        phongMaterials: cnt(/uniform Phong phongMaterials\[(\d+)\]/),
        phongBlinnMaterials: cnt(/uniform Phong phongBlinnMaterials\[(\d+)\]/),
        phongCheckeredMaterials: cnt(/uniform PhongCheckered phongCheckeredMaterials\[(\d+)\]/),
        ray_tree_size: cnt(/RAY_TREE_SIZE = int\((\d+)\)/),
        // the numeric constants as the generated shader text spells them
        constants: (() => {
            const o = {};
            for (const name of ['INFINITY', 'EPSILON', 'SPHERE_EPSILON', 'PLANE_RECT_EPSILON', 'TRIANGLE_EPSILON']) {
                const m = new RegExp('const float ' + name + ' = (?:float\\()?([-+0-9.eE]+)\\)?;').exec(tracerSrc);
                o[name] = m ? +m[1] : null;
            }
            return o;
        })(),
        num_faces: cnt(/const int numFaces = (\d+)/),
        closest_tests: (tracerSrc.match(/&& \(hitInfo\.t < tMin\)\) \{\s*hitInfo\.hit = true;\s*hitPoint = rayOrigin \+ rayDir \* hitInfo\.t;\s*tMin = hitInfo\.t;\s*normal = hitInfo\.normal;\s*localHitPoint = hitInfo\.localHitPoint;\s*materialType = \d+;/g) || []).length,
        intersect_order: (tracerSrc.match(/if\(\s*intersect(\w+)\(rayOrigin, rayDir, hitInfo, model\d+, invModel\d+\)\s*&& \(hitInfo\.t < tMin\)/g) || [])
            .map((t) => /intersect(\w+)\(/.exec(t)[1]),
        material_consts: (tracerSrc.match(/materialType = \d+;\s*materialId = \d+;/g) || [])
            .map((t) => t.match(/\d+/g).map(Number)),
    };
    const manifest = {
        generator: 'oracle/gen_golden/ref_capture.js',
        script: args.script, width: args.width, height: args.height, time: args.time,
        constant_edits: usedSets, pixel_pattern: args.pixelPattern,
        WIDTH: get('WIDTH'), HEIGHT: get('HEIGHT'),
        n_verts: nV, n_faces: nF,
        sha256: {
            verts_f64: sha(v64), verts_f32: sha(v32), faces_u32: sha(faces), face_mat_u32: sha(faceMat),
        },
        uniforms,
        // array sizes / loop trip counts baked into the generated GLSL (not the GLSL text itself)
        glsl_counts: glslCounts,
        n_triangles_ch6: has('triangles') ? get('triangles.length') : null,
        gl_call_count: glCalls.length, shader_source_count: shaderSources.length,
        camera: {
            position: get('[camera.position.x, camera.position.y, camera.position.z]'),
            matrixWorld: Array.from(get('camera.matrixWorld.elements')),
            projectionMatrix: Array.from(get('camera.projectionMatrix.elements')),
        },
        Lbar: rt ? rt.Lbar : null, sf: rt ? rt.sf : null,
        first_vert_f64: nV ? [v64[0], v64[1], v64[2]] : null,
        drag: dragRecord,
        trackball: trackballRecord,
    };

    if (rt && rt.kdTree) {
        const data = rt.kdTreeData;
        let n = 0;
        for (let i = 0; i < data.length; i++) if (data[i] !== 0) n++;
        const pairs = new Uint32Array(n * 2);
        let k = 0, nInt = 0, nLeaf = 0, nEmpty = 0;
        for (let i = 0; i < data.length; i++) {
            if (data[i] !== 0) {
                pairs[k++] = i; pairs[k++] = data[i];
                const t = data[i] >>> 30;
                if (t === 1) nInt++; else if (t === 2) nEmpty++; else nLeaf++;
            }
        }
        const nl = rt.kdTree.leafNodes.length;
        const leaves = new Int32Array(nl * 4);
        let truncated = 0;
        for (let i = 0; i < nl; i++) for (let j = 0; j < 4; j++) leaves[4 * i + j] = rt.kdTree.leafNodes[i].faceIndices[j];
        const root = rt.kdTree.root;
        const box = new Float64Array([root.min.x, root.min.y, root.min.z, root.max.x, root.max.y, root.max.z]);
        manifest.kd = {
            n_nodes: n, n_internal: nInt, n_leaf: nLeaf, n_empty: nEmpty, n_leaves: nl, depth: rt.kdTree.depth,
            root_box_f64: Array.from(box),
            sha256: { nodes_sparse_u32: sha(pairs), leaves_i32: sha(leaves), root_box_f64: sha(box) },
        };
        if (args.fullArrays) { writeBin('kd_nodes_sparse_u32.bin', pairs); writeBin('kd_leaves_i32.bin', leaves); }
    }
    if (args.fullArrays) {
        writeBin('verts_f64.bin', v64); writeBin('faces_u32.bin', faces); writeBin('face_mat_u32.bin', faceMat);
    }
    fs.writeFileSync(path.join(args.out, 'manifest.json'), JSON.stringify(manifest, null, 1));
    console.log('wrote', args.out, 'verts', nV, 'faces', nF, manifest.kd ? JSON.stringify(manifest.kd.sha256) : '');
});
